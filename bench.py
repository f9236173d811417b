#!/usr/bin/env python
"""Episode throughput of the few-shot detection hot path on MI355X.

One "step" = one episode: B query images (SxS) + N support images with masks (SmxSm) through the
reweighting net, the Darknet-19 meta feature extractor, the fused reweighting (x) 1x1 head and
RegionLossV2 (+ backward + SGD in --mode train).  Inputs are synthetic and resident in HBM before the
timed region.  Default workload = BASELINE.json configs[1]: darknet_dynamic.cfg + reweighting_net.cfg,
B=64, 15 base classes, 416x416, fp32, 1 MI355X.  (BASELINE.json's metric STRING quotes "64x416x416 query + 20x224x224
support": 20 supports of 224x224, which neither configs[1] (15 base classes) nor the cfg (support 416x416,
cfg/reweighting_net.cfg:4-5) has.  The default is the heavier, cfg-true episode -- 2022 vs 1941.5 GFLOP forward -- and
the metric-string episode is timed as well and reported under `also_measured`; `--classes 20 --support 224` makes it
the headline line.)

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3

Multi-GPU: one process per GPU, each rank runs its own episode shard (B queries + its own N supports,
like the reference's per-GPU MetaDataset draw) -> weak scaling; in train mode gradients are SUM
all-reduced over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import gc
import json
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 chip peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (AMD's 5 PF figure includes 2:1 sparsity)


def synth_targets(rng, bs, cs):
    """(bs, cs, 250) float64: 1-5 boxes per image, [cls, cx, cy, w, h], zero-terminated."""
    tgt = np.zeros((bs, cs, 250), np.float64)
    fill = np.zeros((bs, cs), np.int64)
    for b in range(bs):
        for _ in range(rng.randint(1, 6)):
            n = rng.randint(0, cs)
            w, h = rng.uniform(0.05, 0.5, 2)
            cx = float(np.clip(rng.uniform(0.1, 0.9), w / 2, 0.999 - w / 2))
            cy = float(np.clip(rng.uniform(0.1, 0.9), h / 2, 0.999 - h / 2))
            t = fill[b, n]
            tgt[b, n, 5 * t:5 * t + 5] = [n, cx, cy, w, h]
            fill[b, n] += 1
    return tgt


def synth_episode(seed, B, N, S, Sm):
    g = torch.Generator().manual_seed(seed)
    rng = np.random.RandomState(seed)
    x = torch.rand(B, 3, S, S, generator=g)
    metax = torch.rand(N, 3, Sm, Sm, generator=g)
    mask = torch.zeros(N, 1, Sm, Sm)
    for n in range(N):
        y0, x0 = rng.randint(0, Sm // 2, 2)
        h, w = rng.randint(Sm // 8, Sm // 2, 2)
        mask[n, 0, y0:y0 + h, x0:x0 + w] = 1
    return x, metax, mask, torch.from_numpy(synth_targets(rng, B, N))


def conv_flops_per_image(blocks, S):
    """2*k*k*Cin*Cout*H*W summed over the convolutional blocks (bias/BN/activation not counted)."""
    total, c, h = 0.0, int(blocks[0]["channels"]), S
    widths = []
    for ind, b in enumerate(blocks[1:]):
        if b["type"] == "convolutional" and not ("dynamic" in b and int(b["dynamic"])):
            co, k = int(b["filters"]), int(b["size"])
            total += 2.0 * k * k * c * co * h * h
            c = co
        elif b["type"] == "maxpool" and int(b["stride"]) == 2:
            h //= 2
        elif b["type"] == "reorg":
            h //= int(b["stride"]); c *= int(b["stride"]) ** 2
        elif b["type"] == "route":
            src = [int(v) if int(v) > 0 else int(v) + ind for v in b["layers"].split(",")]
            c = sum(widths[s][0] for s in src); h = widths[src[0]][1]
        widths.append((c, h))
    return total


def cpu_baseline(dyn_cfg, rw_cfg, args, full_flops):
    """The oracle (PyTorch-CPU fp32 restatement of the reference) timed on this host on a bounded sample
    of the same workload: a smaller episode, scaled to the full episode by conv FLOPs."""
    from oracle.net import OracleDarknet
    from oracle.region import region_loss_v2
    from fewshot_detection_amd.cfg import parse_cfg
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    Bs, Ns = min(args.batch, 32), args.classes          # about 10-20 s of CPU work on a 64-core host
    ora = OracleDarknet(dyn_cfg, rw_cfg).train()
    x, metax, mask, tgt = synth_episode(123, Bs, Ns, args.size, args.support)
    blocks, lblocks = parse_cfg(dyn_cfg), parse_cfg(rw_cfg)
    sample_flops = Bs * conv_flops_per_image(blocks, args.size) + Ns * conv_flops_per_image(lblocks, args.support)
    mult = 3.0 if args.mode == "train" else 1.0

    def once():
        t0 = time.time()
        out = ora(x, metax, mask)
        r = region_loss_v2(out, tgt, ora.region.anchors, seen=0)
        if args.mode == "train":
            r["loss"].backward()
        return time.time() - t0

    once()
    t = once()
    eps = 1.0 / (t * (full_flops * mult) / (sample_flops * mult))
    return {"value": eps, "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": "oracle (PyTorch-CPU fp32) %s of B=%d queries %dx%d + N=%d supports %dx%d in %.2f s, "
                      "scaled by conv FLOPs (%.1f -> %.1f GFLOP) to the full episode"
                      % (args.mode, Bs, args.size, args.size, Ns, args.support, args.support, t,
                         sample_flops / 1e9, full_flops / 1e9)}


def extras(net, region, opt, args, dev, x, metax, mask, target, full_flops, det_flops_img, rw224_flops_img):
    """Two more timings of the same model on the same device (N=1 only, ~1 s): the forward pass alone (the
    north_star's ">= 0.6x MFMA roofline on the forward" target) and the episode shape quoted in BASELINE.json's
    metric string (20 supports of 224x224 instead of configs[1]'s 15 classes at the cfg's 416x416)."""
    peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS

    def timed(fn, n=5, w=2):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def fwd():
        with torch.no_grad():
            region(net(x, metax, mask), target)

    out = {}
    t = timed(fwd)
    out["forward_only"] = {"what": "forward (train-mode BN) + RegionLoss forward/grad kernel, no backward, same episode",
                           "ms": t * 1e3, "episodes_per_s": 1.0 / t, "algorithmic_tflops": full_flops / t / 1e12,
                           "frac_of_mfma_peak": full_flops / t / 1e12 / peak}
    if args.mode == "train" and opt is not None:
        x2, metax2, mask2, target2 = synth_episode(2000, args.batch, 20, args.size, 224)
        metax2, mask2 = metax2.to(dev), mask2.to(dev)

        def train20():
            region.seen += args.batch
            opt.backward_and_step(region(net(x, metax2, mask2), target2))

        t = timed(train20)
        g = args.size // 32
        fl = args.batch * det_flops_img + 20 * rw224_flops_img + 2.0 * 1024 * 30 * (20 - 1) * args.batch * g * g
        out["metric_string_episode"] = {"what": "train step on B=%d queries %dx%d + 20 supports 224x224 (the shape in "
                                                "BASELINE.json's metric string)" % (args.batch, args.size, args.size),
                                        "ms_per_step": t * 1e3, "episodes_per_s": 1.0 / t,
                                        "img_per_s": args.batch / t, "episode_forward_gflop": fl / 1e9}
    return out


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/r01_conv_traffic.json, produced by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE
    passes of this very command).  Counters cannot be read live; None if the summary is absent."""
    p = os.path.join(ROOT, "profiles", "r01_conv_traffic.json")
    try:
        return json.load(open(p))["hbm_bytes_per_launch"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="query images per GPU")
    ap.add_argument("--classes", type=int, default=15, help="episode classes N")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--support", type=int, default=416, help="support image side (cfg/reweighting_net.cfg: 416)")
    ap.add_argument("--mode", choices=["train", "forward"], default=None)
    ap.add_argument("--neg", default="1", help="cfg.neg_ratio ('full' or a number; metayolo.data uses 1)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="conv compute mode: f32 = exact fp32 MFMA (BASELINE C2, default); bf16 = bf16 operands, fp32 "
                         "accumulate, fp32 BN/loss/master weights and fp32 weight gradients (BASELINE C3/C5)")
    ap.add_argument("--profile-steps", type=int, default=5, help="timed steps that carry the per-launch HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward-only / metric-string-episode timings")
    ap.add_argument("--per-layer", action="store_true", help="print per-launch conv timing to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback)"
    # one rank per GPU.  (Functional check of the N>1 path on a single-GPU box: FSD_BENCH_BACKEND=gloo lets several ranks
    # share device 0 -- RCCL refuses two ranks on one device; tests/test_gpu_dp.py uses this.)
    backend = os.environ.get("FSD_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from fewshot_detection_amd import backward as bw
    from fewshot_detection_amd import cfgs, ops
    from fewshot_detection_amd.cfg import cfg, parse_cfg
    from fewshot_detection_amd.darknet_meta import Darknet

    if args.mode is None:
        args.mode = "train" if getattr(bw, "AVAILABLE", False) else "forward"
    cfg.neg_ratio = args.neg if args.neg == "full" else float(args.neg)
    if isinstance(cfg.neg_ratio, float) and cfg.neg_ratio.is_integer():
        cfg.neg_ratio = int(cfg.neg_ratio)
    tmp = tempfile.mkdtemp()
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tmp)
    torch.manual_seed(0)
    random.seed(0)
    with contextlib.redirect_stdout(sys.stderr):      # the constructor prints like the reference; stdout carries ONE JSON line
        net = Darknet(dyn_cfg, rw_cfg).to(dev).train().set_compute_dtype(args.dtype)
    region = net.models[len(net.models) - 1]
    region.verbose = False
    x, metax, mask, target = synth_episode(1000 + rank, args.batch, args.classes, args.size, args.support)
    x, metax, mask = x.to(dev), metax.to(dev), mask.to(dev)

    opt = None
    if args.mode == "train":
        from fewshot_detection_amd.dp import EpisodeTrainer
        # train_meta.py:123-147: lr = 0.001/factor/global_batch, wd = decay*global_batch*factor (factor 3 for
        # neg=1).  From RANDOM init (no pretrained darknet19 weights here) that step size diverges within
        # two steps, so the bench shrinks lr by 1e-4; the work per step is unchanged.
        opt = EpisodeTrainer(net, lr=1e-4 * 0.001 / 3 / (args.batch * world), momentum=0.9,
                             weight_decay=0.0005 * args.batch * world * 3, process_group=dist)

    def step():
        region.seen += args.batch * world
        out = net(x, metax, mask)
        loss = region(out, target)
        if opt is not None:
            opt.backward_and_step(loss)
        return loss

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # the per-launch event records below create thousands of python objects: keep the cyclic collector from
    # stopping the host for a full-heap pass in the middle of the timed region
    gc.collect()
    gc.disable()
    # Per-launch HIP events (roofline) are recorded inside the timed region, on its first `prof_steps` steps only: each
    # record is a barrier packet in the queue and 230 of them per step cost ~1 ms of the 39 ms step.
    prof_steps = min(args.steps, args.profile_steps)
    prof = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.PROFILE = prof if i < prof_steps else None
        loss = step()
    ops.PROFILE = None
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if rank != 0:                        # only rank 0 evaluates the per-launch events
        for e in prof:
            if e[4]:
                ops.lib().fsd_event_destroy(e[4]); ops.lib().fsd_event_destroy(e[5])
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss)
    assert np.isfinite(loss_val), "non-finite loss"

    if rank == 0:
        conv_ms = sum(e[0].elapsed_time(e[1]) for e in prof)
        conv_flops = sum(e[2] for e in prof)
        exec_flops = sum(e[3] for e in prof)
        # the MFMA kernel alone (events recorded inside the library right around conv_gemm_kernel)
        L = ops.lib()
        gemm = [(L.fsd_event_elapsed_ms(e[4], e[5]), e[3]) for e in prof if e[4]]
        for e in prof:
            if e[4]:
                L.fsd_event_destroy(e[4]); L.fsd_event_destroy(e[5])
        gemm = [(m, f) for m, f in gemm if m > 0]
        gemm_ms = sum(m for m, _ in gemm)
        gemm_flops = sum(f for _, f in gemm)
        gemm_tflops = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        if args.per_layer:
            per = len(prof) // max(1, prof_steps)
            for i in range(per):
                ms_i = sum(prof[s * per + i][0].elapsed_time(prof[s * per + i][1]) for s in range(prof_steps)) / prof_steps
                fl = prof[i][2]
                sys.stderr.write("conv launch %2d: %8.3f ms  %8.2f GFLOP  %6.1f TFLOP/s\n" % (i, ms_i, fl / 1e9, fl / ms_i / 1e9))
        blocks, lblocks = parse_cfg(dyn_cfg), parse_cfg(rw_cfg)
        det = conv_flops_per_image(blocks, args.size)
        full_flops = (args.batch * det + args.classes * conv_flops_per_image(lblocks, args.support)
                      + 2.0 * 1024 * 30 * args.classes * args.batch * (args.size // 32) ** 2
                      - args.batch * 2.0 * 1024 * 30 * (args.size // 32) ** 2)
        ms = elapsed / args.steps * 1e3
        res = {
            "metric": "episodes/sec (%dx%dx%d query + %dx%dx%d support) %s" % (
                args.batch, args.size, args.size, args.classes, args.support, args.support,
                "train step (fwd + RegionLoss + bwd + SGD)" if args.mode == "train" else "forward + RegionLoss fwd/grad"),
            "value": world * args.steps / elapsed, "unit": "episodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "img_per_s": world * args.batch * args.steps / elapsed,
            "loss": loss_val,
            "config": {"workload": "BASELINE configs[1]: darknet_dynamic.cfg + reweighting_net.cfg base-training "
                                   "episode, B=%d queries %dx%d + N=%d supports %dx%d per GPU, %s, neg_ratio=%s"
                                   % (args.batch, args.size, args.size, args.classes, args.support, args.support,
                                      "fp32" if args.dtype == "f32" else "bf16 convs / fp32 BN+loss+master weights", args.neg),
                       "mode": args.mode, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "episode_forward_gflop": full_flops / 1e9},
            "roofline": {"bound": "mfma",
                         "kernel": "conv_gemm_kernel (fp32 MFMA implicit GEMM: direct 3x3/1x1 convolutions and the 36 / 16 "
                                   "batched GEMMs of the Winograd F(4x4,3x3) / F(2x2,3x3) layers; forward + data gradient)"
                         if args.dtype == "f32" else "conv_gemm_bf16_kernel (bf16 implicit-GEMM conv, all launches)",
                         "note": "achieved/frac follow the contract: ALGORITHMIC direct-convolution FLOPs "
                                 "(2*k*k*Cin*Cout*pixels, SURVEY 8d) / HIP-event time of the conv launches (a launch = one "
                                 "direct kernel, or Winograd input transform + batched GEMM + output transform). Winograd "
                                 "issues 4x / 2.25x fewer multiplications on its layers, which is why the algorithmic rate "
                                 "can exceed the MFMA peak; `mfma_kernel` is the hardware-utilisation view: FLOPs really "
                                 "issued by conv_gemm_kernel / its own duration (events recorded right around that kernel; "
                                 "avg_kernel_ms is what rocprofv3 --stats shows for conv_gemm_kernel)",
                         "achieved": achieved, "peak": (PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS),
                         "unit": "TFLOP/s",
                         "frac": achieved / (PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS),
                         "traffic": pmc_traffic() if args.mode == "train" and args.batch == 64 and args.dtype == "f32" else None,
                         "mfma_kernel": {"issued_tflops": gemm_tflops, "frac": gemm_tflops / (PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS),
                                         "kernel_ms_per_step": gemm_ms / max(1, prof_steps),
                                         "avg_kernel_ms": gemm_ms / max(1, len(gemm)), "kernels_timed": len(gemm)},
                         "launch_issued_tflops": exec_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
                         "flop_per_launch": conv_flops / max(1, len(prof)),
                         "avg_launch_ms": conv_ms / max(1, len(prof)),
                         "launches_per_step": len(prof) // max(1, prof_steps), "profiled_steps": prof_steps,
                         "conv_ms_per_step": conv_ms / max(1, prof_steps)},
        }
        if world == 1 and not args.no_extras:
            res["also_measured"] = extras(net, region, opt, args, dev, x, metax, mask, target, full_flops,
                                          conv_flops_per_image(blocks, args.size), conv_flops_per_image(lblocks, 224))
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(dyn_cfg, rw_cfg, args, full_flops)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
