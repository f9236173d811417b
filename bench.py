#!/usr/bin/env python
"""Episode throughput of the few-shot detection hot path on MI355X.

One "step" = one episode: B query images (SxS) + N support images with masks (SmxSm) through the
reweighting net, the Darknet-19 meta feature extractor, the fused reweighting (x) 1x1 head and
RegionLossV2 (+ backward + SGD in --mode train).  Inputs are synthetic and resident in HBM before the
timed region.

Headline workload = the episode BASELINE.json's `metric` string quotes: 64 queries 416x416 + 20 supports 224x224 on
darknet_dynamic.cfg + reweighting_net.cfg, fp32, train step, 1 MI355X.  BASELINE configs[1] as the cfg files spell it
(15 base classes, supports at the cfg's 416x416) is timed in the same run and reported under `also_measured`
(`--classes 15 --support 416` makes it the headline line instead).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3 [--scaling strong]

Multi-GPU: one process per GPU over RCCL.  --scaling weak (default): every rank runs its own episode (B queries + its
own N supports, like the reference's per-GPU MetaDataset draw).  --scaling strong: ONE global episode per step, its B
queries split over the ranks, the N supports replicated on every rank.  In train mode gradients are SUM
all-reduced.  Rank 0 prints ONE JSON line.
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
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured with a float4 copy)


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


def episode_flops(blocks, lblocks, B, N, S, Sm):
    """Algorithmic forward FLOPs of one episode: detector on B queries + reweighting net on N supports + the
    1x1 head on B*N (image, class) rows."""
    g = S // 32
    head = 2.0 * 1024 * 30 * g * g
    return B * conv_flops_per_image(blocks, S) + N * conv_flops_per_image(lblocks, Sm) + head * N * B - B * head


def cpu_baseline_and_parity(dyn_cfg, rw_cfg, args, full_flops, dev):
    """The oracle (PyTorch-CPU fp32 restatement of the reference) timed on this host on a bounded sample of the same
    workload -- a smaller query batch, scaled to the full episode by conv FLOPs; median of 3 repetitions after one
    warm-up.  The oracle's outputs on that sample are then the checker for the HIP path on the same
    weights and inputs: BASELINE.json's metric names "RegionLoss max|delta| vs ref"."""
    from oracle.net import OracleDarknet
    from oracle.region import region_loss_v2
    from fewshot_detection_amd.cfg import cfg, parse_cfg
    from fewshot_detection_amd.darknet_meta import Darknet
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    Bs, Ns = min(args.batch, 32), args.classes          # about 7 s of CPU work per repetition on a 64-core host
    torch.manual_seed(4242)
    ora = OracleDarknet(dyn_cfg, rw_cfg).train()
    x, metax, mask, tgt = synth_episode(123, Bs, Ns, args.size, args.support)
    blocks, lblocks = parse_cfg(dyn_cfg), parse_cfg(rw_cfg)
    sample_flops = episode_flops(blocks, lblocks, Bs, Ns, args.size, args.support)
    state = {k: v.clone() for k, v in ora.state_dict().items()}        # before train-mode forwards move the BN statistics

    def once():
        ora.zero_grad()
        t0 = time.time()
        out = ora(x, metax, mask)            # the CPU baseline is the reference's arithmetic: fp32
        r = region_loss_v2(out, tgt, ora.region.anchors, seen=0)
        if args.mode == "train":
            r["loss"].backward()
        return time.time() - t0, out, r

    once()
    reps = []
    for _ in range(3):
        ora.load_state_dict(state)
        t, out, r = once()
        reps.append(t)
    t = sorted(reps)[1]
    eps = 1.0 / (t * full_flops / sample_flops)
    base = {"value": eps, "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": "oracle (PyTorch-CPU fp32) %s of B=%d queries %dx%d + N=%d supports %dx%d: median of 3 repetitions "
                      "after 1 warm-up = %.2f s (%s), scaled by conv FLOPs (%.1f -> %.1f GFLOP forward) to the full episode"
                      % (args.mode, Bs, args.size, args.size, Ns, args.support, args.support, t,
                         ", ".join("%.2f" % v for v in reps), sample_flops / 1e9, full_flops / 1e9)}
    if args.no_parity:
        return base, None

    # ---- parity of the HIP path on the very same sample, weights and (for the loss) inputs --------------------------
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        net2 = Darknet(dyn_cfg, rw_cfg)
    net2.load_state_dict(state)
    net2 = net2.to(dev).train().set_compute_dtype(args.dtype)
    region2 = net2.models[len(net2.models) - 1]
    region2.verbose = False
    region2.seen = 0
    keep_neg = cfg.neg_ratio
    cfg.neg_ratio = "full"
    try:
        hip_out = net2(x.to(dev), metax.to(dev), mask.to(dev))
        hip_loss = region2(hip_out, tgt)
        hip_out_cpu = hip_out.detach().cpu()
        # the loss on IDENTICAL inputs: feed the HIP network's own output to both implementations
        leaf = hip_out.detach().clone().requires_grad_(True)
        loss_same = region2(leaf, tgt)
        loss_same.backward()
        ref_in = hip_out_cpu.clone().requires_grad_(True)
        r_same = region_loss_v2(ref_in, tgt, ora.region.anchors, seen=0)
        r_same["loss"].backward()
        st = region2.stats()
    finally:
        cfg.neg_ratio = keep_neg
    if args.dtype == "bf16":                 # the checker of the bf16 mode is the oracle's restatement of that mode
        ora.load_state_dict(state)
        with torch.no_grad():
            out, _ = ora.forward_bf16(x, metax, mask)
        r = region_loss_v2(out, tgt, ora.region.anchors, seen=0)
    ref_out = out.detach()
    ref_loss = float(r["loss"].detach())
    parity = {
        "config": "B=%d queries %dx%d + N=%d supports %dx%d, train-mode BatchNorm, neg_ratio=full, seen=0, fp32 oracle "
                  "weights loaded into the HIP model (%s compute)" % (Bs, args.size, args.size, Ns, args.support,
                                                                      args.support, args.dtype),
        "forward_max_abs_delta": float((hip_out_cpu - ref_out).abs().max()),
        "forward_max_abs": float(ref_out.abs().max()),
        "forward_rel_l2": float((hip_out_cpu - ref_out).norm() / ref_out.norm()),
        "region_loss_end_to_end": {"hip": float(hip_loss.detach()), "oracle": ref_loss,
                                   "abs_delta": abs(float(hip_loss.detach()) - ref_loss),
                                   "rel_delta": abs(float(hip_loss.detach()) - ref_loss) / max(1.0, abs(ref_loss))},
        # RegionLoss on identical inputs (the head output the HIP network produced)
        "region_loss_abs_delta": abs(float(loss_same.detach()) - float(r_same["loss"].detach())),
        "region_loss_max_abs_delta": float((leaf.grad.cpu() - ref_in.grad).abs().max()),
        "region_loss_grad_max_abs": float(ref_in.grad.abs().max()),
        "anchor_assignment_equal": bool((st["nGT"], st["nCorrect"], st["nProposals"]) ==
                                        (r_same["nGT"], r_same["nCorrect"], r_same["nProposals"])),
        "tolerance": 1e-3,
    }
    # bf16 mode: both runs round at the same points; a rounding-boundary flip in layer 0 (2.7e-5) is amplified ~1.35x per
    # layer by this randomly initialised net (tests/test_gpu_bf16.py pins every layer to 1e-4 on identical inputs)
    parity["ok"] = bool((parity["forward_max_abs_delta"] < 1e-3 if args.dtype == "f32" else parity["forward_rel_l2"] < 0.2)
                        and parity["region_loss_max_abs_delta"] < 1e-3 and parity["anchor_assignment_equal"]
                        and parity["region_loss_abs_delta"] < 1e-3 * max(1.0, abs(ref_loss)))
    del net2
    torch.cuda.empty_cache()
    return base, parity


def timed(fn, n=5, w=2):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def extras(net, region, opt, args, dev, x, metax, mask, target, full_flops, blocks, lblocks):
    """More timings of the same model on the same device (N=1 only, ~2 s): the forward pass alone (north_star:
    ">= 0.6x MFMA roofline on the Darknet-19 forward") and BASELINE configs[1] exactly as the cfg files spell it
    (15 base classes, supports 416x416) -- or the metric-string episode when configs[1] is the headline."""
    peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    out = {}

    def fwd():
        with torch.no_grad():
            region(net(x, metax, mask), target)

    t = timed(fwd)
    out["forward_only"] = {"what": "forward (train-mode BN) + RegionLoss forward/grad kernel, no backward, same episode",
                           "ms": t * 1e3, "episodes_per_s": 1.0 / t, "algorithmic_tflops": full_flops / t / 1e12,
                           "frac_of_mfma_peak_algorithmic": full_flops / t / 1e12 / peak}
    if args.mode == "train" and opt is not None:
        other = (15, 416) if (args.classes, args.support) != (15, 416) else (20, 224)
        x2, metax2, mask2, target2 = synth_episode(2000, args.batch, other[0], args.size, other[1])
        metax2, mask2 = metax2.to(dev), mask2.to(dev)

        def train_other():
            region.seen += args.batch
            opt.backward_and_step(region(net(x, metax2, mask2), target2))

        t = timed(train_other)
        fl = episode_flops(blocks, lblocks, args.batch, other[0], args.size, other[1])
        key = "configs1_cfg_episode" if other == (15, 416) else "metric_string_episode"
        out[key] = {"what": "train step on B=%d queries %dx%d + %d supports %dx%d (%s)"
                            % (args.batch, args.size, args.size, other[0], other[1], other[1],
                               "BASELINE configs[1] with the cfg's own support size and 15 base classes" if other == (15, 416)
                               else "the shape in BASELINE.json's metric string"),
                    "ms_per_step": t * 1e3, "episodes_per_s": 1.0 / t, "img_per_s": args.batch / t,
                    "episode_forward_gflop": fl / 1e9, "dtype": args.dtype}
    return out


def pmc_traffic(name):
    """A committed rocprofv3 PMC summary (profiles/<name>, produced by tools/pmc_traffic.py from separate FETCH_SIZE /
    WRITE_SIZE passes of this very command).  Counters cannot be read live; None if the summary is absent."""
    for rnd in ("r02", "r01"):
        p = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, name))
        try:
            return json.load(open(p)), "profiles/%s_%s (offline rocprofv3 PMC passes of this command)" % (rnd, name)
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64, help="query images per GPU (weak) / per global episode (strong)")
    ap.add_argument("--classes", type=int, default=20, help="episode classes N = support images")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--support", type=int, default=224, help="support image side (BASELINE.json metric: 224; cfg/reweighting_net.cfg: 416)")
    ap.add_argument("--mode", choices=["train", "forward"], default=None)
    ap.add_argument("--neg", default="1", help="cfg.neg_ratio ('full' or a number; metayolo.data uses 1)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="conv compute mode: f32 = exact fp32 MFMA (BASELINE C2, default); bf16 = bf16 operands, fp32 "
                         "accumulate, fp32 BN/loss/master weights (BASELINE C3/C5)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="multi-GPU: weak = --batch queries + N supports per rank; strong = --batch queries split over "
                         "the ranks, supports replicated (SURVEY 8e)")
    ap.add_argument("--profile-steps", type=int, default=2, help="timed steps that carry the per-kernel HIP events")
    ap.add_argument("--streams", type=int, choices=[0, 1], default=None,
                    help="side HIP streams (reweighting net, weight gradients, target upload beside the main stream); "
                         "default: on unless FSD_STREAMS=0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the HIP-vs-oracle comparison on the cpu_baseline sample")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward-only / other-episode timings")
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
    strong = args.scaling == "strong" and world > 1
    if strong and args.batch % world:
        raise SystemExit("--scaling strong needs --batch (%d) divisible by the number of ranks (%d)" % (args.batch, world))
    local_batch = args.batch // world if strong else args.batch
    global_batch = args.batch if strong else args.batch * world

    from fewshot_detection_amd import backward as bw
    from fewshot_detection_amd import cfgs, ops, streams
    from fewshot_detection_amd.cfg import cfg, parse_cfg
    from fewshot_detection_amd.darknet_meta import Darknet

    if args.mode is None:
        args.mode = "train" if getattr(bw, "AVAILABLE", False) else "forward"
    if args.streams is not None:
        streams.ENABLED = bool(args.streams)
    streams_on = streams.ENABLED
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
    if strong:      # one global episode: this rank's slice of the queries and targets, every support on every rank
        gx, metax, mask, gt = synth_episode(1000, args.batch, args.classes, args.size, args.support)
        x, target = gx[rank * local_batch:(rank + 1) * local_batch], gt[rank * local_batch:(rank + 1) * local_batch]
    else:
        x, metax, mask, target = synth_episode(1000 + rank, args.batch, args.classes, args.size, args.support)
    x, metax, mask = x.to(dev).contiguous(), metax.to(dev), mask.to(dev)

    opt = None
    if args.mode == "train":
        from fewshot_detection_amd.dp import EpisodeTrainer
        # train_meta.py:123-147: lr = 0.001/factor/global_batch, wd = decay*global_batch*factor (factor 3 for
        # neg=1).  From RANDOM init (no pretrained darknet19 weights here) that step size diverges within
        # two steps, so the bench shrinks lr by 1e-4; the work per step is unchanged.
        opt = EpisodeTrainer(net, lr=1e-4 * 0.001 / 3 / global_batch, momentum=0.9,
                             weight_decay=0.0005 * global_batch * 3, process_group=dist,
                             grad_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32)
        opt.time_allreduce = world > 1

    def step():
        region.seen += global_batch
        out = net(x, metax, mask)
        loss = region(out, target)
        if opt is not None:
            opt.backward_and_step(loss)
        return loss

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The shader clock the chip sustains under matrix-core load (a dependent-MFMA chain, ~1 ms).  A box that was throttled
    # by an earlier tenant / process shows up here as a fraction of the nominal 2400 MHz: wait (bounded) for it to recover
    # instead of timing a throttled GPU, and say so in the line.
    clock = {"nominal_mhz": 2400.0, "probe_mhz_start": ops.clock_probe_mhz(dev), "waited_s": 0.0}
    t_wait = time.perf_counter()
    while clock["probe_mhz_start"] < 0.6 * 2400.0 and time.perf_counter() - t_wait < 30.0:
        time.sleep(3.0)
        clock["probe_mhz_start"] = ops.clock_probe_mhz(dev)
        clock["waited_s"] = time.perf_counter() - t_wait
    for _ in range(args.warmup):
        step()
    fence()
    clock["probe_mhz_after_warmup"] = ops.clock_probe_mhz(dev)
    if opt is not None:
        opt.allreduce_wait_ms = [0.0] * len(opt.buckets)
    # the per-launch event records below create python objects: keep the cyclic collector from stopping the host for a
    # full-heap pass in the middle of the timed region
    gc.collect()
    gc.disable()
    # Per-kernel HIP events (roofline) are recorded INSIDE the timed region, on its first `prof_steps` steps only: each
    # record is a barrier packet in the queue and a few hundred of them per step cost ~1 ms of the step.
    # The profiled steps run on ONE stream: with the side streams a kernel shares the chip with the launches of the other
    # strands and its duration says how the chip was shared, not how good the kernel is.  They are part of the timed
    # region (the headline therefore includes a few un-overlapped steps); ms_per_step_unprofiled is the rest ...
    prof_steps = min(args.steps, args.profile_steps) if rank == 0 else 0
    # ... and they sit in the MIDDLE of the timed region: the first step after the fence runs at ramping clocks (its
    # kernels measured 6-8 % slower than the same kernels a few steps later)
    prof_lo = (args.steps - prof_steps) // 2
    prof_hi = prof_lo + prof_steps
    prof = []
    t_a = t_b = None
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == prof_lo and prof_steps:
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            ops.PROFILE = prof
            ops.kernel_profile(True)
            streams.ENABLED = False
        elif i == prof_hi and prof_steps:
            ops.PROFILE = None
            ops.kernel_profile(False)
            streams.ENABLED = streams_on
            torch.cuda.synchronize()
            t_b = time.perf_counter()
        loss = step()
    ops.PROFILE = None
    ops.kernel_profile(False)
    streams.ENABLED = streams_on
    fence()
    t_end = time.perf_counter()
    if prof_steps and t_b is None:           # the profiled steps were the last ones
        t_b = t_end
    elapsed = t_end - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss)
    assert np.isfinite(loss_val), "non-finite loss"
    clock["probe_mhz_after_timing"] = ops.clock_probe_mhz(dev)
    clock["throttled"] = bool(min(clock["probe_mhz_after_warmup"], clock["probe_mhz_after_timing"]) < 0.6 * 2400.0)

    if rank == 0:
        peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        kp = ops.kernel_profile_collect()
        per = max(1, prof_steps)

        def mfma(cls):
            k = kp[cls]
            tf = k["work"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            return {"achieved": tf, "peak": peak if cls != "gemm_bf16" else PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / (peak if cls != "gemm_bf16" else PEAK_BF16_MFMA_TFLOPS),
                    "kernel_ms_per_step": k["ms"] / per, "launches_per_step": k["launches"] / per,
                    "avg_kernel_ms": k["ms"] / max(1, k["launches"]), "issued_gflop_per_step": k["work"] / per / 1e9}

        def hbm(cls):
            k = kp[cls]
            gbs = k["work"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
            return {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                    "kernel_ms_per_step": k["ms"] / per, "launches_per_step": k["launches"] / per,
                    "algorithmic_mb_per_step": k["work"] / per / 1e6}

        conv_ms = sum(e[0].elapsed_time(e[1]) for e in prof)
        conv_flops = sum(e[2] for e in prof)
        algorithmic = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        if args.per_layer and prof_steps:
            n_l = len(prof) // prof_steps
            for i in range(n_l):
                ms_i = sum(prof[s_ * n_l + i][0].elapsed_time(prof[s_ * n_l + i][1]) for s_ in range(prof_steps)) / prof_steps
                fl = prof[i][2]
                sys.stderr.write("conv launch %2d: %8.3f ms  %8.2f GFLOP  %6.1f TFLOP/s\n" % (i, ms_i, fl / 1e9, fl / ms_i / 1e9))
        blocks, lblocks = parse_cfg(dyn_cfg), parse_cfg(rw_cfg)
        full_flops = episode_flops(blocks, lblocks, local_batch, args.classes, args.size, args.support)
        ms = elapsed / args.steps * 1e3
        episodes_per_step = 1 if strong else world
        dom = "gemm_fwd" if args.dtype == "f32" or kp["gemm_bf16"]["ms"] < kp["gemm_fwd"]["ms"] else "gemm_bf16"
        roof = mfma(dom)
        traffic, traffic_src = pmc_traffic("conv_traffic.json")
        is_c2 = (args.mode == "train" and local_batch == 64 and args.dtype == "f32")
        roof.update({
            "bound": "mfma",
            "kernel": ("conv_gemm_kernel: the fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit-GEMM kernel behind the direct "
                       "3x3/1x1 convolutions, the data gradients and the 36 / 16 position GEMMs of the Winograd layers"
                       if dom == "gemm_fwd" else "conv_gemm_bf16_kernel: bf16-operand MFMA implicit GEMM"),
            "note": "achieved = MFMA FLOPs this kernel really issues (2*rows*Cout*K per launch; the Winograd layers count "
                    "their (tile+2)^2 position GEMMs, i.e. 4x / 2.25x fewer multiplications than the direct algorithm) / "
                    "its own duration, HIP events recorded by the library right around every launch on the launch stream "
                    "during `profiled_steps` steps in the middle of the timed region, which run on one stream (no side-stream overlap: a "
                    "kernel's duration in isolation); avg_kernel_ms is what rocprofv3 --kernel-trace --stats shows for this "
                    "kernel under `bench.py --streams 0` (profiles/)",
            "profiled_steps": prof_steps,
            "traffic": (traffic or {}).get("hbm_bytes_per_launch") if is_c2 else None,
            "traffic_source": traffic_src if is_c2 else None,
            "traffic_note": "HBM bytes per CONV LAUNCH (direct kernel, or transform + GEMM + transform of a Winograd layer), "
                            "FETCH_SIZE x2 + WRITE_SIZE from separate --pmc passes; measured on the configs[1] episode",
            "algorithmic_speedup": {
                "what": "direct-convolution FLOPs (2*k*k*Cin*Cout*pixels, SURVEY 8d) of the forward + data-gradient conv "
                        "launches / the HIP-event time of those launches (transforms included)",
                "algorithmic_tflops": algorithmic, "x_mfma_peak": algorithmic / peak,
                "conv_launch_ms_per_step": conv_ms / per, "launches_per_step": len(prof) // per},
            "wgrad_kernel": dict(mfma("gemm_wgrad"), kernel="wgrad_kernel: fp32 MFMA weight-gradient reduction GEMMs "
                                                            "(direct layers and the F(3x3,4x4) Winograd batches)"),
            "hbm": dict(hbm("wino_transform"), bound="hbm",
                        kernel="Winograd input / output / gradient transform kernels (wino4_input, wino4_output, wino4_dy, ...)",
                        note="achieved = algorithmic bytes (activation once + transformed positions once, per launch) / "
                             "kernel duration"),
            "hbm_other": {"bn_leaky_pool_backward": hbm("act_bwd"), "bn_leaky_pool_forward": hbm("act_fwd"),
                          "region_loss": hbm("region"), "sgd": hbm("sgd"), "first_layer": hbm("first_layer")},
        })
        if kp["gemm_bf16"]["launches"] and dom != "gemm_bf16":
            roof["bf16_kernels"] = mfma("gemm_bf16")
        mf = kp["gemm_fwd"]["ms"] + kp["gemm_wgrad"]["ms"] + kp["gemm_bf16"]["ms"]
        mw = kp["gemm_fwd"]["work"] + kp["gemm_wgrad"]["work"] + kp["gemm_bf16"]["work"]
        roof["mfma_all"] = {"issued_tflops": mw / (mf * 1e-3) / 1e12 if mf > 0 else 0.0, "kernel_ms_per_step": mf / per,
                            "issued_gflop_per_step": mw / per / 1e9,
                            "frac_of_step_time": (mf / per) / ms if ms > 0 else 0.0,
                            "whole_step_issued_tflops": (mw / per) / (ms * 1e-3) / 1e12}
        roof["timed_kernel_ms_per_step"] = sum(v["ms"] for v in kp.values()) / per
        sname = "B=%d queries %dx%d + N=%d supports %dx%d" % (local_batch, args.size, args.size, args.classes,
                                                                args.support, args.support)
        which = ("the episode of BASELINE.json's metric string (64x416x416 query + 20x224x224 support) on configs[1]'s "
                 "darknet_dynamic.cfg + reweighting_net.cfg base-training model"
                 if (args.batch, args.classes, args.size, args.support) == (64, 20, 416, 224) else
                 "BASELINE configs[1] darknet_dynamic.cfg + reweighting_net.cfg base-training episode")
        res = {
            "metric": "episodes/sec (%dx%dx%d query + %dx%dx%d support) %s" % (
                args.batch, args.size, args.size, args.classes, args.support, args.support,
                "train step (fwd + RegionLoss + bwd + SGD)" if args.mode == "train" else "forward + RegionLoss fwd/grad"),
            "value": episodes_per_step * args.steps / elapsed, "unit": "episodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "img_per_s": global_batch * args.steps / elapsed,
            "loss": loss_val,
            "config": {"workload": "%s: %s per %s, %s, neg_ratio=%s" % (
                           which, sname, "rank (supports replicated, queries split)" if strong else "GPU",
                           "fp32" if args.dtype == "f32" else "bf16 convs / fp32 BN+loss+master weights", args.neg),
                       "mode": args.mode, "global_batch": global_batch, "parallelism": "dp%d" % world,
                       "episode_forward_gflop": full_flops / 1e9},
            "roofline": roof,
            "gpu_clock": dict(clock, what="shader clock from a dependent fp32-MFMA chain on every SIMD (fsd_clock_probe), right "
                                          "before the warm-up, after it and after the timed region; the MFMA peaks in "
                                          "`roofline` are quoted at the nominal clock"),
            "streams": {"enabled": bool(streams_on),
                        "what": "reweighting net on its own stream beside the detector, weight gradients beside the data-gradient "
                                "chain, target upload on a copy stream (fewshot_detection_amd/streams.py); bit-identical results",
                        "profiled_steps_on_one_stream": prof_steps,
                        "profiled_step_index": [prof_lo, prof_hi] if prof_steps else None,
                        "ms_per_step_unprofiled": (((t_a - t0) + (t_end - t_b)) / (args.steps - prof_steps) * 1e3
                                                   if t_a is not None and args.steps > prof_steps else None),
                        "ms_per_step_profiled": ((t_b - t_a) / prof_steps * 1e3 if t_a is not None else None)},
        }
        if opt is not None:
            res["dp"] = {"world_size": opt.world_size, "backend": backend if world > 1 else None, "scaling": args.scaling,
                         "gradient_buckets": len(opt.buckets), "allreduce_dtype": str(opt.grad_dtype).replace("torch.", ""), "bucket_mb": [4e-6 * (hi - lo) for lo, hi in opt.buckets],
                         "allreduce_wait_ms_per_step": [v / args.steps for v in opt.allreduce_wait_ms]}
        if world == 1 and not args.no_extras:
            res["also_measured"] = extras(net, region, opt, args, dev, x, metax, mask, target, full_flops, blocks, lblocks)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"], parity = cpu_baseline_and_parity(dyn_cfg, rw_cfg, args, full_flops, dev)
            if parity is not None:
                res["parity"] = parity
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
