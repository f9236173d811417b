"""bench.py's host-side arithmetic: the algorithmic FLOP counts behind `roofline.achieved` are the SURVEY 8(d) /
BASELINE.md figures, the synthetic episode has the contract's shapes, and without a GPU the bench fails loudly."""
import io

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_algorithmic_flops_match_baseline_md(tmp_path):
    import bench
    from fewshot_detection_amd import cfgs
    from fewshot_detection_amd.cfg import parse_cfg
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    det = bench.conv_flops_per_image(parse_cfg(dyn_cfg), 416)            # layers 0-29 + one head 1x1 (layer 31)
    rw416 = bench.conv_flops_per_image(parse_cfg(rw_cfg), 416)
    rw224 = bench.conv_flops_per_image(parse_cfg(rw_cfg), 224)
    head = 2.0 * 1024 * 30 * 13 * 13                                      # 0.0104 GFLOP per (image, class)
    assert abs((det - head) / 1e9 - 29.317) < 0.002                       # full detector, GFLOP / image
    assert abs(rw416 / 1e9 - 9.053) < 0.002 and abs(rw224 / 1e9 - 2.598) < 0.002
    c2 = 64 * det + 15 * rw416 + head * 15 * 64 - 64 * head               # the expression bench.py uses (B=64, N=15)
    assert abs(c2 / 1e9 - 2022.0) < 0.5                                   # "episode forward, C2 (Sm=416)"
    blocks, lblocks = parse_cfg(dyn_cfg), parse_cfg(rw_cfg)
    assert abs(bench.episode_flops(blocks, lblocks, 64, 15, 416, 416) / 1e9 - 2022.0) < 0.5
    assert abs(bench.episode_flops(blocks, lblocks, 64, 20, 416, 224) / 1e9 - 1941.5) < 0.5    # the metric-string episode
    assert abs(bench.episode_flops(blocks, lblocks, 32, 20, 416, 416) / 1e9 - 1125.8) < 0.5    # C4
    det608 = bench.conv_flops_per_image(parse_cfg(dyn_cfg), 608) - 2.0 * 1024 * 30 * 19 * 19
    assert abs(det608 / 1e9 - 62.624) < 0.005                             # C5 detector GFLOP / image


def test_synthetic_episode_contract():
    import bench
    x, metax, mask, tgt = bench.synth_episode(7, 4, 3, 64, 32)
    assert x.shape == (4, 3, 64, 64) and metax.shape == (3, 3, 32, 32) and mask.shape == (3, 1, 32, 32)
    assert tgt.shape == (4, 3, 250) and tgt.dtype == torch.float64
    assert float(x.min()) >= 0 and float(x.max()) < 1                    # ToTensor range, no normalisation
    assert set(np.unique(mask.numpy())) <= {0.0, 1.0} and all(mask[n].sum() > 0 for n in range(3))
    rows = tgt.numpy().reshape(12, 50, 5)
    used = rows[:, :, 3] > 0
    assert used.any()
    for r in range(12):                                                   # zero-terminated, class field = row's class
        k = int(used[r].sum())
        assert not used[r, k:].any() and np.all(rows[r, :k, 0] == r % 3)
    boxes = rows[used]
    assert np.all(boxes[:, 1] - boxes[:, 3] / 2 >= -1e-9) and np.all(boxes[:, 1] + boxes[:, 3] / 2 <= 0.999 + 1e-9)
    x2 = bench.synth_episode(7, 4, 3, 64, 32)[0]
    assert torch.equal(x, x2)                                             # seeded


def test_bench_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "no CPU fallback" in out.stderr
    assert out.stdout.strip() == ""                                       # and prints no JSON line


def test_bench_quotes_the_newest_committed_traffic_summary():
    """bench.py looked for profiles/r02_conv_traffic.json while the current file was r02b_...; it now takes the
    newest rNN[letter]_<name> by round number, and the committed round-3 summary was measured on the headline episode."""
    import bench
    data, src = bench.newest_profile("conv_traffic.json")
    assert data is not None and "r06" in src, src
    assert data["episode"] == "metric_string" and 5e8 < data["hbm_bytes_per_launch"] < 2e9
    assert any(k.startswith("conv_gemm_split8_kernel") for k in data["kernels"])      # the round-4 kernels are counted
    assert bench.newest_profile("no_such_summary.json") == (None, None)


def test_roofline_block_quotes_the_peak_of_the_engine_the_kernel_issues_on():
    """`peak` / `frac` are against the instruction that runs -- under the split arithmetic the bf16 MFMA,
    six terms per fp32 product (2500 / 6 TFLOP/s of fp32 GEMM work); the fp32-MFMA figure stays as frac_of_fp32_mfma_peak."""
    import bench
    from fewshot_detection_amd.ops import PROFILE_CLASSES
    kp = {c: dict(ms=0.0, work=0.0, launches=0) for c in PROFILE_CLASSES}
    kp["gemm_fwd"] = dict(ms=10.0, work=1.3e12, launches=56)          # 130 TFLOP/s of fp32 GEMM work
    kp["gemm_wgrad"] = dict(ms=5.0, work=0.6e12, launches=28)
    kp["wino_transform"] = dict(ms=5.0, work=25e9, launches=100)
    r = dict(kp=kp, prof=[], prof_steps=1)
    for mode in ("split", "native"):
        roof = bench.roofline_block(r, "f32", 28.0, mode)
        peak = 2500.0 / 6.0 if mode == "split" else bench.PEAK_FP32_MFMA_TFLOPS
        assert roof["bound"] == "mfma" and abs(roof["peak"] - peak) < 1e-9
        assert abs(roof["achieved"] - 130.0) < 1e-6 and abs(roof["frac"] - 130.0 / peak) < 1e-9
        ar = roof["f32_gemm_arithmetic"]
        assert ar["mode"] == mode
        if mode == "split":
            assert abs(ar["issued_bf16_tflops"] - 780.0) < 1e-6
            assert abs(roof["frac_of_fp32_mfma_peak"] - 130.0 / 157.3) < 1e-9
            assert abs(roof["yardstick_frac"] - 130.0 / (bench.YARDSTICK_BF16_TFLOPS / 6.0)) < 1e-9
            assert abs(roof["wgrad_kernel"]["peak"] - peak) < 1e-9
        else:
            assert "issued_bf16_tflops" not in ar and "frac_of_fp32_mfma_peak" not in roof
        assert roof["hbm"]["unit"] == "GB/s" and abs(roof["hbm"]["achieved"] - 5000.0) < 1e-6


def _fat_result():
    """A result dict with EVERY extra populated (the structure main() builds), padded with round-3-sized prose."""
    import bench
    from fewshot_detection_amd.ops import PROFILE_CLASSES
    prose = "x" * 900
    kp = {c: dict(ms=3.0, work=2.0e10, launches=17) for c in PROFILE_CLASSES}
    kp["gemm_fwd"] = dict(ms=10.0, work=1.3e12, launches=56)
    kp["gemm_wgrad"] = dict(ms=5.0, work=0.6e12, launches=28)
    r = dict(kp=kp, prof=[], prof_steps=1)
    roof = bench.roofline_block(r, "f32", 28.0, "split")
    roof.update(traffic=749152106.4828, traffic_source="profiles/r04_conv_traffic.json (offline rocprofv3 PMC passes)",
                traffic_note=prose)
    parity = {"config": "B=64 queries 416x416 + N=20 supports 224x224, " + prose, "forward_max_abs_delta": 2.3e-4,
              "forward_max_abs": 3.77, "forward_rel_l2": 5.8e-5, "region_loss_end_to_end": {"hip": 6.1e5, "oracle": 6.1e5,
              "abs_delta": 0.19, "rel_delta": 3.07e-7}, "region_loss_abs_delta": 0.0, "region_loss_max_abs_delta": 8.2e-8,
              "region_loss_grad_max_abs": 3.5, "anchor_assignment_equal": True, "tolerance": 1e-3, "ok": True}
    cfgs_ = {k: {"what": prose, "ms_per_step": 29.4385123, "episodes_per_s": 33.9, "img_per_s": 2174.0,
                 "ms_each_step": [29.1] * 6, "episode_forward_gflop": 2022.0, "dtype": "f32"}
             for k in ("configs1_cfg_episode", "configs3_tuning_C4", "configs4_shape_C5")}
    inf = {"what": prose, "batch_2": {"eager_unfolded": 1.37, "eager_folded": 0.83, "graph_folded": 0.85, "kernels": 55,
                                      "img_per_s_graph": 2358.4}, "batch_32": {"eager_unfolded": 4.6, "eager_folded": 4.25,
                                                                                 "graph_folded": 4.31, "img_per_s_graph": 7428.2}}
    bb = {"what": prose, "algorithmic_gflop": 1210.0, "mfma_peak_tflops": 157.3, "note": prose,
          "train_bn": {"ms": 6.65, "img_per_s": 9620.0, "frac_of_mfma_peak_algorithmic": 1.156, "kernel_ms_by_class": dict.fromkeys(PROFILE_CLASSES, 1.0)},
          "eval_folded": {"ms": 6.4, "img_per_s": 10000.0, "frac_of_mfma_peak_algorithmic": 1.2}}
    also = dict(cfgs_)
    also.update({
        "sustained_run": {"what": prose, "ms_per_step": 27.445, "episodes_per_s": 36.4, "probe_mhz_after": 2218.0},
        "forward_only": {"what": prose, "ms": 9.066, "episodes_per_s": 110.0, "algorithmic_tflops": 214.0, "frac_of_mfma_peak_algorithmic": 1.36},
        "backbone_forward": bb, "backbone_forward_bf16": bb, "inference": {"f32": inf, "bf16": inf},
        "f32_gemm_native": {"what": prose, "ms_per_step": 31.87, "episodes_per_s": 31.4, "ms_per_step_unprofiled": 31.5,
                            "loss": 58000.0, "roofline": bench.roofline_block(r, "f32", 31.9, "native")},
        "bf16_mode": {"what": prose, "ms_per_step": 13.08, "episodes_per_s": 76.4, "img_per_s": 4892.0, "dtype": "bf16",
                      "ms_per_step_unprofiled": 12.8, "loss": 58011.0, "roofline": bench.roofline_block(r, "bf16", 13.1, "split"),
                      "other_configs": cfgs_, "parity": dict(parity, forward_rel_l2_vs_fp32_oracle=0.0744)}})
    return {
        "metric": "episodes/sec (64x416x416 query + 20x224x224 support) train step (fwd + RegionLoss + bwd + SGD)",
        "value": 35.8983123, "unit": "episodes/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 27.8565123,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "img_per_s": 2297.4897, "loss": 58042.4961,
        "config": {"workload": "the episode of BASELINE.json's metric string " + prose, "mode": "train", "global_batch": 64,
                   "parallelism": "dp1", "f32_gemm": "split", "episode_forward_gflop": 1941.5349},
        "roofline": roof,
        "gpu_clock": {"nominal_mhz": 2400.0, "probe_mhz_start": 2192.2, "waited_s": 0.0, "probe_mhz_after_timing": 2226.4,
                      "throttled": False, "what": prose},
        "streams": {"enabled": True, "what": prose, "profiled_steps_on_one_stream": 2, "profiled_step_index": [9, 11],
                    "ms_per_step_unprofiled": 27.57, "ms_per_step_profiled": 31.8},
        "dp": {"world_size": 1, "backend": None, "scaling": "weak", "gradient_buckets": 6, "allreduce_dtype": "float32",
               "bucket_mb": [49.3] * 6, "bucket_launch_order": [], "allreduce_wait_ms_per_step": [0.0] * 6, "overlap": None},
        "also_measured": also,
        "cpu_baseline": {"value": 0.0801666, "unit": "episodes/s", "cores": 64, "kind": "port", "sample": prose},
        "parity": parity,
    }


def test_final_line_fits_the_driver_tail(tmp_path, capsys):
    """VERDICT r3 (row d): round 3's JSON line was 23 KB, the driver keeps an ~8 KB stdout tail and parsed nothing.  The LAST
    stdout line is now a compact record < 4 KB whatever the extras hold; the verbose record goes to a file + stderr."""
    import io
    import json
    import bench
    res = _fat_result()
    assert len(json.dumps(res)) > 20000                                   # as fat as the line that broke round 3
    out = io.StringIO()
    line = bench.emit(res, out_dir=str(tmp_path), stream=out)
    stdout = out.getvalue()
    assert stdout.count("\n") == 1 and stdout.splitlines()[-1] == line
    assert len(line) < 4096 and len(line) < bench.COMPACT_LIMIT
    d = json.loads(stdout.splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == round(res["value"], 4) and d["ms_per_step"] == round(res["ms_per_step"], 4)
    assert d["config"]["workload"] and d["config"]["mode"] == "train"
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["peak"] - 2500.0 / 6.0) < 1e-3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["traffic"] > 0 and len(rf["kernel"]) <= 120
    assert "frac_of_fp32_mfma_peak" in rf and "yardstick_frac" in rf and "avg_kernel_ms" in rf and "launches_per_step" in rf
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] == 64 and d["cpu_baseline"]["kind"] == "port"
    assert d["parity"]["ok"] is True and d["parity"]["anchor_assignment_equal"] is True
    am = d["also_measured"]
    for k in ("bf16_ms", "bf16_frac", "backbone_f32_ms", "backbone_f32_frac_alg", "c1cfg_ms", "c4_ms", "c5_ms",
              "infer_b2_f32_ms", "infer_b2_f32_kernels", "native_ms"):
        assert k in am, k
    full = json.load(open(os.path.join(str(tmp_path), "bench_full_f32_n1.json")))
    assert full["also_measured"]["bf16_mode"]["roofline"]["kernel"]     # nothing is lost: the verbose record is on disk
    assert capsys.readouterr().out == ""                                  # emit() wrote to the stream it was given only


def test_main_keeps_module_chatter_off_stdout():
    """`class_scale 1` (RegionLossV2's constructor prints like the reference) and friends must not precede the JSON line on
    stdout: main() points sys.stdout at stderr for everything but emit()."""
    import inspect
    import bench
    src = inspect.getsource(bench.main)
    assert "sys.stdout = sys.stdout, sys.stderr" in src.replace("real_stdout, ", "") or "sys.stdout = sys.stderr" in src or \
        "real_stdout, sys.stdout = sys.stdout, sys.stderr" in src
    assert "emit(res, stream=real_stdout)" in inspect.getsource(bench._main)


def test_every_conv_family_kernel_of_the_newest_profiles_is_classified():
    """tools/pmc_traffic.py filtered by a literal name list and silently dropped kernels after a rename
    (`wino4_output4_kernel<32>`: 7.85 GB per step).  Every kernel spelled conv_* / wino* / wgrad* in the newest committed
    kernel-stats summaries must fall into a class, and the bracketed ("conv_launch") set must hold the renamed ones."""
    import csv
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic as pt
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.csv")))
    assert files
    newest = files[-6:]
    seen = set()
    for f in newest:
        for r in csv.DictReader(open(f)):
            name = r.get("Name") or r.get("Kernel_Name")
            if pt.is_conv_family(name):
                seen.add(pt.short_name(name))
                assert pt.classify(name) is not None, (os.path.basename(f), pt.short_name(name))
    assert any(s.startswith("conv_gemm_kernel") for s in seen)
    assert pt.classify("void (anonymous namespace)::wino4_output4_kernel<32>(float const*, float const*)") == "conv_launch"
    assert pt.classify("void (anonymous namespace)::wino4_dy_kernel<2>(float*, long long)") == "grad_transform"
    assert pt.classify('"conv_gemm_kernel<128, 128, 2, 2, false, 1, true, false, true>"') == "conv_launch"
    assert pt.classify("(anonymous namespace)::sgd_kernel(float*, float const*)") is None


def test_bench_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 8` (WORLD_SIZE unset) re-executes itself under torch.distributed.run on 127.0.0.1 and passes
    rank 0's JSON line through as the last stdout line (no GPU needed to check the command and the plumbing)."""
    import subprocess
    import bench
    seen = {}

    class FakeProc(object):
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(["[gloo] rank 0 connected\n", '{"metric":"episodes/sec","value":1.0}\n'])

        def wait(self):
            return 0
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    out = io.StringIO()
    monkeypatch.setattr(sys, "stdout", out)
    rc = bench.self_launch(8)
    assert rc == 0 and out.getvalue().splitlines()[-1] == '{"metric":"episodes/sec","value":1.0}'
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_committed_bench_line_of_this_round_has_no_device_allocation_in_its_timed_region():
    """a hipMalloc inside the timed region drains the device.  The committed contract line of round 6 onward
    (profiles/rNN_bench_line_f32.json = the LAST stdout line of `python bench.py --gpus 1 --steps 20 --warmup 5` on the GPU box)
    must carry allocator.device_allocs_in_timed_region == 0 next to the contract fields."""
    import glob
    import json
    import re
    lines = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line_f32.json"))
                   if int(re.match(r"r(\d+)", os.path.basename(p)).group(1)) >= 6)
    if not lines:
        pytest.skip("no bench line of round >= 6 committed yet")
    d = json.load(open(lines[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline", "cpu_baseline", "parity", "allocator"):
        assert k in d, k
    assert d["allocator"]["device_allocs_in_timed_region"] == 0, d["allocator"]
    assert d["parity"]["ok"] is True and d["parity"].get("other_shapes_ok") is True
    assert d["roofline"]["bound"] == "mfma" and 0.0 < d["roofline"]["frac"] < 1.0 and d["cpu_baseline"]["kind"] == "port"
