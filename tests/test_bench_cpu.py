"""bench.py's host-side arithmetic: the algorithmic FLOP counts behind `roofline.achieved` are the SURVEY 8(d) /
BASELINE.md figures, the synthetic episode has the contract's shapes, and without a GPU the bench fails loudly."""
import os
import subprocess
import sys

import numpy as np
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
    assert data is not None and "r03" in src, src
    assert data["episode"] == "metric_string" and 5e8 < data["hbm_bytes_per_launch"] < 2e9
    assert bench.newest_profile("no_such_summary.json") == (None, None)


def test_roofline_block_quotes_the_fp32_peak_and_names_the_arithmetic():
    """dtype f32: `peak` / `frac` stay the dense fp32 MFMA figures the contract asks for, whatever instruction runs; under the
    split arithmetic the block also carries the same achieved figure against the bf16 instruction's ceiling."""
    import bench
    from fewshot_detection_amd.ops import PROFILE_CLASSES
    kp = {c: dict(ms=0.0, work=0.0, launches=0) for c in PROFILE_CLASSES}
    kp["gemm_fwd"] = dict(ms=10.0, work=1.3e12, launches=56)          # 130 TFLOP/s of fp32 GEMM work
    kp["gemm_wgrad"] = dict(ms=5.0, work=0.6e12, launches=28)
    kp["wino_transform"] = dict(ms=5.0, work=25e9, launches=100)
    r = dict(kp=kp, prof=[], prof_steps=1)
    for mode in ("split", "native"):
        roof = bench.roofline_block(r, "f32", 28.0, mode)
        assert roof["bound"] == "mfma" and roof["peak"] == bench.PEAK_FP32_MFMA_TFLOPS
        assert abs(roof["achieved"] - 130.0) < 1e-6 and abs(roof["frac"] - 130.0 / 157.3) < 1e-9
        ar = roof["f32_gemm_arithmetic"]
        assert ar["mode"] == mode
        if mode == "split":
            assert abs(ar["issued_bf16_tflops"] - 780.0) < 1e-6
            assert abs(ar["frac_of_fp32_equivalent_peak"] - 130.0 / (2500.0 / 6.0)) < 1e-9
        else:
            assert "issued_bf16_tflops" not in ar
        assert roof["hbm"]["unit"] == "GB/s" and abs(roof["hbm"]["achieved"] - 5000.0) < 1e-6
