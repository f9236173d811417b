"""Experiment (round 6): does giving the side streams a CU MASK make the MFMA-bound and the HBM-bound strands of the step
overlap?  Today the sum of the kernels' one-stream durations equals the step time (24.9 vs 25.1 ms): the weight-gradient GEMMs
(side stream) and the data-gradient chain's HBM-bound transforms (main stream) do not run beside each other, because a GEMM
launch fills every CU to its register / LDS limit and the other queue's workgroups only get the slots it frees.

hipExtStreamCreateWithCUMask pins a stream's kernels to a subset of the CUs; the main stream keeps all of them.  The probe runs the
headline train step with the "wgrad" (and optionally "meta") stream masked to several CU subsets and prints ms/step.

    python tools/experiments_r06/cu_mask_probe.py [f32|bf16]
"""
import ctypes
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
from fewshot_detection_amd import cfgs, streams  # noqa: E402
from fewshot_detection_amd.cfg import cfg  # noqa: E402

HIP = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    """A torch stream whose kernels may only run on the CUs whose bits are set in `words` (8 x uint32 = 256 CUs)."""
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = HIP.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    return torch.cuda.ExternalStream(st.value, device=0)


def timed(step, n=12, w=4):
    for _ in range(w):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


MASKS = [
    ("none (all 256 CUs)", None),
    ("every 2nd CU (128)", [0x55555555] * 8),
    ("3 of 4 CUs (192)", [0x77777777] * 8),
    ("1 of 4 CUs (64)", [0x11111111] * 8),
    ("first 128 bits", [0xffffffff] * 4 + [0] * 4),
    ("first 64 bits", [0xffffffff] * 2 + [0] * 6),
    ("bits 0-15 of every word (128)", [0x0000ffff] * 8),
]


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg.neg_ratio = 1
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
    sys.stdout, real = sys.stderr, sys.stdout
    leg = bench.Leg(dyn_cfg, rw_cfg, dtype, dev, None, 64, "train")
    x, metax, mask, target = bench.synth_episode(1000, 64, 20, 416, 224)
    x, metax, mask = x.to(dev).contiguous(), metax.to(dev), mask.to(dev)
    step = leg.stepper(x, metax, mask, target)
    sys.stdout = real
    streams.ENABLED = False
    print("%s one stream: %.3f ms" % (dtype, timed(step)), flush=True)
    streams.ENABLED = True
    for which in (("wgrad",), ("wgrad", "meta")):
        for name, words in MASKS:
            streams.reset()
            if words is not None:
                for w_ in which:
                    streams._SIDE[(0, w_)] = masked_stream(words)
            ms = [timed(step, n=10, w=4) for _ in range(2)]
            print("%s masked %-12s %-32s %.3f / %.3f ms" % (dtype, "+".join(which), name, ms[0], ms[1]), flush=True)
            if words is None and which != ("wgrad",):
                continue
    streams.reset()


if __name__ == "__main__":
    main()
