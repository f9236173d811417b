"""Inference (eval-mode detect_forward with fixed reweighting vectors + decode + NMS) time per batch at small batch sizes:
is it launch-bound?  python tools/probes/inference_time.py"""
import sys, time, tempfile
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import cfgs, utils
from fewshot_detection_amd.darknet_meta import Darknet
dev = torch.device("cuda:0")
d = tempfile.mkdtemp()
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(d)
torch.manual_seed(0)
net = Darknet(dyn_cfg, rw_cfg).to(dev).eval()
N = 20
vecs = [torch.rand(N, 1024, 1, 1, device=dev)]
import os
for dtype, graphs in (("f32", False), ("f32", True), ("bf16", False), ("bf16", True)):
    net.set_compute_dtype(dtype)
    net.inference_graphs = graphs
    for B in (1, 2, 8, 32):
        x = torch.rand(B, 3, 416, 416, device=dev)
        def fwd():
            with torch.no_grad():
                return net.detect_forward(x, vecs)
        for _ in range(3): fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): out = fwd()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # GPU-only time via events
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fwd()
        e1.record(); torch.cuda.synchronize()
        print(("graph " if graphs else "eager ") + "%s B=%2d: host enqueue %.2f ms/batch, wall %.2f ms/batch (%.0f img/s), stream time %.2f ms/batch" % (
            dtype, B, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, B * 20 / (t2 - t0), e0.elapsed_time(e1) / 20))
