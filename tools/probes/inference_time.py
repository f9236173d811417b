"""Eval-mode detect_forward latency at valid_ensemble.py's batch shape (2 images 416x416, 20 ensembled vectors): mean of 200
calls, eager and hipGraph replay.  python tools/probes/inference_time.py [batch]"""
import sys
import tempfile
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import cfgs, ops  # noqa: E402
from fewshot_detection_amd.darknet_meta import Darknet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
torch.manual_seed(0)
net = Darknet(dyn_cfg, rw_cfg).to(dev).eval()
vecs = [torch.rand(20, 1024, 1, 1, device=dev)]
x = torch.rand(B, 3, 416, 416, device=dev)
for graphs in (False, True):
    net.inference_graphs = graphs
    with torch.no_grad():
        for _ in range(10):
            net.detect_forward(x, vecs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            net.detect_forward(x, vecs)
        torch.cuda.synchronize()
    print("B=%d mode=%s graphs=%d: %.3f ms" % (B, ops.f32_gemm_mode(), graphs, (time.perf_counter() - t0) / 200 * 1e3), flush=True)
