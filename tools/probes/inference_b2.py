"""valid_ensemble.py's batch shape for a rocprofv3 kernel trace: 50 eval-mode detect_forward calls on 2 images 416x416 with
20 ensembled reweighting vectors, inference form (BatchNorm folded), eager launches (a hipGraph replay shows up as one
graph launch in the trace).  python tools/probes/inference_b2.py [f32|bf16]"""
import sys
import tempfile

import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import cfgs  # noqa: E402
from fewshot_detection_amd.darknet_meta import Darknet  # noqa: E402

dev = torch.device("cuda:0")
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
torch.manual_seed(0)
net = Darknet(dyn_cfg, rw_cfg).to(dev).eval().set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "f32")
vecs = [torch.rand(20, 1024, 1, 1, device=dev)]
x = torch.rand(2, 3, 416, 416, device=dev)
with torch.no_grad():
    for _ in range(55):
        net.detect_forward(x, vecs)
torch.cuda.synchronize()
