"""Which python call sites issue device-to-device tensor copies in a bf16-mode training step?"""
import collections, os, sys, tempfile, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fewshot_detection_amd import cfgs
from fewshot_detection_amd.cfg import cfg

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
cfg.neg_ratio = 1
tmp = tempfile.mkdtemp()
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tmp)
leg = bench.Leg(dyn_cfg, rw_cfg, dt, dev, None, 16, "train")
x, metax, mask, target = bench.synth_episode(1000, 16, 20, 416, 224)
step = leg.stepper(x.to(dev).contiguous(), metax.to(dev), mask.to(dev), target)
for _ in range(3):
    step()
torch.cuda.synchronize()
sites = collections.Counter()
orig = {}
def wrap(name):
    f = getattr(torch.Tensor, name)
    orig[name] = f
    def g(self, *a, **k):
        if self.is_cuda:
            st = traceback.extract_stack(limit=4)[:-1]
            sites[(name, tuple("%s:%d" % (os.path.basename(s.filename), s.lineno) for s in st[-2:]), tuple(self.shape))] += 1
        return f(self, *a, **k)
    setattr(torch.Tensor, name, g)
for n in ("copy_", "clone", "contiguous", "to", "float", "zero_", "fill_"):
    wrap(n)
step()
torch.cuda.synchronize()
for n, f in orig.items():
    setattr(torch.Tensor, n, f)
for k, v in sites.most_common(40):
    print(v, k)
