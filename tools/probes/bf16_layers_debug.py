import os, sys, tempfile, torch
sys.path.insert(0, "/root/repo")
from fewshot_detection_amd import cfgs, ops
from fewshot_detection_amd.darknet_meta import Darknet
from oracle import net as onet
from oracle.net import OracleDarknet
torch.set_num_threads(64)
d = tempfile.mkdtemp()
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(d)
torch.manual_seed(31)
ora = OracleDarknet(dyn_cfg, rw_cfg).train()
net = Darknet(dyn_cfg, rw_cfg)
net.load_state_dict(ora.state_dict())
dev = torch.device("cuda:0")
net = net.to(dev).train().set_compute_dtype("bf16")
B, S = 2, 416
g = torch.Generator().manual_seed(32)
x = torch.rand(B, 3, S, S, generator=g)
dyn = torch.rand(3, 1024, 1, 1, generator=g)
with torch.no_grad():
    out, tape = net._det.forward([x.to(dev)], dyn=[dyn.to(dev)], training=True, record=False)
    # oracle walk with captured outputs
    outs = {}
    blocks, mods = ora.blocks, ora.models
    xx = x
    for idx, blk in enumerate(blocks[1:]):
        kind = blk["type"]
        if kind == "route":
            src = [int(v) if int(v) > 0 else int(v) + idx for v in blk["layers"].split(",")]
            xx = outs[src[0]] if len(src) == 1 else torch.cat([outs[s] for s in src], 1)
        elif kind in ("region", "cost"):
            continue
        elif kind == "convolutional" and onet.is_dynamic(blk):
            break
        elif kind == "convolutional":
            xx = onet._conv_block_bf16(mods[idx], xx, True)
        else:
            xx = mods[idx](xx)
        outs[idx] = xx
recs = [r for r in tape if r["kind"] == "conv"]
for r in recs:
    ind = r["ind"]
    z = r["z"]
    zz = z.t[:, z.c0:z.c0 + z.C].float().reshape(z.B, z.H, z.W, z.C).permute(0, 3, 1, 2).cpu()
    ref = outs[ind + 1] if r["pool"] else outs[ind]
    if ref.shape != zz.shape:
        print(ind, "shape", tuple(zz.shape), tuple(ref.shape)); continue
    print("layer %2d pool %d C %4d: rel L2 %.3e  max|d| %.3e" % (ind, r["pool"], z.C, float((zz - ref).norm() / ref.norm()), float((zz - ref).abs().max())))
