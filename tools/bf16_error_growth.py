"""Where does the bf16 storage mode's end-to-end error come from?  

darknet_dynamic.cfg + reweighting_net.cfg at random initialisation, B = 8 queries 416x416, N = 5 supports 224x224, train-mode
BatchNorm.  After every conv block (conv + BatchNorm + leaky, and the max pool that follows it) three runs are compared:

    fp32    the oracle's fp32 forward (oracle/net.py::_walk) -- the reference's arithmetic
    walk    the oracle's restatement of the bf16 storage mode (oracle/net.py::_conv_block_bf16: bf16-rounded operands and
            stored tensors, fp32 accumulation, statistics from the unrounded conv output) -- no HIP kernel involved
    hip     the product's bf16 mode (free-running: every layer consumes the previous HIP layer's output)

Columns (relative L2): hip vs fp32, walk vs fp32 (what the storage mode ITSELF costs at this depth), hip vs walk (kernels vs
their definition: summation order and rounding-boundary flips, amplified by the layers in between).  If `hip vs fp32` tracks
`walk vs fp32` layer by layer, the 0.23 at the head is accumulated storage rounding through a randomly initialised net, not a
kernel.  The detector rows use the fp32 oracle's reweighting vectors in all three runs; the reweighting net has its own rows;
the last row is end to end (each run with its own vectors).

    python tools/bf16_error_growth.py [--out profiles/r06_bf16_error_growth]     (needs the MI355X and the oracle)
"""
import argparse
import json
import os
import sys
import tempfile

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B, N, S, SM, SEED = 8, 5, 416, 224, 606


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def oracle_walk(blocks, mods, x, dyn, mode):
    """The oracle's layer walk (oracle/net.py::_walk / _walk_bf16) with the output of every conv block captured AFTER the
    max pool that follows it (the tensor the next layer reads).  -> (result, {layer index or 'head': tensor})"""
    from oracle import net as onet
    outs, caps = {}, {}
    layers = blocks[1:]
    skip = -1
    for idx, blk in enumerate(layers):
        if idx <= skip:
            continue
        kind = blk["type"]
        if kind == "route":
            src = [int(v) if int(v) > 0 else int(v) + idx for v in blk["layers"].split(",")]
            x = outs[src[0]] if len(src) == 1 else torch.cat([outs[s] for s in src], 1)
        elif kind in ("region", "cost"):
            continue
        elif kind == "convolutional" and onet.is_dynamic(blk):
            head = mods[idx + 1][0]
            if mode == "f32":
                x = head(onet.reweight(x, dyn))
            else:                                   # as the product fuses it: the folded weight rounded once
                n_cls, o_ch, c = dyn.shape[0], head.weight.shape[0], head.weight.shape[1]
                w_eff = onet._q((head.weight.view(1, o_ch, c) * dyn.view(n_cls, 1, c)).reshape(n_cls * o_ch, c, 1, 1))
                bias = None if head.bias is None else head.bias.repeat(n_cls)
                y = F.conv2d(x, w_eff, bias)
                x = y.view(x.shape[0] * n_cls, o_ch, x.shape[2], x.shape[3])
            caps["head"] = x
            skip = idx + 1
            outs[idx + 1] = x
            continue
        elif kind == "convolutional":
            x = mods[idx](x) if mode == "f32" else onet._conv_block_bf16(mods[idx], x, True)
            caps[idx] = x
        else:
            x = mods[idx](x)
            if kind == "maxpool" and (idx - 1) in caps:
                caps[idx - 1] = x                    # the pooled tensor is what the next layer reads
            if kind == "globalmax":
                caps["vectors"] = x
        outs[idx] = x
    return x, caps


def hip_walk(engine, inputs, dyn):
    """The product's engine on the device, free-running; the same capture points from its tape."""
    from fewshot_detection_amd import ops
    res, tape = engine.forward(inputs, dyn=dyn, training=True, record=False)
    caps = {}
    for rec in tape:
        if rec["kind"] == "conv":
            z = rec["z"]
            t = z.t[:, z.c0:z.c0 + z.C].float().reshape(z.B, z.H, z.W, z.C).permute(0, 3, 1, 2).contiguous().cpu()
            caps[rec["ind"]] = t
    return res.detach().cpu(), caps


def measure(dev):
    from fewshot_detection_amd import cfgs
    from fewshot_detection_amd.darknet_meta import Darknet
    from oracle.net import OracleDarknet
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
    torch.manual_seed(SEED)
    ora = OracleDarknet(dyn_cfg, rw_cfg).train()
    state = {k: v.clone() for k, v in ora.state_dict().items()}
    net = Darknet(dyn_cfg, rw_cfg)
    net.load_state_dict(state)
    net = net.to(dev).train().set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(SEED + 1)
    x, metax = torch.rand(B, 3, S, S, generator=g), torch.rand(N, 3, SM, SM, generator=g)
    mask = torch.zeros(N, 1, SM, SM)
    mask[:, :, 40:160, 30:170] = 1
    rows = []
    with torch.no_grad():
        m_in = torch.cat([metax, mask], 1)
        # ---- reweighting net ----
        v32, c32 = oracle_walk(ora.learnet_blocks, ora.learnet_models, m_in, None, "f32")
        ora.load_state_dict(state)
        v16, c16 = oracle_walk(ora.learnet_blocks, ora.learnet_models, m_in, None, "bf16")
        ora.load_state_dict(state)
        vh, ch = hip_walk(net._meta, [metax.to(dev), mask.to(dev)], None)
        for k in sorted(k for k in c32 if isinstance(k, int)):
            rows.append(("reweighting net layer %d" % k, rel(ch[k], c32[k]), rel(c16[k], c32[k]), rel(ch[k], c16[k])))
        rows.append(("reweighting vectors", rel(vh, v32), rel(v16, v32), rel(vh, v16)))
        # ---- detector, the fp32 oracle's vectors in all three runs ----
        o32, d32 = oracle_walk(ora.blocks, ora.models, x, v32, "f32")
        ora.load_state_dict(state)
        o16, d16 = oracle_walk(ora.blocks, ora.models, x, v32, "bf16")
        ora.load_state_dict(state)
        net.load_state_dict(state)
        oh, dh = hip_walk(net._det, [x.to(dev)], [v32.to(dev)])
        for k in sorted(k for k in d32 if isinstance(k, int)):
            rows.append(("detector layer %d" % k, rel(dh[k], d32[k]), rel(d16[k], d32[k]), rel(dh[k], d16[k])))
        rows.append(("head output (common vectors)", rel(oh, o32), rel(o16, o32), rel(oh, o16)))
        # ---- end to end: every run with its own vectors ----
        e16, _ = oracle_walk(ora.blocks, ora.models, x, v16, "bf16")
        ora.load_state_dict(state)
        net.load_state_dict(state)
        eh, _ = hip_walk(net._det, [x.to(dev)], [vh.to(dev)])
        rows.append(("head output (end to end)", rel(eh, o32), rel(e16, o32), rel(eh, e16)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_bf16_error_growth"))
    a = ap.parse_args()
    rows = measure(torch.device("cuda:0"))
    with open(a.out + ".json", "w") as f:
        json.dump({"config": {"B": B, "N": N, "size": S, "support": SM, "seed": SEED, "bn": "train"},
                   "columns": ["hip_vs_fp32", "walk_vs_fp32", "hip_vs_walk"],
                   "rows": [{"layer": r[0], "hip_vs_fp32": r[1], "walk_vs_fp32": r[2], "hip_vs_walk": r[3]} for r in rows]}, f,
                  indent=1)
    with open(a.out + ".md", "w") as f:
        f.write("# bf16 storage mode: relative-L2 error after every block (tools/bf16_error_growth.py)\n\n")
        f.write("darknet_dynamic.cfg + reweighting_net.cfg, random init (seed %d), B=%d queries %dx%d, N=%d supports %dx%d, "
                "train-mode BatchNorm.\n`fp32` = oracle fp32 forward; `walk` = oracle/net.py::_conv_block_bf16 (the mode's definition, "
                "CPU); `hip` = the product, free-running.\n\n" % (SEED, B, S, S, N, SM, SM))
        f.write("| tensor | hip vs fp32 | walk vs fp32 | hip vs walk |\n|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %.3e | %.3e | %.3e |\n" % r)
    for r in rows:
        print("%-34s hip/fp32 %.3e   walk/fp32 %.3e   hip/walk %.3e" % r)


if __name__ == "__main__":
    main()
