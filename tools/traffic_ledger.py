"""Per-layer HBM byte ledger of the convolution launches (fp32 mode, the headline configuration): for every 3x3 / 1x1 layer
shape of the detector at B = 64 and each direction (forward, data gradient, weight gradient), the bytes the launch MUST move
(operands once, result once) against the bytes its kernels really moved (rocprofv3 PMC: FETCH_SIZE / WRITE_SIZE).

    # on the GPU box, one pass per counter (never combined with other trace domains):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o run -- python tools/traffic_ledger.py run
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -o run -- python tools/traffic_ledger.py run
    # anywhere:
    python tools/traffic_ledger.py table out/fetch out/write > profiles/r05_traffic_ledger.csv

`run` executes every (layer, direction) segment REPS times through fewshot_detection_amd.ops exactly as the engine launches it
(F(4x4) Winograd pipeline, halo kernel or implicit GEMM -- whatever ops.wino_tile / the library picks for the shape) and
brackets each segment with a marker kernel (a torch erfinv_ whose element count encodes the segment number), so that `table` can
cut the counter file by dispatch order without knowing which kernels a path consists of.
Units: FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md section HBM)."""
import collections
import csv
import glob
import os
import re
import sys

SHAPES = [  # H (= W), cin, cout, k   at B = 64: the 3x3 / 1x1 layers of darknet_dynamic.cfg behind the first layer
    (208, 32, 64, 3), (104, 64, 128, 3), (104, 128, 64, 1), (52, 128, 256, 3), (52, 256, 128, 1), (26, 256, 512, 3),
    (26, 512, 256, 1), (13, 512, 1024, 3), (13, 1024, 512, 1), (13, 1024, 1024, 3), (13, 1280, 1024, 3),
]
DIRS = ("fwd", "dgrad", "wgrad")
B = 64
REPS = 3
MARK = 1 << 18          # marker s = an in-place erfinv over (s + 1) * MARK floats (no other kernel of the run has that name)


def segments():
    return [(s, d) for s in SHAPES for d in DIRS]


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fewshot_detection_amd import ops
    dev = torch.device("cuda:0")
    marks = torch.zeros(MARK * (len(segments()) + 2), device=dev)
    torch.manual_seed(0)
    discard = lambda: marks[:MARK * (len(segments()) + 1)].erfinv_()     # noqa: E731  (what follows belongs to no segment)
    discard()
    for si, ((H, cin, cout, k), d) in enumerate(segments()):
        x = ops.nchw_to_nhwc(torch.randn(B, cin, H, H, device=dev))
        dy = ops.nchw_to_nhwc(torch.randn(B, cout, H, H, device=dev))
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        if d == "fwd":
            tile = ops.wino_tile(cin, cout, k, H, H)
            wp = ops.pack_weight_wino(w, 0, tile) if tile else ops.pack_weight(w)
            fn = (lambda: ops.conv3x3_wino(x, wp, cout, tile=tile, bn_partial=True)) if tile else \
                (lambda: ops.conv2d(x, wp, cout, k, bn_partial=True))
        elif d == "dgrad":
            tile = ops.wino_tile(cout, cin, k, H, H)
            wp = ops.pack_weight_wino(w, 1, tile) if tile else ops.pack_weight(w, 1)
            fn = (lambda: ops.conv3x3_wino(dy, wp, cin, tile=tile)) if tile else (lambda: ops.conv2d(dy, wp, cin, k))
        else:
            fn = lambda: ops.conv2d_wgrad(dy, cout, x, cin, k)      # noqa: E731
        fn()                                    # warm (allocations, plan caches) outside the segment
        torch.cuda.synchronize()
        marks[:MARK * (si + 1)].erfinv_()       # segment marker
        for _ in range(REPS):
            fn()
        discard()
        torch.cuda.synchronize()


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name.strip('"'))
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def read_pass(path, counter):
    """-> {segment index: {kernel: KiB summed over the segment's REPS}} from one counter_collection.csv"""
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + path
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows.sort()
    fills = sorted({g for _, n, g, _ in rows if "erfinv" in n})
    # marker grids grow with the segment number: the i-th smallest marker grid = segment i
    seg_of = {g: i for i, g in enumerate(fills)}
    out, cur = collections.defaultdict(lambda: collections.defaultdict(float)), None
    for _, n, g, v in rows:
        if "erfinv" in n:
            cur = seg_of[g]
            continue
        if cur is not None and cur < len(segments()):
            out[cur][n] += v
    return out


def table(fetch_dir, write_dir):
    fe, wr = read_pass(fetch_dir, "FETCH_SIZE"), read_pass(write_dir, "WRITE_SIZE")
    print("layer,dir,algorithmic_MB,wino_unfused_floor_MB,hbm_MB_per_launch,ratio_to_algorithmic,ratio_to_floor,kernels (fetch+write MB each)")
    tot_alg = tot_hbm = 0.0
    for si, ((H, cin, cout, k), d) in enumerate(segments()):
        px = B * H * H
        xb, yb, wb = 4.0 * px * cin, 4.0 * px * cout, 4.0 * k * k * cin * cout
        alg = xb + yb + wb                                          # every direction reads two of the three and writes the third
        pad = ((H + 3) // 4 * 4) ** 2 / float(H * H)
        src, dst = (xb, yb) if d == "fwd" else (yb, xb)
        # unfused F(4x4): transform planes V (36/16 of the padded source) and M (36/16 of the padded result), each written once
        # and read once, + the 36/9-fold weight planes
        floor = src + 2 * 2.25 * pad * src + 2 * 2.25 * pad * dst + dst + 4.0 * wb if (k == 3 and d != "wgrad") else alg
        kern = {}
        for n, v in fe.get(si, {}).items():
            kern[n] = kern.get(n, 0.0) + 2.0 * v * 1024 / REPS
        for n, v in wr.get(si, {}).items():
            kern[n] = kern.get(n, 0.0) + v * 1024 / REPS
        hbm = sum(kern.values())
        tot_alg += alg
        tot_hbm += hbm
        ks = "; ".join("%s %.0f" % (n[:48], v / 1e6) for n, v in sorted(kern.items(), key=lambda kv: -kv[1]) if v > 1e6)
        print("%dx%d %d->%d k%d,%s,%.1f,%.1f,%.1f,%.2f,%.2f,\"%s\"" % (H, H, cin, cout, k, d, alg / 1e6, floor / 1e6, hbm / 1e6,
                                                                    hbm / alg, hbm / floor, ks))
    print("total,,%.1f,,%.1f,%.2f,," % (tot_alg / 1e6, tot_hbm / 1e6, tot_hbm / tot_alg))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) == 4 and sys.argv[1] == "table":
        table(sys.argv[2], sys.argv[3])
    else:
        sys.exit(__doc__)
