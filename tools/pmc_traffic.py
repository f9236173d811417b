"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) into HBM bytes per conv launch of bench.py.

A "conv launch" is what bench.py brackets with HIP events (ops.PROFILE: conv2d, conv3x3_wino, conv3x3_c4 -- forward and
data-gradient launches): one direct implicit-GEMM kernel, or the kernels of a Winograd pipeline (input transform, the
36 / 16 batched position GEMMs, output transform), or the fused Winograd kernel.  Kernels are CLASSIFIED by name
(classify() below; tests/test_bench_cpu.py checks that every conv_* / wino* kernel of the newest committed kernel-stats
summary falls into a class -- round 3's literal name list silently dropped `wino4_output4_kernel<32>` after a rename and
under-counted the traffic by 135 MB per launch).  Reported:
  hbm_bytes_per_launch                      the kernels inside the brackets                                  [roofline.traffic]
  hbm_bytes_per_launch_with_grad_transforms + the weight-gradient side's dy transforms (wino*_dy*, wino4_grad), which
                                              write the second operand of the SAME layers' backward
FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md section HBM); both counters are in KiB.

Usage: python tools/pmc_traffic.py <fetch_prefix> <write_prefix> <steps_profiled | 0 = count them> <conv_launches_per_step> <out.json> [episode]
episode: "metric_string" (64 queries 416x416 + 20 supports 224x224, the headline) or "configs1" -- bench.py only quotes a
traffic file on the episode it was measured on."""
import csv
import json
import re
import sys

# (class, regex on the demangled kernel name without namespace / return type)
CLASSES = (
    ("conv_launch", r"^(conv_gemm_kernel|conv_gemm_split8_kernel|conv3x3_halo_kernel|conv3x3_halo_h_kernel|conv_gemm_bf16_kernel|conv_bf16_dma_kernel|conv_bf16_kernel|conv_first_kernel|conv_first_split_kernel|"
                    r"wino_input_kernel|wino_output_kernel|wino4_input_kernel|wino4_output_kernel|wino4_output4_kernel|"
                    r"wino4_fused_kernel|wino4_fused_[a-z0-9_]*kernel|wino4_gemm_out_kernel|wino4_rowfused_kernel|wino4_input_planes_kernel|wino4_split_planes_kernel)\b"),
    ("grad_transform", r"^(wino_dy_kernel|wino4_dy_kernel|wino4_grad_kernel)\b"),
    ("wgrad", r"^(wgrad_kernel|wgrad_split8_kernel|wgrad3x3_halo_kernel|wgrad3x3_halo_h_kernel|wgrad_first_kernel|wgrad_reduce_kernel|wgrad_bf16_tr_kernel|wgrad_bf16_tr8_kernel|"
              r"wgrad_bf16_kernel|wgrad_fold_h_kernel|wgrad_h_fold_kernel|wgrad_h_partial_kernel|wino_dw_kernel|wino4_dw_kernel|conv_wgrad_[a-z0-9_]*kernel)\b"),
    ("weight_pack", r"^(wino_weight_kernel|wino4_weight_kernel|wino4_weight_wide_kernel|wino4_weight_split_kernel|pack_weight_kernel|"
                    r"pack_weight_bf16_kernel|pack_weight_bf16_pair_kernel|pack_weight_split_kernel|sgd_pack_multi_kernel)\b"),
)


def short_name(kernel_name):
    """'void (anonymous namespace)::wino4_output4_kernel<32>(float const*, ...)' -> 'wino4_output4_kernel<32>'"""
    s = kernel_name.strip().strip('"')
    s = re.sub(r"^void\s+", "", s)
    s = s.replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in s:                      # cut the argument list: the first '(' at template depth 0
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def classify(kernel_name):
    """-> one of the CLASSES names, or None for kernels that are not part of a convolution's pipeline."""
    s = short_name(kernel_name)
    for cls, rx in CLASSES:
        if re.match(rx, s):
            return cls
    return None


def is_conv_family(kernel_name):
    """Names that MUST be classified (the CPU test's net): anything spelled conv_*, wino*, wgrad*."""
    return re.match(r"^(conv_|wino|wgrad)", short_name(kernel_name)) is not None


def totals(prefix, counter):
    tot, n, names = {}, {}, {}
    for r in csv.DictReader(open(prefix + "_counter_collection.csv")):
        if r["Counter_Name"] != counter:
            continue
        cls = classify(r["Kernel_Name"])
        if cls is None:
            continue
        tot[cls] = tot.get(cls, 0.0) + float(r["Counter_Value"])
        n[cls] = n.get(cls, 0) + 1
        names.setdefault(cls, set()).add(short_name(r["Kernel_Name"]))
    return tot, n, names


def steps_in(prefix, counter):
    """Train steps a pass really ran (warm-up, bench.py's untimed settle steps and the timed ones): one region_finalize_kernel
    launch per step."""
    return sum(1 for r in csv.DictReader(open(prefix + "_counter_collection.csv"))
               if r["Counter_Name"] == counter and "region_finalize_kernel" in r["Kernel_Name"])


def main():
    fetch_prefix, write_prefix, steps, per_step, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    episode = sys.argv[6] if len(sys.argv) > 6 else "metric_string"
    if steps <= 0:                       # 0: count them
        steps = steps_in(fetch_prefix, "FETCH_SIZE")
        if steps != steps_in(write_prefix, "WRITE_SIZE") or steps < 1:
            raise SystemExit("the two passes ran different numbers of steps")
    f, nf, names = totals(fetch_prefix, "FETCH_SIZE")
    w, nw, _ = totals(write_prefix, "WRITE_SIZE")
    launches = steps * per_step

    def per_launch(classes):
        fb = sum(2.0 * f.get(c, 0.0) * 1024 for c in classes) / launches
        wb = sum(w.get(c, 0.0) * 1024 for c in classes) / launches
        return fb, wb
    fb, wb = per_launch(("conv_launch",))
    fg, wg = per_launch(("conv_launch", "grad_transform"))
    res = {"kernels": sorted(names.get("conv_launch", ())), "grad_transform_kernels": sorted(names.get("grad_transform", ())),
           "kernel_dispatches_fetch_pass": nf.get("conv_launch", 0), "kernel_dispatches_write_pass": nw.get("conv_launch", 0),
           "conv_launches": launches, "fetch_bytes_per_launch_x2": fb, "write_bytes_per_launch": wb,
           "hbm_bytes_per_launch": fb + wb, "hbm_bytes_per_launch_with_grad_transforms": fg + wg, "episode": episode}
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --streams 0 --steps 3 --warmup 1` "
                   "(%d train steps per pass with the untimed settle steps, counted from the region_finalize_kernel launches); "
                   "FETCH_SIZE x2 per MI355X_MICROARCH.md; per conv launch as bracketed by bench.py; kernels classified by "
                   "tools/pmc_traffic.py::classify" % steps)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
