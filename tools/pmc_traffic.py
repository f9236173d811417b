"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) into HBM bytes per conv launch of
bench.py.  A "conv launch" is what bench.py brackets with HIP events: one direct implicit-GEMM kernel, or the three
kernels of the Winograd pipeline (input transform, 36 or 16 batched GEMMs, output transform).  FETCH_SIZE is doubled
(gfx950 correction, MI355X_MICROARCH.md section HBM); both counters are in KiB.
Usage: python tools/pmc_traffic.py <fetch_prefix> <write_prefix> <steps_profiled> <conv_launches_per_step> <out.json> [episode]
episode: "metric_string" (64 queries 416x416 + 20 supports 224x224, the headline) or "configs1" -- bench.py only quotes a
traffic file on the episode it was measured on."""
import csv
import json
import sys

NEEDLES = ("conv_gemm_kernel", "conv_gemm_bf16_kernel", "wino_input_kernel", "wino_output_kernel", "wino4_input_kernel",
           "wino4_output_kernel", "conv_first_kernel")


def total(prefix, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(prefix + "_counter_collection.csv")):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in NEEDLES):
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


def main():
    fetch_prefix, write_prefix, steps, per_step, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    episode = sys.argv[6] if len(sys.argv) > 6 else "metric_string"
    f, nf = total(fetch_prefix, "FETCH_SIZE")
    w, nw = total(write_prefix, "WRITE_SIZE")
    launches = steps * per_step
    res = {"kernels": list(NEEDLES), "kernel_dispatches_fetch_pass": nf, "kernel_dispatches_write_pass": nw,
           "conv_launches": launches,
           "fetch_bytes_per_launch_x2": 2.0 * f * 1024 / launches, "write_bytes_per_launch": w * 1024 / launches}
    res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch_x2"] + res["write_bytes_per_launch"]
    res["episode"] = episode
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --streams 0 --steps %d --warmup 1`; "
                   "FETCH_SIZE x2 per MI355X_MICROARCH.md; per conv launch as bracketed by bench.py" % (steps - 1))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
