"""Per-kernel HBM-side traffic (and any other PMC counter) from rocprofv3 --pmc passes.
Usage: python tools/pmc_kernels.py OUT.csv STEPS name=counter_collection.csv [name=...]
Each pass file holds ONE or more counters (rocprofv3 --pmc A B --output-format csv); FETCH_SIZE and WRITE_SIZE are in KiB,
FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md section HBM).  Output: one row per
kernel with launches per step, counter sums per step and -- when both FETCH_SIZE and WRITE_SIZE were collected --
the HBM bytes per launch and per step."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:100]


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    counts = collections.defaultdict(lambda: collections.defaultdict(int))
    counters = []
    for spec in sys.argv[3:]:
        path = spec.split("=", 1)[1]
        for r in csv.DictReader(open(path)):
            k, c = short(r["Kernel_Name"]), r["Counter_Name"]
            sums[k][c] += float(r["Counter_Value"])
            counts[k][c] += 1
            if c not in counters:
                counters.append(c)
    rows = []
    for k in sums:
        n = max(counts[k].values()) / steps
        row = {"kernel": k, "launches_per_step": round(n, 2)}
        for c in counters:
            row[c + "_per_step"] = sums[k][c] / steps
        if "FETCH_SIZE" in sums[k] and "WRITE_SIZE" in sums[k]:
            b = (2.0 * sums[k]["FETCH_SIZE"] + sums[k]["WRITE_SIZE"]) * 1024.0 / steps
            row["hbm_mb_per_step"] = b / 1e6
            row["hbm_mb_per_launch"] = b / 1e6 / n if n else 0.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sums[k] and sums[k].get("SQ_BUSY_CU_CYCLES"):
            row["mfma_busy"] = sums[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * sums[k]["SQ_BUSY_CU_CYCLES"])
        rows.append(row)
    rows.sort(key=lambda r: -r.get("hbm_mb_per_step", r.get("SQ_BUSY_CU_CYCLES_per_step", 0.0)))
    keys = ["kernel", "launches_per_step"] + [k for k in rows[0] if k not in ("kernel", "launches_per_step")]
    allk = []
    for r in rows:
        for k in r:
            if k not in allk:
                allk.append(k)
    with open(out, "w") as fh:
        w = csv.DictWriter(fh, fieldnames=allk)
        w.writeheader()
        for r in rows:
            w.writerow({k: (("%.4f" % v) if isinstance(v, float) else v) for k, v in r.items()})
    tot = sum(r.get("hbm_mb_per_step", 0.0) for r in rows)
    print("wrote %s: %d kernels, %.1f MB HBM-side traffic per step" % (out, len(rows), tot))


if __name__ == "__main__":
    main()
