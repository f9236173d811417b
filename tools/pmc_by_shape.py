"""Per (kernel, grid size) sums of rocprofv3 --pmc counter_collection.csv files (one file per pass).
Usage: python tools/pmc_by_shape.py pass1.csv [pass2.csv ...]   -> table on stdout (derived: mfma_busy, wait fractions)."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            key = (short(r["Kernel_Name"]), r.get("Grid_Size", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
            c = r["Counter_Name"]
            sums[key][c] += float(r["Counter_Value"])
            cnt[key][c] += 1
    rows = []
    for key, d in sums.items():
        n = max(cnt[key].values())
        row = {"kernel": key[0], "grid": key[1], "n": n}
        for c, v in d.items():
            row[c] = v / cnt[key][c]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and row.get("SQ_BUSY_CU_CYCLES"):
            row["mfma_busy"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * row["SQ_BUSY_CU_CYCLES"])
        if row.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in row:
                    row["f_" + c[3:]] = row[c] / row["SQ_WAVE_CYCLES"]
        if row.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_frac"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
        if "TCC_HIT_sum" in row:
            row["l2_hit"] = row["TCC_HIT_sum"] / max(1.0, row["TCC_HIT_sum"] + row.get("TCC_MISS_sum", 0.0))
        rows.append(row)
    rows.sort(key=lambda r: -r.get("SQ_BUSY_CU_CYCLES", r.get("SQ_WAVE_CYCLES", 0.0)))
    keys = []
    for r in rows:
        for k in r:
            if k not in keys:
                keys.append(k)
    print(",".join(keys))
    for r in rows[:40]:
        print(",".join(("%.4g" % r[k]) if isinstance(r.get(k), float) else '"%s"' % r.get(k, "") for k in keys))


if __name__ == "__main__":
    main()
