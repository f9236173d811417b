"""Sum rocprofv3 --pmc counters per kernel name: python pmc_sum.py <substring> <counter_collection.csv> ...  -> one line per file"""
import csv, sys, collections
needle = sys.argv[1]
for path in sys.argv[2:]:
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    d = max(n.values()) if n else 0
    print(path.split("/")[-3] if path.count("/") > 2 else path, "dispatches", d, {k: "%.4g" % (v / max(1, n[k])) for k, v in sorted(acc.items())})
