"""Summarise the rocm-smi samples taken beside each GEMM regime (tools/gpu_r04a.sh) and the per-kernel effective clock
(GRBM_GUI_ACTIVE / 8 XCDs / kernel duration) of a PMC pass -> profiles/rNN_power_probe.txt.
Usage: python tools/power_summary.py <gpurun_out/r04a> [<pmc_dir_with_mfma_pass> ...] > profiles/r04_power_probe.txt"""
import csv
import os
import re
import sys


def smi(path):
    w, mhz, temp = [], [], []
    for line in open(path):
        m = re.search(r"Power \(W\): ([\d.]+)", line)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", line)
        t = re.search(r"junction\) \(C\): ([\d.]+)", line)
        if m and c:
            w.append(float(m.group(1)))
            mhz.append(float(c.group(1)))
            if t:
                temp.append(float(t.group(1)))
    return w, mhz, temp


def main():
    d = sys.argv[1]
    print("# rocm-smi sampled every ~0.4 s beside a loop of ONE workload on an otherwise idle MI355X (tools/gpu_r04a.sh);")
    print("# samples of the first 1.5 s (ramp) dropped; W = socket graphics package power, MHz = sclk")
    print("%-18s %8s %8s %8s %8s  %s" % ("regime", "W_mean", "W_max", "MHz_mean", "T_junc", "result"))
    for name in ("idle", "native", "split", "bf16", "yard_random", "yard_zeros", "yard_random_4096"):
        f = os.path.join(d, "smi_%s.txt" % name)
        if not os.path.exists(f):
            continue
        w, mhz, temp = smi(f)
        w, mhz, temp = w[4:-1] or w, mhz[4:-1] or mhz, temp[4:-1] or temp
        res = open(os.path.join(d, "run_%s.txt" % name)).read().strip().splitlines()
        res = res[-1] if res else ""
        print("%-18s %8.0f %8.0f %8.0f %8.0f  %s" % (name, sum(w) / len(w), max(w), sum(mhz) / len(mhz),
                                                    sum(temp) / max(1, len(temp)), res))
    print()
    print("# regimes: native = Winograd F(4x4) 13x13 1024->1024 layer on v_mfma_f32_32x32x2_f32; split = the same layer on six")
    print("# v_mfma_f32_32x32x16_bf16 terms (round-3 4-wave kernel); bf16 = the bf16 storage mode's DMA kernel on the same layer;")
    print("# yard_* = torch.matmul bf16 8192^3 (hipBLASLt), a measuring stick outside the product path")
    for pd in sys.argv[2:]:
        cc = os.path.join(pd, "run_counter_collection.csv")
        kt = os.path.join(pd, "run_kernel_trace.csv")
        if not (os.path.exists(cc) and os.path.exists(kt)):
            continue
        dur = {}
        for r in csv.DictReader(open(kt)):
            dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"])
        agg = {}
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
                continue
            ns, name = dur[r["Dispatch_Id"]]
            name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name)).split("(")[0]
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += ns
            a[2] += 1
        print()
        print("# effective shader clock per kernel = GRBM_GUI_ACTIVE / 8 (XCDs) / kernel wall time, PMC pass %s" % pd)
        for name, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print("%-70s launches %4d  avg %8.1f us  %6.0f MHz" % (name[:70], n, ns / n / 1e3, cyc / 8.0 / ns * 1e3))


if __name__ == "__main__":
    main()
