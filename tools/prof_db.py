"""Summaries of a rocprofv3 --kernel-trace results DB: per-kernel totals (like --stats) and, with --step, the ordered
launch list of ONE training step (median over the steps in the trace) with grid sizes -- the per-layer view.
Usage: python tools/prof_db.py <results.db> [--steps N] [--csv out.csv] [--launches kernel_substring]"""
import argparse
import collections
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=0, help="training steps in the trace (totals are divided by it)")
    ap.add_argument("--csv")
    ap.add_argument("--launches", help="list every launch of kernels whose name contains this")
    ap.add_argument("--timeline", action="store_true", help="overlap summary: wall span, union of kernel intervals, per-queue busy")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    names = dict(c.execute("select id, kernel_name from %s" % ks))
    rows = list(c.execute("select kernel_id, start, end, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x "
                          "from %s order by start" % kd))
    tot = collections.defaultdict(lambda: [0, 0.0])
    for kid, s, e, gx, gy, gz, wx in rows:
        t = tot[short(names[kid])]
        t[0] += 1
        t[1] += (e - s) / 1e6
    if a.timeline:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
        qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
        iv = list(c.execute("select start, end%s from %s order by start" % ((", " + qcol) if qcol else ", 0", kd)))
        # drop the warm-up: keep the last 60 % of the trace
        t_lo = iv[0][0] + 0.4 * (iv[-1][1] - iv[0][0])
        iv = [x for x in iv if x[0] >= t_lo]
        span = (max(x[1] for x in iv) - iv[0][0]) / 1e6
        union, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
        for s_, e_, _ in iv[1:]:
            if s_ > cur_e:
                union += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        union += cur_e - cur_s
        per_q = collections.defaultdict(float)
        for s_, e_, q in iv:
            per_q[q] += (e_ - s_) / 1e6
        print("# timeline (last 60 %% of the trace): span %.2f ms, GPU busy (union of kernel intervals) %.2f ms = %.1f %%, "
              "sum of kernel durations %.2f ms (x%.2f of busy: overlap)" % (span, union / 1e6, 100 * union / 1e6 / span,
                                                                       sum(per_q.values()), sum(per_q.values()) / (union / 1e6)))
        for q, ms in sorted(per_q.items(), key=lambda kv: -kv[1]):
            print("#   queue %s: %.2f ms of kernels" % (q, ms))
    div = a.steps or 1
    allms = sum(v[1] for v in tot.values())
    out = ["kernel,calls%s,total_ms%s,avg_us,percent" % (("_per_step",) * 2 if a.steps else ("", ""))]
    for k, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        out.append('"%s",%.1f,%.3f,%.1f,%.1f' % (k, n / div, ms / div, ms / n * 1e3, 100 * ms / allms))
    text = "\n".join(out)
    if a.csv:
        open(a.csv, "w").write(text + "\n")
    print(text)
    print("# all kernels: %.3f ms%s" % (allms / div, " per step" if a.steps else ""))
    if a.launches:
        print("# launches of *%s*: start_ms, dur_us, grid (workgroups), kernel" % a.launches)
        t0 = rows[0][1]
        for kid, s, e, gx, gy, gz, wx in rows:
            if a.launches in names[kid]:
                print("%10.3f %9.1f  %6d x %4d x %3d  %s" % ((s - t0) / 1e6, (e - s) / 1e3, gx // max(1, wx), gy, gz, short(names[kid])))


if __name__ == "__main__":
    main()
