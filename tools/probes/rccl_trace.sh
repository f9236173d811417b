#!/bin/bash
# kernel-trace timeline of the train step: plain vs one-rank RCCL (streams on).  Where do the +9 ms of the RCCL form go?
set -u
R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for mode in plain rccl; do
  O="$R/gpurun_out/r06j_$mode"; mkdir -p "$O"; cd /tmp; rm -rf "$O/prof"
  E=""; [ $mode = rccl ] && E="FSD_BENCH_SINGLE_RANK_RCCL=1"
  env $E timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof" -o run -- python "$R/bench.py" --steps 8 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity > "$O/prof.log" 2>&1
  echo "$mode rocprof rc=$?"; tail -1 "$O/prof.log" | cut -c1-200
  cd "$R"; db=$(find "$O/prof" -name '*.db' | head -1)
  python tools/prof_db.py "$db" --timeline --csv "$O/kernels.csv" | grep "^#" | head -12
  python - "$db" > "$O/gaps.txt" <<'P'
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]; kd = [t for t in tabs if "kernel_dispatch" in t][0]
names = dict(c.execute("select id, kernel_name from %s" % ks))
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
q = "queue_id" if "queue_id" in cols else "stream_id"
rows = list(c.execute("select start, end, kernel_id, %s from %s order by start" % (q, kd)))
t_lo = rows[0][0] + 0.6 * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
sh = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::|^void ", "", n))[:60]
# idle gaps of the whole GPU > 20 us, with the kernel before and after
cur_e, last = rows[0][1], rows[0]
gaps = []
for r in rows[1:]:
    if r[0] > cur_e + 20000:
        gaps.append((r[0] - cur_e, sh(names[last[2]]), last[3], sh(names[r[2]]), r[3]))
    if r[1] > cur_e:
        cur_e, last = r[1], r
print("idle gaps > 20 us in the last 40 %% of the trace: %d, total %.2f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
for g in sorted(gaps, reverse=True)[:25]:
    print("%8.1f us  after %s (q%s)  before %s (q%s)" % (g[0] / 1e3, g[1], g[2], g[3], g[4]))
P
  head -30 "$O/gaps.txt"
  find "$O" -name "*.db" -size +40M -delete
done
