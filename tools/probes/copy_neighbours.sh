#!/bin/bash
# which kernels surround the __amd_rocclr_copyBuffer dispatches of a bf16 step (30 per step: who issues them?)
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r06m"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d "$O/p" -o run -- python "$R/bench.py" --dtype bf16 --steps 4 --warmup 2 --profile-steps 0 --no-settle --streams 0 --no-cpu-baseline --no-extras --no-parity > "$O/log" 2>&1
echo "rc=$?"
cd "$R"; db=$(find "$O/p" -name '*.db' | head -1)
python - "$db" <<'P'
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]; kd = [t for t in tabs if "kernel_dispatch" in t][0]
names = dict(c.execute("select id, kernel_name from %s" % ks))
rows = list(c.execute("select start, end, kernel_id, grid_size_x, workgroup_size_x from %s order by start" % kd))
def dem(n):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)(?:I|E)", n)
    return m.group(1) if m else n[:50]
rows = [(s, e, dem(names[k]), g, w) for s, e, k, g, w in rows]
n = len(rows)
seg = rows[int(n * 0.7):]
cnt = collections.Counter()
for i, r in enumerate(seg):
    if "copyBuffer" in r[2]:
        cnt[(seg[i - 1][2] if i else None, r[3], seg[i + 1][2] if i + 1 < len(seg) else None)] += 1
for k, v in cnt.most_common(25):
    print(v, k)
mc = [t for t in tabs if "memory_copy" in t]
for t in mc:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
    print(t, cols)
    print(list(c.execute("select count(*) from %s" % t)))
    try:
        for row in c.execute("select size, count(*) from %s group by size order by count(*) desc limit 12" % t):
            print("  size", row)
    except Exception as e:
        print(e)
P
rm -rf "$O/p"
