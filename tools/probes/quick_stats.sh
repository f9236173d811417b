#!/bin/bash
# rocprofv3 kernel stats of 6 one-stream steps: $1 = f32|bf16; prints the top kernels per step
set -u
R="$GRAFT_REPO_ROOT"; dt=${1:-f32}; O="$R/gpurun_out/r06q_$dt"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/p" -o run -- python "$R/bench.py" --dtype $dt --steps 6 --warmup 2 --profile-steps 0 --no-settle --streams 0 --no-cpu-baseline --no-extras --no-parity ${EXTRA:-} > "$O/log" 2>&1
echo "rc=$?"
cd "$R"; f="$O/p/run_kernel_stats.csv"
[ -f "$f" ] && python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 8.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel time per step: %.3f ms" % (tot / 1e6 / steps))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 34]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-86s %5.1f/step %7.3f ms/step %7.1f us" % (n[:86], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
P
cp "$f" "$O/kernel_stats.csv" 2>/dev/null; rm -rf "$O/p"
