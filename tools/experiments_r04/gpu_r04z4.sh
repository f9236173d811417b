#!/bin/bash
# per-kernel view of the two-image forward with and without split-K (rocprofv3 kernel trace only)
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04z4"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
for ks in off a 4; do
  if [ $ks = off ]; then unset FSD_KSPLIT; else export FSD_KSPLIT=$ks; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ks_$ks" -o run -- python "$R/tools/probes/inference_b2.py" > "$O/ks_$ks.log" 2>&1; echo "ks=$ks rc=$?"
done
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*agent_info.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete
for ks in off a 4; do echo "== $ks"; head -8 "$O/ks_$ks/run_kernel_stats.csv" | cut -c1-150; done
