#!/bin/bash
# Round 5: per-layer HBM byte ledger (PMC FETCH_SIZE / WRITE_SIZE, one pass each) of the fp32 convolution launches.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05l"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/$c" -o run -- python "$R/tools/traffic_ledger.py" run > "$O/$c.log" 2>&1
  echo "$c rc=$?"
done
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
python tools/traffic_ledger.py table "$O/FETCH_SIZE" "$O/WRITE_SIZE" > "$O/ledger.csv"; cat "$O/ledger.csv" | cut -c1-220
du -sh "$O"
