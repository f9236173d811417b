#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
for pad in 0 2112 528; do
  if [ "$pad" != "0" ]; then
    ( cd fewshot_detection_amd/csrc && rm -f winograd.o && make EXTRA=-DFSD_WINO_PAD=$pad 2>&1 | grep -E "error|Error" ; )
  fi
  echo "=== FSD_WINO_PAD=$pad"
  timeout 300 python tools/layer_bench.py all 2>&1 | grep -v amdgpu.ids | grep "k3" | sed -n '2,8p' | cut -c1-150
done
