#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04f7"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
export FSD_LB_ONLY=13,1024,1024
timeout 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$O/p1" -o run -- python "$R/tools/layer_bench.py" fwd > "$O/p1.log" 2>&1
cd "$R"
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/r04f7/p*/run_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'split8' in k:
            short=k.split('::')[-1].split('(')[0][:60]
            agg[short][r['Counter_Name']]+=float(r['Counter_Value']); cnt[short][r['Counter_Name']]+=1
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-40s %.4g per launch (n=%d)'%(c, x/cnt[k][c], cnt[k][c]))
PY
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -delete
