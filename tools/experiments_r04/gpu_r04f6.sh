#!/bin/bash
# PMC counters of the fused kernel on one layer
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04f6"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
export FSD_LB_ONLY=104,64,128
pmc() { local name=$1; local ctr=$2
  timeout 90 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$R/tools/layer_bench.py" fwd > "$O/$name.log" 2>&1
  echo "$name rc=$?"; }
pmc p1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
pmc p2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES"
pmc p3 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE"
pmc p4 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum"
cd "$R"
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/r04f6/p*/run_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'gemm_out' in k or 'input_planes' in k or 'split_planes' in k:
            short=k.split('::')[-1].split('(')[0][:40]
            agg[short][r['Counter_Name']]+=float(r['Counter_Value']); cnt[short][r['Counter_Name']]+=1
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-40s %.4g per launch (n=%d)'%(c, x/cnt[k][c], cnt[k][c]))
PY
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -delete
tail -3 "$O/p4.log"
