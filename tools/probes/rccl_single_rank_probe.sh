# one-rank RCCL form of the data-parallel step against the group-less step (bench.py, headline episode), both storage modes
mkdir -p gpurun_out/r06i; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity --profile-steps 0"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r06i/$tag.json 2> gpurun_out/r06i/$tag.err; python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r06i/$tag.json').read().strip().splitlines()[-1])
    print('$tag', d['ms_per_step'], d['streams'].get('enabled'), d['step_gpu_ms_median_max'], d['allocator'])
except Exception as e:
    print('$tag FAILED', e); print(open('gpurun_out/r06i/$tag.err').read()[-1500:])
P
}
R=FSD_BENCH_SINGLE_RANK_RCCL=1
run fix_plain_f32 A=1
run fix_rccl_f32 $R
B="$B --dtype bf16"
run fix_plain_bf16 A=1
run fix_rccl_bf16 $R
