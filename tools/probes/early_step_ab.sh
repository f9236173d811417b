# A/B of EpisodeTrainer.EARLY_STEP on the headline episode, both storage modes, group-less and over one-rank RCCL
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06n
B="python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity --profile-steps 0"
run() { tag=$1; shift; env "$@" $B > gpurun_out/r06n/$tag.json 2> gpurun_out/r06n/$tag.err; python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r06n/$tag.json').read().strip().splitlines()[-1])
    print('$tag', d['ms_per_step'], d['step_gpu_ms_median_max'], d['allocator'])
except Exception as e:
    print('$tag FAILED', e); print(open('gpurun_out/r06n/$tag.err').read()[-1500:])
P
}
for rep in 1 2; do
run f32_late_$rep FSD_EARLY_STEP=0
run f32_early_$rep FSD_EARLY_STEP=1
done
B="$B --dtype bf16"
for rep in 1 2; do
run bf16_late_$rep FSD_EARLY_STEP=0
run bf16_early_$rep FSD_EARLY_STEP=1
done
run bf16_rccl_early FSD_EARLY_STEP=1 FSD_BENCH_SINGLE_RANK_RCCL=1
B="python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity --profile-steps 0"
run f32_rccl_early FSD_EARLY_STEP=1 FSD_BENCH_SINGLE_RANK_RCCL=1
