# A/B of EpisodeTrainer.EARLY_STEP and of the bucket count on the headline episode, both storage modes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06n
run() { tag=$1; shift; env $ENVV python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity --profile-steps 0 "$@" > gpurun_out/r06n/$tag.json 2> gpurun_out/r06n/$tag.err; python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r06n/$tag.json').read().strip().splitlines()[-1])
    print('$tag', d['ms_per_step'], d['step_gpu_ms_median_max'], d['allocator'])
except Exception as e:
    print('$tag FAILED', e); print(open('gpurun_out/r06n/$tag.err').read()[-1500:])
P
}
for rep in 1 2; do
ENVV="FSD_EARLY_STEP=0" run f32_late_b6_$rep --buckets 6
ENVV="FSD_EARLY_STEP=1" run f32_early_b6_$rep --buckets 6
ENVV="FSD_EARLY_STEP=1" run f32_early_b8_$rep --buckets 8
ENVV="FSD_EARLY_STEP=1" run f32_early_b12_$rep --buckets 12
done
for rep in 1 2; do
ENVV="FSD_EARLY_STEP=0" run bf16_late_b6_$rep --buckets 6 --dtype bf16
ENVV="FSD_EARLY_STEP=1" run bf16_early_b6_$rep --buckets 6 --dtype bf16
ENVV="FSD_EARLY_STEP=1" run bf16_early_b8_$rep --buckets 8 --dtype bf16
done
