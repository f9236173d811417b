# is a small per-rank batch (the strong-scaling rank slices, configs[3]) host-bound?  step time vs sum of kernel time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06l
for dt in f32 bf16; do for b in 8 16 32 64; do
python bench.py --dtype $dt --batch $b --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity > gpurun_out/r06l/b${b}_$dt.json 2> gpurun_out/r06l/b${b}_$dt.err
python - <<P
import json
d=json.loads(open('gpurun_out/r06l/b${b}_$dt.json').read().strip().splitlines()[-1])
print('$dt B=$b', 'ms/step %.2f'%d['ms_per_step'], 'unprofiled %.2f'%(d['streams'].get('ms_per_step_unprofiled') or 0), 'kernel-sum(one stream) %.2f'%d['roofline'].get('timed_kernel_ms_per_step',0), d['step_gpu_ms_median_max'])
P
done; done
