# A/B of the stream the collectives are launched from (EpisodeTrainer.collective_stream) under two gloo ranks on one GPU.
# The FSD_COLLECTIVE_STREAM variable it sets was a probe-time hook (tools/experiments_r06/rccl_stream_probes.patch); the
# product keeps the class attribute only.  Recorded result: the readiness timeline of this time-shared harness fluctuates
# run to run with every choice (meta / comm / wgrad).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06i
for cs in meta comm wgrad; do
for sc in strong; do
FSD_COLLECTIVE_STREAM=$cs FSD_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 4 --warmup 2 --batch 4 --classes 3 --size 160 --support 160 --scaling $sc --no-extras --no-cpu-baseline --no-parity > gpurun_out/r06i/g2_$cs.json 2> gpurun_out/r06i/g2_$cs.err
python - <<P
import json
for ln in open('gpurun_out/r06i/g2_$cs.err'):
    if ln.startswith("bench_full "):
        d=json.loads(ln[len("bench_full "):])
        print("$cs $sc", d["ms_per_step"], d["dp"]["overlap"]["gpu_ms_ready_before_backward_end"], d["dp"]["overlap"]["launch_host_ms_after_backward_start"], d["dp"]["overlap"]["backward_enqueue_host_ms"])
P
done; done
