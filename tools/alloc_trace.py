"""Where do the device allocations inside bench.py's timed region come from?  

Replays bench.py's Leg.run() schedule -- warm-up, stream autotune, one profiled-form step, the settle loop, then the timed
steps with one profiled step in the middle -- with the caching allocator's history recorder on, and prints every
`segment_alloc` (= hipMalloc) that falls inside the timed region: its size, the step it happened in, the stream, and the python
frames that asked for the memory.

    python tools/alloc_trace.py [--dtype f32|bf16] [--steps 20] > gpurun_out/alloc_trace.txt
"""
import argparse
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
from fewshot_detection_amd import cfgs, ops, streams  # noqa: E402
from fewshot_detection_amd.cfg import cfg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg.neg_ratio = 1
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
    sys.stdout, real = sys.stderr, sys.stdout
    leg = bench.Leg(dyn_cfg, rw_cfg, a.dtype, dev, None, 64, "train")
    x, metax, mask, target = bench.synth_episode(1000, 64, 20, 416, 224)
    x, metax, mask = x.to(dev).contiguous(), metax.to(dev), mask.to(dev)
    step = leg.stepper(x, metax, mask, target)
    marks = []                              # (label, number of device allocations so far)

    def counted():
        marks.append(("step", torch.cuda.memory_stats(dev).get("num_device_alloc", 0)))
        return step()

    torch.cuda.memory._record_memory_history(max_entries=400000, context="alloc", stacks="python")
    # the same schedule as the driver's command; Leg.run counts the timed region's allocations itself
    before_all = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    r = leg.run(counted, a.steps, a.warmup, 1, streams.ENABLED)
    snap = torch.cuda.memory._snapshot()
    torch.cuda.memory._record_memory_history(enabled=None)
    sys.stdout = real
    n_timed = r["device_allocs_in_timed_region"]
    print("device allocations: %d before the timed region, %d inside it (settle steps %d, prof index %s)"
          % (torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - before_all - n_timed, n_timed, r["settle_steps"],
             r["prof_index"]))
    # the timed steps are the LAST a.steps calls of counted(); the allocation counter at their starts
    starts = [m[1] for m in marks][-a.steps:]
    print("allocation counter at the start of each timed step (delta to the next):",
          [starts[i + 1] - starts[i] for i in range(len(starts) - 1)])
    events = []
    for tr in snap.get("device_traces", []):
        for e in tr:
            if e.get("action") == "segment_alloc":
                events.append(e)
    print("segment_alloc events recorded: %d (the last %d are inside the timed region)" % (len(events), n_timed))
    for e in events[-n_timed:] if n_timed else []:
        frames = [f for f in e.get("frames", []) if "site-packages/torch" not in f.get("filename", "")]
        where = " <- ".join("%s:%s %s" % (os.path.basename(f["filename"]), f["line"], f["name"]) for f in frames[:7])
        print("  %9.1f MB  stream %s  %s" % (e["size"] / 2.0 ** 20, e.get("stream"), where))
    st = torch.cuda.memory_stats(dev)
    print("reserved %.2f GiB, allocated peak %.2f GiB, inactive split %.2f GiB" % (
        st["reserved_bytes.all.current"] / 2 ** 30, st["allocated_bytes.all.peak"] / 2 ** 30,
        st["inactive_split_bytes.all.current"] / 2 ** 30))
    print("ms/step %.3f, step_gpu_ms %s" % (r["elapsed"] / r["steps"] * 1e3, r["step_gpu_ms"]))


if __name__ == "__main__":
    main()
