"""HIP-stream plumbing of one episode step: what runs beside the detector's main stream.

The step has three independent strands (train_meta.py:201-226 runs them back to back on one CUDA stream):

  main    detector forward, loss, detector backward (activation / data-gradient chain), optimizer
  "meta"  the reweighting net -- forward beside the detector backbone (the two only meet at the fused
          reweight (x) head GEMM), backward beside the detector's backward (it needs nothing but d(vectors),
          which the head's backward produces first)
  "wgrad" the detector's weight gradients: off the critical path (nothing in the backward sweep reads a dW),
          MFMA-bound, so they fill the matrix cores while the main stream runs its HBM-bound passes
          (activation backward, Winograd transforms) and the tails of its own GEMM launches

The per-step host data (RegionLoss targets) goes up through a small ring of PINNED staging buffers (upload() below), as an
asynchronous copy on the current stream: a pageable copy blocks the host until its stream has drained, and the side
stream that used to hide that ("copy", rounds 3-5) was one stream too many -- HIP maps streams onto four hardware queues,
and with a collective backend's own stream in the process the copy stream landed on the main stream's queue: every step
of a one-rank RCCL run took 34 ms instead of 25 (round 6, tools/experiments_r06/rccl_stream_probes.patch).  For the same
reason the trainer's collectives are launched from the "meta" stream (dp.py) instead of a stream of their own.

Kernels are unchanged and deterministic, so results are bit-identical with and without the side streams;
only the order in which independent launches reach the GPU differs.  Ordering is by events; tensors that cross a
stream are handed to the caching allocator with record_stream so their memory is not re-used under a pending reader.

FSD_STREAMS=0 (or streams.ENABLED = False) runs everything on the current stream.
"""
import os
import threading

import torch

ENABLED = os.environ.get("FSD_STREAMS", "1") != "0"
META = os.environ.get("FSD_STREAMS_META", "1") != "0"        # the reweighting net on its own stream
WGRAD = os.environ.get("FSD_STREAMS_WGRAD", "1") != "0"      # the detector's weight gradients on their own stream

_SIDE = {}        # (device index, name) -> torch.cuda.Stream
_READY = {}       # (device index, data_ptr) -> event that fires when the tensor stored there is complete (insertion-ordered)
_LOCK = threading.Lock()
_MAX_READY = 64


def side(device, name):
    """The side stream `name` of `device` (created on first use, kept for the life of the process)."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), name)
    s = _SIDE.get(key)
    if s is None:
        s = torch.cuda.Stream(device=key[0])
        _SIDE[key] = s
    return s


_PINNED = {}      # device index -> [slot, ...]; slot = [pinned uint8 buffer, event of the last copy out of it]
_PIN_NEXT = {}
_PIN_SLOTS = 6    # more than the steps a trainer keeps in flight (dp.EpisodeTrainer.max_steps_in_flight = 2) x uploads per step


def upload(array, device):
    """Host numpy array -> new device tensor without ever blocking the host: the array is written into a pinned staging
    slot and a KERNEL on the current stream reads it from there (fsd_upload_words; on this runtime a hipMemcpyAsync --
    pageable or pinned -- queued behind pending work makes the host wait for that work).  A slot is re-used only after the
    kernel that last read it has completed (its event; by then long past)."""
    import numpy as np
    from . import ops
    dev = torch.device(device)
    arr = np.ascontiguousarray(array)
    t = torch.from_numpy(arr)
    if dev.type != "cuda":
        raise RuntimeError("fewshot_detection_amd needs a HIP device (got %s); there is no CPU fallback" % dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    nbytes = (arr.nbytes + 3) // 4 * 4
    with _LOCK:
        slots = _PINNED.setdefault(idx, [])
        if len(slots) < _PIN_SLOTS:
            slots.append([None, None])
        k = _PIN_NEXT.get(idx, 0) % len(slots)
        _PIN_NEXT[idx] = k + 1
        slot = slots[k]
    if slot[1] is not None:
        slot[1].synchronize()
    if slot[0] is None or slot[0].numel() < nbytes:
        slot[0] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
    out = torch.empty(t.shape, dtype=t.dtype, device=dev)
    if arr.nbytes:
        # (numpy's single-threaded memcpy: torch's copy_ fans a 2 MB copy out to the intra-op thread pool, whose wake-up
        # stalled the host for tens of ms now and then on a shared box)
        np.copyto(slot[0].numpy()[:arr.nbytes], arr.reshape(-1).view(np.uint8))
        if out.numel() * out.element_size() % 4:        # (odd byte counts: the kernel moves whole words into a padded twin)
            pad = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ops.upload_words(slot[0], pad, nbytes // 4)
            out.view(torch.uint8).reshape(-1).copy_(pad[:arr.nbytes])
        else:
            ops.upload_words(slot[0], out, nbytes // 4)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    slot[1] = ev
    return out


def reset():
    """Forget the side streams (new ones are taken from torch's pool on next use) and everything keyed on them.  HIP maps
    streams onto a handful of hardware queues; an unlucky mapping (a side stream sharing the main stream's queue) turns the
    overlap into a slowdown for the life of the streams -- autotune() below uses this to draw a new mapping."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    with _LOCK:
        _SIDE.clear()
        _READY.clear()
        _FROM_SIDE.clear()
        _EARLY.clear()


def autotune(step, tries=3, reps=2, slack=1.03, good=0.96, fixed_schedule=False):
    """Find side streams that pay on THIS process / GPU: time `reps` synchronised calls of `step` on one stream and with the
    side streams.  A set that brings the step to `good` x the one-stream time or better is kept at once (the overlap is worth
    ~5 % of the step when it works); otherwise the streams are re-created (a new draw of the stream -> hardware-queue mapping)
    up to `tries` times and the BEST set seen is the one that stays -- unless even that is slower than `slack` x the one-stream
    form, in which case everything runs on one stream (ENABLED = False).  Measured on MI355X boxes of the pool: about one
    process in eight starts with a stream set on which a step takes 36-43 ms instead of 26 (the one-stream step: 27.5), and
    now and then with one that overlaps only half as well (28.8 against 25.4 ms), for as long as those streams live.
    -> dict(report).  A no-op when the streams are disabled.
    fixed_schedule: EVERY timing is run whatever the earlier ones showed (1 + tries timings of 1 + reps steps) -- for steps
    that hold collectives, where every rank must execute the same number of them; which set a rank keeps stays its own decision.
    NOTE for library users: `step` is EXECUTED (1 + reps calls per timing, up to 2 * (tries + 1) timings) -- if it is a training
    step, the weights, the optimizer state and BatchNorm's running statistics advance by that many steps and stay advanced."""
    import time
    global ENABLED
    report = {"enabled_before": ENABLED, "tries": []}
    # (fixed_schedule: a rank whose streams were switched off by an earlier call still runs the whole schedule -- the ranks
    # stay in lockstep -- and may switch them on again)
    if (not ENABLED and not fixed_schedule) or not torch.cuda.is_available():
        report["enabled_after"] = ENABLED
        return report

    def timed(on):
        global ENABLED
        ENABLED = on
        step()                                    # settle (allocator, caches) in this mode
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    try:
        off = timed(False)
        best = None                               # (ms, the stream set that gave it)
        for _ in range(tries):
            on = timed(True)
            report["tries"].append({"streams_ms": on, "one_stream_ms": off})
            if best is None or on < best[0]:
                with _LOCK:
                    best = (on, dict(_SIDE))
            if on <= good * off and not fixed_schedule:
                break
            reset()
        if best[0] <= slack * off:
            with _LOCK:
                _SIDE.clear()
                _SIDE.update(best[1])
            ENABLED = True
        else:
            ENABLED = False
        report["kept_ms"] = best[0] if ENABLED else off
    except Exception:
        ENABLED = report["enabled_before"]
        raise
    report["enabled_after"] = ENABLED
    return report


def _key(t):
    return (t.device.index if t.is_cuda else -1, t.data_ptr())


def publish(t, event):
    """`t` was produced on another stream; whoever reads it first calls await_tensor(t)."""
    with _LOCK:
        _READY.pop(_key(t), None)
        _READY[_key(t)] = event
        # entries nobody claimed (an exception between publish and use, a consumer that never ran): evict the OLDEST
        # ones only -- a blanket clear could drop another thread's live entry and turn its wait into a silent race
        while len(_READY) > _MAX_READY:
            _READY.pop(next(iter(_READY)))


def await_tensor(t, stream=None):
    """Make `stream` (default: the current one) wait for the producer of `t`, if one was published.  -> bool."""
    with _LOCK:
        ev = _READY.pop(_key(t), None)
    if ev is None:
        return False
    (stream or torch.cuda.current_stream()).wait_event(ev)
    return True


_FROM_SIDE = {}   # (device index, data_ptr) of the latest outputs computed on a side stream (bounded, insertion-ordered)


def mark_side_output(t):
    """Remember that `t` came out of a network that ran on a side stream (its backward will run there too and can
    start as soon as d(t) exists: backward.py publishes d(t) only for such tensors)."""
    with _LOCK:
        _FROM_SIDE.pop(_key(t), None)
        _FROM_SIDE[_key(t)] = True
        while len(_FROM_SIDE) > 16:
            _FROM_SIDE.pop(next(iter(_FROM_SIDE)))


def from_side(t):
    with _LOCK:
        return _key(t) in _FROM_SIDE


_EARLY = {}       # device index -> (data_ptr, shape, weakref to the autograd context of the network that made the vectors)


def register_early(t, ctx):
    """`t` = vectors that go straight into the detector of the same forward() call (single consumer): the detector's
    backward may run the producer's backward sweep itself as soon as d(t) exists (backward.run_early).
    ONE pending entry per device, held by a weak reference: a forward() that is never followed by a backward (the loss raised,
    an evaluation in grad mode) neither pins its tape nor survives the next forward of a reweighting net (clear_early)."""
    import weakref
    with _LOCK:
        _EARLY[_key(t)[0]] = (t.data_ptr(), tuple(t.shape), weakref.ref(ctx))


def clear_early(device):
    """A new reweighting-net forward starts on `device`: whatever an earlier forward registered is stale."""
    idx = torch.device(device).index
    with _LOCK:
        _EARLY.pop(idx if idx is not None else (torch.cuda.current_device() if torch.cuda.is_available() else -1), None)


def take_early(t):
    """The context registered for exactly these vectors (same device, address AND shape), if it is still alive."""
    with _LOCK:
        ent = _EARLY.get(_key(t)[0])
        if ent is None or ent[0] != t.data_ptr() or ent[1] != tuple(t.shape):
            return None
        del _EARLY[_key(t)[0]]
    return ent[2]()


def keep_alive(stream, *tensors):
    """Tell the allocator that `stream` has work pending on these tensors (allocated on another stream)."""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(stream)
