"""HIP-stream plumbing of one episode step: what runs beside the detector's main stream.

The step has three independent strands (train_meta.py:201-226 runs them back to back on one CUDA stream):

  main    detector forward, loss, detector backward (activation / data-gradient chain), optimizer
  "meta"  the reweighting net -- forward beside the detector backbone (the two only meet at the fused
          reweight (x) head GEMM), backward beside the detector's backward (it needs nothing but d(vectors),
          which the head's backward produces first)
  "wgrad" the detector's weight gradients: off the critical path (nothing in the backward sweep reads a dW),
          MFMA-bound, so they fill the matrix cores while the main stream runs its HBM-bound passes
          (activation backward, Winograd transforms) and the tails of its own GEMM launches
  "copy"  the per-step target upload (so the pageable copy does not make the host wait for the forward)

Kernels are unchanged and deterministic, so results are bit-identical with and without the side streams;
only the order in which independent launches reach the GPU differs.  Ordering is by events; tensors that cross a
stream are handed to the caching allocator with record_stream so their memory is not re-used under a pending reader.

FSD_STREAMS=0 (or streams.ENABLED = False) runs everything on the current stream.
"""
import os

import torch

ENABLED = os.environ.get("FSD_STREAMS", "1") != "0"
META = os.environ.get("FSD_STREAMS_META", "1") != "0"        # the reweighting net on its own stream
WGRAD = os.environ.get("FSD_STREAMS_WGRAD", "1") != "0"      # the detector's weight gradients on their own stream

_SIDE = {}        # (device index, name) -> torch.cuda.Stream
_READY = {}       # data_ptr -> event that fires when the tensor stored there is complete


def side(device, name):
    """The side stream `name` of `device` (created on first use, kept for the life of the process)."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), name)
    s = _SIDE.get(key)
    if s is None:
        s = torch.cuda.Stream(device=key[0])
        _SIDE[key] = s
    return s


def publish(t, event):
    """`t` was produced on another stream; whoever reads it first calls await_tensor(t)."""
    if len(_READY) > 64:          # entries nobody claimed (an exception between publish and use): drop them
        _READY.clear()
    _READY[t.data_ptr()] = event


def await_tensor(t, stream=None):
    """Make `stream` (default: the current one) wait for the producer of `t`, if one was published.  -> bool."""
    ev = _READY.pop(t.data_ptr(), None)
    if ev is None:
        return False
    (stream or torch.cuda.current_stream()).wait_event(ev)
    return True


def keep_alive(stream, *tensors):
    """Tell the allocator that `stream` has work pending on these tensors (allocated on another stream)."""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(stream)
