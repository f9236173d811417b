"""Host-side bookkeeping of fewshot_detection_amd/streams.py (no GPU): who waits for what."""
import types

import torch

from fewshot_detection_amd import streams


class _FakeStream(object):
    def __init__(self):
        self.waited = []

    def wait_event(self, ev):
        self.waited.append(ev)


def test_publish_then_await_makes_the_reader_wait_exactly_once(monkeypatch):
    cur = _FakeStream()
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: cur)
    streams._READY.clear()
    t = torch.zeros(4)
    ev = object()
    streams.publish(t, ev)
    view = t[:2]                                   # same storage address: the reader may hold a view
    assert streams.await_tensor(view) is True and cur.waited == [ev]
    assert streams.await_tensor(t) is False and cur.waited == [ev]          # consumed: nothing left to wait for
    other = _FakeStream()
    streams.publish(t, ev)
    assert streams.await_tensor(t, other) is True and other.waited == [ev] and cur.waited == [ev]


def test_unclaimed_events_do_not_pile_up():
    streams._READY.clear()
    keep = [torch.zeros(1) for _ in range(200)]
    for t in keep:
        streams.publish(t, object())
    assert len(streams._READY) <= 65


def test_keep_alive_ignores_host_tensors_and_none():
    rec = types.SimpleNamespace(calls=0)
    streams.keep_alive(rec, None, torch.zeros(2))              # CPU tensors have no stream to record: no call, no error
    assert rec.calls == 0


def test_switches_follow_the_environment(monkeypatch):
    import importlib
    monkeypatch.setenv("FSD_STREAMS", "0")
    monkeypatch.setenv("FSD_STREAMS_WGRAD", "0")
    mod = importlib.reload(streams)
    try:
        assert mod.ENABLED is False and mod.WGRAD is False and mod.META is True
    finally:
        monkeypatch.delenv("FSD_STREAMS")
        monkeypatch.delenv("FSD_STREAMS_WGRAD")
        importlib.reload(streams)
    assert streams.ENABLED is True and streams.WGRAD is True


def test_early_backward_registry_hands_a_context_out_once_and_never_a_stale_one():
    """streams.register_early / take_early: ONE pending entry per device, held by a weak reference and matched
    on address AND shape; a forward that is never followed by a backward neither pins its tape nor survives the next
    forward of a reweighting net (clear_early), so a later step whose vectors land at the same address cannot take it."""
    class Ctx(object):
        pass
    streams._EARLY.clear()
    outs = [torch.zeros(3) for _ in range(5)]
    ctxs = [Ctx() for _ in outs]
    for t, c in zip(outs, ctxs):
        streams.register_early(t, c)
    assert len(streams._EARLY) == 1                                # the newest forward replaced the older ones
    assert streams.take_early(outs[0]) is None and streams.take_early(outs[3]) is None
    assert streams.take_early(outs[4][:2]) is None                 # same address, another shape: not these vectors
    assert streams.take_early(outs[4]) is ctxs[4] and streams.take_early(outs[4]) is None     # handed out once
    # a context that died (its graph was freed without a backward) is not resurrected
    streams.register_early(outs[1], ctxs[1])
    ctxs[1] = None
    import gc
    gc.collect()
    assert streams.take_early(outs[1]) is None
    # a new producer forward on the device drops whatever is pending
    streams.register_early(outs[2], ctxs[2])
    streams.clear_early(outs[2].device)
    assert streams.take_early(outs[2]) is None


def test_side_output_marks_are_bounded_and_queryable():
    streams._FROM_SIDE.clear()
    keep = [torch.zeros(2) for _ in range(40)]
    for t in keep:
        streams.mark_side_output(t)
    assert len(streams._FROM_SIDE) <= 16
    assert streams.from_side(keep[-1]) and not streams.from_side(keep[0]) and not streams.from_side(torch.zeros(2))


def test_autotune_is_a_noop_without_a_gpu_or_with_streams_off(monkeypatch):
    """streams.autotune only ever acts on a HIP device with the side streams enabled; elsewhere it reports and leaves the
    switch alone (the timing logic itself runs on the GPU box: tests/test_gpu_streams.py)."""
    calls = []
    monkeypatch.setattr(streams, "ENABLED", False)
    rep = streams.autotune(lambda: calls.append(1))
    assert rep["enabled_before"] is False and rep["enabled_after"] is False and rep["tries"] == [] and not calls
    if not torch.cuda.is_available():
        monkeypatch.setattr(streams, "ENABLED", True)
        rep = streams.autotune(lambda: calls.append(1))
        assert rep["enabled_after"] is True and rep["tries"] == [] and not calls
