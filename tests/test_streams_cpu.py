"""engine.thread_streams: persistent per-thread stream slots (host logic; torch.cuda.Stream is stubbed — no GPU here)."""
import threading

import pytest


@pytest.fixture
def E(monkeypatch):
    import torch
    from face_crop_plus_amd import engine
    made = []

    class FakeStream:
        def __init__(self, device=None):
            self.q = [1, 2, 3, 3, 2, 1, 0, 3, 2, 1, 0][len(made) % 11]          # the measured hardware-queue assignment pattern
            made.append(self)
    import contextlib
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(engine, "STREAM_PROBE", False)           # the queue probe launches kernels: GPU only (tests/test_dist_gpu.py)
    monkeypatch.setattr(engine, "_stream_sets", {})
    monkeypatch.setattr(engine, "_stream_free", {})
    monkeypatch.setattr(engine, "_stream_tls", threading.local())
    engine._made = made
    return engine


def _in_thread(fn):
    out = []
    t = threading.Thread(target=lambda: out.append(fn()))
    t.start()
    t.join()
    return out[0]


def test_a_thread_keeps_its_streams_and_later_threads_inherit_the_slot(E):
    dev = "cuda:0"
    side = E.thread_side_streams(dev, 2)
    assert len(side) == 2 and E.thread_side_streams(dev, 2) is side          # every detector of the thread: the same pair
    assert E.thread_main_stream(dev) is E.thread_main_stream(dev)
    assert len(E._made) == 3
    # a worker thread claims the next slot ...
    w1 = _in_thread(lambda: (E.thread_main_stream(dev), E.thread_side_streams(dev, 2)))
    assert w1[0] is not E.thread_main_stream(dev) and w1[1] is not side and len(E._made) == 6
    # ... which the worker thread of the NEXT run inherits (the first one has ended): no new streams
    w2 = _in_thread(lambda: (E.thread_main_stream(dev), E.thread_side_streams(dev, 2)))
    assert w2[0] is w1[0] and w2[1] is w1[1] and len(E._made) == 6
    # two workers alive at once: distinct slots; afterwards both slots are free again and handed out lowest first
    gate, got = threading.Barrier(2), {}

    def work(name):
        got[name] = E.thread_side_streams(dev, 2)
        gate.wait()
    ts = [threading.Thread(target=work, args=(n,)) for n in "ab"]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert got["a"] is not got["b"] and {id(got["a"]), id(got["b"])} >= {id(w1[1])} and len(E._made) == 8
    assert _in_thread(lambda: E.thread_side_streams(dev, 2)) is w1[1]
    # another device has its own slots; another k its own pair inside the slot
    assert E.thread_side_streams("cuda:1", 2) is not side
    assert len(E.thread_side_streams(dev, 3)) == 3 and E.thread_side_streams(dev, 2) is side
    # a worker that runs on its main stream keeps one sub-batch there
    wm = E.thread_side_streams(dev, 2, with_main=True)
    assert wm[0] is E.thread_main_stream(dev) and wm[1] is not side[0] and len(wm) == 2


def test_candidates_that_alias_a_chosen_stream_are_set_aside(E, monkeypatch):
    """The hardware-queue probe: side streams of one thread never share a queue; aliasing candidates are kept alive but not used;
    the main stream is not probed."""
    monkeypatch.setattr(E, "STREAM_PROBE", True)
    monkeypatch.setattr(E, "_stream_rejects", [])
    import torch
    monkeypatch.setattr(torch.cuda, "_sleep", lambda c: None, raising=False)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    queue_of = lambda s_: s_.q
    monkeypatch.setattr(E, "_streams_overlap", lambda a, b: queue_of(a) != queue_of(b))
    main = E.thread_main_stream("cuda:0")                 # queue 1, no probe
    side = E.thread_side_streams("cuda:0", 2)             # queues 2 and 3
    assert [queue_of(x) for x in [main] + side] == [1, 2, 3] and not E._stream_rejects
    out = {}

    def second_worker():
        out["main"] = E.thread_main_stream("cuda:0")
        out["side"] = E.thread_side_streams("cuda:0", 2, with_main=True)
    t = threading.Thread(target=second_worker)
    t.start(); t.join()
    qs = [queue_of(x) for x in out["side"]]
    # the second worker's main is stream number 4 of the process and lands on queue 3 like the 3rd stream did; it hosts one
    # sub-batch itself, the other goes to the next stream (queue 2): no aliasing inside the thread, nothing set aside
    assert out["side"][0] is out["main"] and qs == [3, 2] and not E._stream_rejects
    # a thread whose first two candidates alias (streams 3 and 4 of the pattern: both queue 3) sets one aside
    monkeypatch.setattr(E, "_made", E._made)
    E._made.clear(); E._made.extend([None] * 2)                                  # the next streams are numbers 3 and 4: queues 3, 3
    pair = E._new_stream_beside("cuda:0", []), None
    pair = (pair[0], E._new_stream_beside("cuda:0", [pair[0]]))
    assert [queue_of(x) for x in pair] == [3, 2] and len(E._stream_rejects) == 1 and queue_of(E._stream_rejects[0]) == 3
    monkeypatch.setattr(E, "_streams_overlap", lambda a, b: False)               # everything aliases: give up after six candidates
    n0 = len(E._made)
    assert E._new_stream_beside("cuda:0", [main]) is E._made[-1] and len(E._made) - n0 == 7
