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


def test_candidates_that_alias_a_chosen_stream_are_set_aside(E, monkeypatch):
    """The hardware-queue probe: a candidate that does not overlap with every chosen stream is kept alive but not used; after
    six candidates the last one is taken as it is."""
    monkeypatch.setattr(E, "STREAM_PROBE", True)
    monkeypatch.setattr(E, "_stream_rejects", [])
    import torch
    monkeypatch.setattr(torch.cuda, "_sleep", lambda c: None, raising=False)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    order = {}
    queue_of = lambda s_: order.setdefault(id(s_), [1, 2, 3, 3, 2, 1, 0, 3][len(order) % 8])     # the measured assignment pattern
    monkeypatch.setattr(E, "_streams_overlap", lambda a, b: queue_of(a) != queue_of(b))
    main = E.thread_main_stream("cuda:0")                 # queue 1
    side = E.thread_side_streams("cuda:0", 2)             # queues 2 and 3; the 4th stream (queue 3 again) is never asked for
    assert [queue_of(x) for x in [main] + side] == [1, 2, 3] and not E._stream_rejects
    more = E.thread_side_streams("cuda:0", 3)             # 3 streams beside `main` on 4 queues: 3 (alias of nothing chosen), then 2, then 0
    assert len({queue_of(x) for x in more} | {queue_of(main)}) == 4 and len(E._stream_rejects) >= 1
    monkeypatch.setattr(E, "_streams_overlap", lambda a, b: False)               # everything aliases: give up after six candidates
    n0 = len(E._made)
    assert E._new_stream_beside("cuda:0", [main]) is E._made[-1] and len(E._made) - n0 == 7
