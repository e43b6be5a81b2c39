"""Decode / encode worker processes of process_dir (face-crop-plus_amd/_io_pool.py): host logic, no GPU."""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest


def _img(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.fixture
def pool():
    from face_crop_plus_amd._io_pool import IOProcesses
    p = IOProcesses(readers=2, writers=1, ring_mb=1)
    yield p
    p.close()


def test_round_trip_ring_wrap_pipe_fallback_and_release(pool, tmp_path):
    from PIL import Image
    # 300 KB images against a 1 MiB ring: three fit, the fourth goes through the pipe until regions are released
    imgs = [_img(320, 320, s) for s in range(8)]
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(tmp_path / f"{i}.png")
    got = [pool.read(str(tmp_path / f"{i}.png")) for i in range(5)]           # one thread = one worker
    assert all(np.array_equal(g[0], imgs[i]) for i, g in enumerate(got))
    in_ring = [tok is not None for _, tok in got]
    assert in_ring == [True, True, True, False, False]
    pool.release([tok for _, tok in got[1:3]])                                 # out of order: nothing is given back yet
    more, tok = pool.read(str(tmp_path / "5.png"))
    assert tok is None and np.array_equal(more, imgs[5])
    pool.release([got[0][1]])                                                  # the prefix is complete: all three regions free
    del got
    for i in (6, 7, 0):                                                        # the ring wraps (tail skipped) and keeps going
        arr, tok = pool.read(str(tmp_path / f"{i}.png"))
        assert tok is not None and np.array_equal(arr, imgs[i])
    # writes: same bytes as the in-process encoder, warnings and the skip rule travel back
    from face_crop_plus_amd.utils import write_image
    assert pool.write(str(tmp_path / "w.png"), imgs[0]) is True
    write_image(str(tmp_path / "ref.png"), imgs[0])
    assert (tmp_path / "w.png").read_bytes() == (tmp_path / "ref.png").read_bytes()
    assert pool.write(str(tmp_path / "m.jpg"), imgs[1][..., 0].copy()) is True   # single-channel mask
    with pytest.warns(UserWarning, match="Could not write"):
        assert pool.write(str(tmp_path / "w.xyz"), imgs[0]) is False
    with pytest.raises(RuntimeError, match="I/O worker"):
        pool.write(str(tmp_path / "no_dir" / "w.ppm"), imgs[0])                   # a real I/O error is raised in the parent
    assert pool.write(str(tmp_path / "after.png"), imgs[2]) is True               # ... and the worker lives on


def test_unreadable_file_warns_and_threads_get_their_own_worker(pool, tmp_path):
    from PIL import Image
    (tmp_path / "bad.png").write_bytes(b"nope")
    with pytest.warns(UserWarning, match="Could not read"):
        assert pool.read(str(tmp_path / "bad.png")) == (None, None)
    for i in range(6):
        Image.fromarray(_img(40, 50, i)).save(tmp_path / f"t{i}.png")
    pool.begin()
    seen = set()

    def job(i):
        arr, tok = pool.read(str(tmp_path / f"t{i}.png"))
        seen.add((threading.get_ident(), id(tok[0])))
        ok = np.array_equal(arr, _img(40, 50, i))
        pool.release([tok])
        return ok
    with ThreadPoolExecutor(2) as ex:
        assert all(ex.map(job, range(6)))
    assert len({w for _, w in seen}) <= 2 and len({t: w for t, w in seen}) == len({t for t, _ in seen})   # a thread keeps its worker
    pool.begin()
    with ThreadPoolExecutor(3) as ex:                                           # more threads than readers: the third fails loudly
        barrier = threading.Barrier(3)
        def claim(i):
            barrier.wait()
            return pool.read(str(tmp_path / f"t{i}.png"))
        futs = [ex.submit(claim, i) for i in range(3)]
        errs = [f.exception() for f in futs]
    assert sum(e is not None for e in errs) == 1 and "more I/O threads" in str([e for e in errs if e][0])


def test_workers_exit_with_the_pool(tmp_path):
    from face_crop_plus_amd._io_pool import IOProcesses
    p = IOProcesses(1, 1, ring_mb=1)
    pids = [w.proc.pid for w in p._readers + p._writers]
    assert p.write(str(tmp_path / 'x.png'), _img(8, 8, 0)) is True
    p.close()
    for pid in pids:
        with pytest.raises(OSError):
            os.kill(pid, 0)                                                     # gone (and reaped)


def test_read_many_one_reply_keeps_order_with_unreadable_and_oversize_files(pool, tmp_path):
    from PIL import Image
    big = _img(700, 700, 1)                                                     # 1.47 MB > the 1 MiB ring: through the socket
    imgs = {"a.png": _img(100, 120, 2), "big.png": big, "b.png": _img(90, 80, 3), "c.png": _img(64, 64, 4)}
    for name, im in imgs.items():
        Image.fromarray(im).save(tmp_path / name)
    (tmp_path / "bad.png").write_bytes(b"nope")
    order = ["a.png", "big.png", "bad.png", "b.png", "c.png"]
    with pytest.warns(UserWarning, match="Could not read"):
        got = pool.read_many([str(tmp_path / n) for n in order])
    assert [g[0] is None for g in got] == [False, False, True, False, False]
    for name, (arr, tok) in zip(order, got):
        if name != "bad.png":
            assert np.array_equal(arr, imgs[name]), name
    assert [tok is not None for _, tok in got] == [True, False, False, True, True]
    pool.release([tok for _, tok in got])


def test_workers_exit_when_the_parent_dies_without_closing(tmp_path):
    """Workers hold only their own descriptors (subprocess + pass_fds): when the parent is gone — killed, crashed — their
    socket reads end and they leave; no orphan decoder keeps a box busy."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from face_crop_plus_amd._io_pool import IOProcesses\n"
            "p = IOProcesses(2, 1, ring_mb=1)\n"
            "print(' '.join(str(w.proc.pid) for w in p._readers + p._writers), flush=True)\n"
            "os._exit(0)\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    pids = [int(x) for x in out.stdout.split()]
    assert len(pids) == 3, out.stderr
    deadline = time.time() + 10
    alive = pids
    while alive and time.time() < deadline:
        alive = [pid for pid in alive if os.path.exists(f"/proc/{pid}") and "Z" not in open(f"/proc/{pid}/stat").read().split(")")[1].split()[0]]
        time.sleep(0.1)
    assert not alive, f"worker processes {alive} survived their parent"


def test_unhealthy_pool_is_detected(tmp_path):
    """A pool with a dead worker or with ring regions an aborted run never gave back must not be reused
    (Cropper._io_processes replaces it): ``healthy()`` is the test."""
    from PIL import Image
    from face_crop_plus_amd._io_pool import IOProcesses
    Image.fromarray(_img(64, 64, 1)).save(tmp_path / "a.png")
    p = IOProcesses(2, 1, ring_mb=1)
    try:
        assert p.healthy()
        arr, tok = p.read(str(tmp_path / "a.png"))
        assert tok is not None and not p.healthy()                  # a region is outstanding (a prefetched, uncollected batch)
        p.release([tok])
        assert p.healthy()
        victim = p._writers[0].proc
        victim.kill()
        victim.wait(timeout=5)
        assert not p.healthy()                                      # a dead encoder
    finally:
        p.close()
    assert not p.healthy()                                          # closed


@pytest.mark.parametrize("when", ["before_send", "between_send_and_reply"])
@pytest.mark.parametrize("op", ["read", "write"])
def test_worker_death_is_one_documented_error_whichever_side_it_falls_on(tmp_path, when, op):
    """A dead worker surfaces as RuntimeError("I/O worker process <pid> died ...") from every parent-side socket operation:
    the send (BrokenPipeError / ConnectionResetError when the process is already gone) as well as the reply (EOFError when it
    dies with the request in its socket buffer).  Both orderings are made deterministic here: 'before' waits for the
    process to be reaped, 'between' stops the worker (SIGSTOP), queues the request, then kills it."""
    import signal
    from PIL import Image
    from face_crop_plus_amd._io_pool import IOProcesses
    Image.fromarray(_img(32, 32, 1)).save(tmp_path / "a.png")
    p = IOProcesses(1, 1, ring_mb=1)
    try:
        w = (p._readers if op == "read" else p._writers)[0]
        call = (lambda: p.read(str(tmp_path / "a.png"))) if op == "read" else (lambda: p.write(str(tmp_path / "o.png"), _img(8, 8, 2)))
        call()                                                      # the worker is up and serving
        if op == "read":
            p.release([tok for _, tok in [p.read(str(tmp_path / "a.png"))]])
        if when == "before_send":
            w.proc.kill()
            w.proc.wait(timeout=5)
            with pytest.raises(RuntimeError, match=f"I/O worker process {w.proc.pid} died"):
                call()
        else:
            os.kill(w.proc.pid, signal.SIGSTOP)
            killer = threading.Timer(0.3, w.proc.kill)              # the request is in the socket buffer by then
            killer.start()
            try:
                with pytest.raises(RuntimeError, match=f"I/O worker process {w.proc.pid} died"):
                    call()
            finally:
                killer.cancel()
                w.proc.wait(timeout=5)
        assert not p.healthy()
    finally:
        p.close()
