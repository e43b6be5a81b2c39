"""Decode / encode worker PROCESSES behind ``Cropper.process_dir`` (reference ``utils.py:228-271``
read side, ``cropper.py:554-609`` write side).

Round 3 ran Pillow on an I/O thread pool: 1300-1440 images/s end to end against 2600 for the device path alone — the
rest was Python under the GIL (array conversion, EXIF handling, encoder set-up, executor bookkeeping).  Here every I/O
thread of the executor owns ONE worker process and does a blocking request / reply with it, so the thread architecture
of ``process_dir`` (prefetch depth, back-pressure, error surfacing, file naming, warn-and-skip) is unchanged while the
CPU-heavy part runs outside the parent's interpreter:

* read:  the worker decodes the file (``_io_codec.read_image``: same EXIF / RGB rules) into its own ring of shared
  memory (a ``memfd`` mapped by both sides: no /dev/shm quota, no pickling of pixels) and replies ``(offset, shape)``;
  the parent wraps the region as a numpy view and gives it back (``release``) once the batch that used it is done.
  A worker NEVER waits for ring space — an image that does not fit right now travels through the socket instead — so a
  ring that is too small for a batch of 4K frames costs speed, not progress.
* write: the parent sends the crop's bytes through the socket (a syscall, GIL released), the worker encodes with
  ``_io_codec.write_image`` (same encoder table, same warn-and-skip) and replies.

Workers are fresh interpreters (``python -m face_crop_plus_amd._io_pool`` with the ring / control / socket descriptors
passed explicitly): they import numpy + Pillow only.  They are deliberately NOT forked from the parent: a fork of a
process that holds HIP streams, events and pinned buffers in other threads' thread-local storage destroys those objects
in the child (CPython clears the dead threads' state after fork), i.e. calls into a HIP runtime that does not exist
there (measured: "terminate called without an active exception" with two GPU worker threads).
``FCP_IO_PROCESSES=0`` keeps everything on threads (the round-3 behaviour).
"""
from __future__ import annotations

import mmap
import os
import sys
import threading
import warnings
from collections import OrderedDict

import numpy as np

RING_MB = int(os.environ.get("FCP_IO_RING_MB", "128"))


def _serve(conn, ring, ring_bytes: int, ctl, slot: int):
    """Worker process: serve requests until the socket closes."""
    from ._io_codec import read_image, write_image
    consumed = np.frombuffer(ctl, dtype=np.int64)          # consumed[slot]: bytes the parent has given back (monotonic)
    produced = 0                                            # bytes handed out so far, incl. skipped ring tails (monotonic)
    while True:
        try:
            msg = conn.recv()
        except (EOFError, OSError):
            return
        if msg is None:
            return
        kind = msg[0]
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                if kind == "read":                          # one request for a list of files, ONE reply for all of them
                    replies, through_socket = [], []
                    for path in msg[1]:
                        seen = len(caught)
                        img = read_image(path)
                        notes = [str(w.message) for w in caught[seen:]]
                        if img is None:
                            replies.append(("none", notes))
                            continue
                        need = img.nbytes
                        pos = produced % ring_bytes if ring_bytes else 0
                        skip = ring_bytes - pos if ring_bytes and pos + need > ring_bytes else 0
                        if ring_bytes and need <= ring_bytes and produced + skip + need - int(consumed[slot]) <= ring_bytes:
                            off = (pos + skip) % ring_bytes
                            np.frombuffer(ring, dtype=np.uint8, count=need, offset=off)[:] = img.reshape(-1)
                            produced += skip + need
                            replies.append(("ring", off, img.shape, skip + need, notes))
                        else:                               # no room right now: through the socket, never wait
                            replies.append(("pipe", img.shape, notes))
                            through_socket.append(img)
                    conn.send(("read", replies))
                    for img in through_socket:
                        conn.send_bytes(memoryview(np.ascontiguousarray(img)).cast("B"))
                elif kind == "write":
                    _, path, shape = msg
                    pixels = np.frombuffer(conn.recv_bytes(), dtype=np.uint8).reshape(shape)
                    ok = write_image(path, pixels)
                    conn.send(("done", bool(ok), [str(w.message) for w in caught]))
                else:
                    conn.send(("error", f"unknown request {kind!r}"))
        except Exception as e:                              # noqa: BLE001 - reported to the parent, which re-raises
            conn.send(("error", f"{type(e).__name__}: {e}"))


def _main(argv):
    """``python -m face_crop_plus_amd._io_pool <sock_fd> <ring_fd | -1> <ring_bytes> <ctl_fd> <ctl_bytes> <slot>``"""
    from multiprocessing.connection import Connection
    sock_fd, ring_fd, ring_bytes, ctl_fd, ctl_bytes, slot = (int(a) for a in argv)
    ring = mmap.mmap(ring_fd, ring_bytes) if ring_fd >= 0 else None
    ctl = mmap.mmap(ctl_fd, ctl_bytes)
    try:
        _serve(Connection(sock_fd), ring, ring_bytes, ctl, slot)
    finally:
        os._exit(0)


class _Worker:
    """Parent-side handle of one worker process; used by exactly one I/O thread at a time."""

    def __init__(self, ctl_fd, ctl_bytes, ctl, slot, ring_bytes, register=None):
        import socket
        import subprocess
        from multiprocessing.connection import Connection
        self.ring_bytes = ring_bytes
        self.ring, ring_fd = None, -1
        if ring_bytes:
            ring_fd = os.memfd_create(f"fcp-io-ring-{slot}")
            os.ftruncate(ring_fd, ring_bytes)
            self.ring = mmap.mmap(ring_fd, ring_bytes)
        mine, theirs = socket.socketpair()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # the directory holding the import shim
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        fds = [theirs.fileno(), ctl_fd] + ([ring_fd] if ring_fd >= 0 else [])
        self.proc = subprocess.Popen([sys.executable, "-m", "face_crop_plus_amd._io_pool", str(theirs.fileno()), str(ring_fd),
                                      str(ring_bytes), str(ctl_fd), str(ctl_bytes), str(slot)], pass_fds=fds, env=env,
                                     stdin=subprocess.DEVNULL)
        theirs.close()
        if ring_fd >= 0:
            os.close(ring_fd)                       # the mappings keep the memory alive
        self.conn = Connection(mine.detach())
        self.slot, self.ctl = slot, np.frombuffer(ctl, dtype=np.int64)
        self.pinned = False
        if self.ring is not None and register is not None:
            self.pinned = register(self.ring, ring_bytes)     # page-locked for HIP: uploads DMA straight out of the ring
            self._unregister = register
        self.lock = threading.Lock()
        self.regions = OrderedDict()                 # seq -> [bytes, released]; the released PREFIX is given back
        self.seq = 0
        self.given_back = 0

    def _io(self, fn, *args):
        """Every socket operation of the parent goes through here: whichever side of a request the worker's death falls on —
        the send (``BrokenPipeError`` / ``ConnectionResetError``), the reply (``EOFError``) or a payload that follows it — the
        caller sees the one documented error."""
        try:
            return fn(*args)
        except (EOFError, OSError) as e:
            raise RuntimeError(f"I/O worker process {self.proc.pid} died ({type(e).__name__})") from e

    def _reply(self):
        rep = self._io(self.conn.recv)
        if rep[0] == "error":
            raise RuntimeError(f"I/O worker: {rep[1]}")
        return rep

    def read_many(self, paths):
        """-> [(RGB uint8 HWC array or None, release token or None)] in the order of ``paths``: one request, one reply.
        Warnings of the decoder are re-issued here."""
        self._io(self.conn.send, ("read", list(paths)))
        out = []
        for rep in self._reply()[1]:
            for note in rep[-1]:
                warnings.warn(note)
            if rep[0] == "none":
                out.append((None, None))
            elif rep[0] == "pipe":
                out.append((rep[1], "pipe"))                 # payload follows the reply, in order
            else:
                _, off, shape, nbytes, _ = rep
                arr = np.frombuffer(self.ring, dtype=np.uint8, count=int(np.prod(shape)), offset=off).reshape(shape)
                with self.lock:
                    self.seq += 1
                    self.regions[self.seq] = [nbytes, False]
                    out.append((arr, (self, self.seq)))
        for k, (shape, tok) in enumerate(out):
            if tok == "pipe":
                buf = bytearray(int(np.prod(shape)))
                self._io(self.conn.recv_bytes_into, buf)
                out[k] = (np.frombuffer(buf, dtype=np.uint8).reshape(shape), None)
        return out

    def read(self, path):
        return self.read_many([path])[0]

    @staticmethod
    def is_pinned(token) -> bool:
        return token is not None and token[0].pinned

    def release(self, seq):
        with self.lock:
            self.regions[seq][1] = True
            while self.regions:
                first = next(iter(self.regions))
                if not self.regions[first][1]:
                    break
                self.given_back += self.regions.pop(first)[0]
            self.ctl[self.slot] = self.given_back    # one aligned 8-byte store: the worker only ever reads it

    def write(self, path, pixels: np.ndarray) -> bool:
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
        self._io(self.conn.send, ("write", path, pixels.shape))
        self._io(self.conn.send_bytes, memoryview(pixels).cast("B"))
        rep = self._reply()
        for note in rep[2]:
            warnings.warn(note)
        return rep[1]

    def close(self):
        try:
            self.conn.send(None)
        except (OSError, ValueError):
            pass
        try:
            self.proc.wait(timeout=2)
        except Exception:                            # noqa: BLE001 - subprocess.TimeoutExpired
            self.proc.kill()
            self.proc.wait(timeout=2)
        self.conn.close()
        if self.ring is not None:
            if self.pinned:
                self._unregister(self.ring, 0)
                self.pinned = False
            try:
                self.ring.close()
            except BufferError:                      # a numpy view of a region is still alive somewhere: leave it to the GC
                pass


class IOProcesses:
    """``readers`` decode workers (each with a ring) + ``writers`` encode workers; I/O threads borrow one each through
    ``read()`` / ``write()`` (thread-local, so a worker's socket is only ever used by one thread)."""

    def __init__(self, readers: int, writers: int, ring_mb: int = RING_MB, register=None):
        """``register(mmap, nbytes) -> bool``: optional hook that page-locks a ring for the GPU runtime (nbytes == 0:
        undo it); this module itself stays free of torch / HIP imports."""
        if not hasattr(os, "memfd_create"):
            raise OSError("os.memfd_create is unavailable on this platform")
        n = readers + writers
        self._ctl_bytes = max(mmap.PAGESIZE, 8 * n)
        ctl_fd = os.memfd_create("fcp-io-ctl")
        os.ftruncate(ctl_fd, self._ctl_bytes)
        self._ctl = mmap.mmap(ctl_fd, self._ctl_bytes)
        self._readers, self._writers = [], []
        try:
            for i in range(readers):                 # Popen returns at once: the interpreters start in parallel
                self._readers.append(_Worker(ctl_fd, self._ctl_bytes, self._ctl, i, ring_mb << 20, register))
            for i in range(writers):
                self._writers.append(_Worker(ctl_fd, self._ctl_bytes, self._ctl, readers + i, 0))
        finally:
            os.close(ctl_fd)
        self._lock = threading.Lock()
        self.closed = False
        self.begin()

    @property
    def readers(self):
        return len(self._readers)

    @property
    def writers(self):
        return len(self._writers)

    def begin(self):
        """Start of a ``process_dir`` run: its (new) I/O threads claim workers afresh."""
        self._free_r, self._free_w = list(self._readers), list(self._writers)
        self._tls = threading.local()

    def healthy(self) -> bool:
        """Whether the pool can serve another run: every worker process is alive and no ring region of an earlier run is still
        outstanding.  A run that failed half-way (a worker OOM-killed, an error reply in the middle of a request, prefetched
        batches that were read but never collected) leaves dead processes or regions nobody will give back — such a pool is
        closed and replaced by its owner (``Cropper._io_processes``), never reused: a leaked region would push that decoder to
        the socket path for good, a dead one would fail every later run."""
        if self.closed:
            return False
        for w in self._readers + self._writers:
            if w.proc.poll() is not None:
                return False
            with w.lock:
                if w.regions:
                    return False
        return True

    def _mine(self, name, free):
        w = getattr(self._tls, name, None)
        if w is None:
            with self._lock:
                if not free:
                    raise RuntimeError("more I/O threads than worker processes")
                w = free.pop()
            setattr(self._tls, name, w)
        return w

    def read(self, path):
        return self._mine("r", self._free_r).read(path)

    def read_many(self, paths):
        return self._mine("r", self._free_r).read_many(paths)

    def write(self, path, pixels):
        return self._mine("w", self._free_w).write(path, pixels)

    @staticmethod
    def pinned_flags(tokens):
        return [_Worker.is_pinned(t) for t in tokens]

    @staticmethod
    def release(tokens):
        for tok in tokens:
            if tok is not None:
                tok[0].release(tok[1])

    def close(self):
        if not self.closed:
            self.closed = True
            for w in self._readers + self._writers:
                w.close()

    def __del__(self):
        try:
            self.close()
        except Exception:                            # noqa: BLE001 - interpreter shutdown
            pass


if __name__ == "__main__":
    _main(sys.argv[1:])
