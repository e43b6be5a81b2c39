"""Per-stage profiler ranges (the reference has no tracing; rocprofv3 traces of this build are
otherwise readable by kernel name only).  ``with trace.range("detect"): ...`` pushes a ROCTx range that
``rocprofv3 --marker-trace --kernel-trace`` records next to the kernels.  Ranges are host-side brackets around
the *enqueue* of a stage's kernels (nothing synchronises); correlate with the kernel trace by timestamps.

The ROCTx library (``librocprofiler-sdk-roctx.so``, else the roctracer ``libroctx64.so``) is looked up once;
without it — or with FCP_TRACE=0 — ranges are no-ops: tracing is an observer, never part of the data path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

_push = _pop = None
_tried = False


def _load():
    global _push, _pop, _tried
    _tried = True
    if os.environ.get("FCP_TRACE", "1") == "0":
        return
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        for path in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", name), name):
            try:
                lib = C.CDLL(path)
                push, pop = lib.roctxRangePushA, lib.roctxRangePop
                push.argtypes, push.restype = [C.c_char_p], C.c_int
                pop.argtypes, pop.restype = [], C.c_int
                _push, _pop = push, pop
                return
            except (OSError, AttributeError):
                continue


def available() -> bool:
    if not _tried:
        _load()
    return _push is not None


@contextlib.contextmanager
def range(name: str):          # noqa: A001 - mirrors roctxRange / nvtx.range
    if not _tried:
        _load()
    if _push is None:
        yield
        return
    _push(name.encode())
    try:
        yield
    finally:
        _pop()
