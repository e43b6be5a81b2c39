"""``torch.ops.fcp.*`` — the PyTorch-ROCm custom-op face of the native library (SURVEY.md 8b / north_star:
"Python host code calling hand-written CDNA4 HIP kernels through PyTorch-ROCm custom ops").

``csrc/libfcp_torch.so`` (built by ``build_native.py`` from ``csrc/torch_ops/fcp_torch_ops.cpp``) registers the ops with
``TORCH_LIBRARY(fcp, ...)`` / ``TORCH_LIBRARY_IMPL(fcp, CUDA, ...)``: tensors in / out, outputs from torch's caching
allocator, work enqueued on the current HIP stream, misuse -> ``RuntimeError`` via ``TORCH_CHECK``.  They are a veneer
over the C ABI of ``include/fcp_hip.h`` (same kernels, same bits).  They are the DEFAULT boundary of the Python host:
the convolution engine, the detector's post-processing and align / crop go through the registered ops whenever
``libfcp_torch.so`` has been built (measured on the MI355X, INTEGRATION.md section 2b: 1.8-1.9 ms of host time per batch-64
detection step through the dispatcher against 1.9-2.0 ms through ctypes, identical step time).  ``FCP_BOUNDARY=ctypes``
calls the C ABI directly instead (also what happens, with a warning, when the veneer was not built — it is the same
kernels either way, never a CPU path); ``FCP_BOUNDARY=torch`` insists on the ops and raises when the library is missing.
Both paths are tested to give identical tensors (``tests/test_torch_ops.py``).
"""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libfcp_torch.so")
_loaded = False


def _default_enabled() -> bool:
    mode = os.environ.get("FCP_BOUNDARY", "auto").lower()
    if mode == "ctypes":
        return False
    if mode == "torch":
        return True
    if os.path.isfile(LIB_PATH):
        return True
    import warnings
    warnings.warn(f"{LIB_PATH} is missing (the torch.ops.fcp veneer was not built): calling the C ABI through ctypes")
    return False


ENABLED = _default_enabled()


def load():
    """Register the ops (once).  Raises when the library has not been built: there is no fallback."""
    global _loaded
    if not _loaded:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(f"torch op library not built: {LIB_PATH} is missing "
                               f"(run `python face-crop-plus_amd/build_native.py`)")
        torch.ops.load_library(LIB_PATH)
        from ._native import ABI_VERSION
        built, running = torch.ops.fcp.abi_version()
        if built != ABI_VERSION or running != ABI_VERSION:
            raise RuntimeError(f"libfcp_torch.so is stale: compiled against ABI {built}, libfcp_hip.so reports {running}, "
                               f"host code expects {ABI_VERSION}; rebuild (python face-crop-plus_amd/build_native.py --force)")
        _loaded = True
    return torch.ops.fcp


OPS = ("conv2d", "conv2d_out", "bottleneck_chain", "retina_decode", "nms_select", "gather_faces", "similarity_from_5pt",
       "warp_affine_u8", "bicubic_down4_round", "parse_argmax_hist")
