"""ctypes binding of the C ABI in ``include/fcp_hip.h`` (``csrc/libfcp_hip.so``).

There is deliberately no fallback: if the HIP library is missing or a call
fails, a RuntimeError is raised — the product path never runs on a CPU or an
eager-PyTorch substitute.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libfcp_hip.so")
ABI_VERSION = 15

c_f32p = C.c_void_p
_lib = None


class ConvDesc(C.Structure):
    """Mirror of ``fcp_conv_desc``."""
    _fields_ = [
        ("in_", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("res1", C.c_void_p), ("res2", C.c_void_p), ("wscale", C.c_void_p), ("in2", C.c_void_p),
        ("n", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("cin", C.c_int32), ("in_ld", C.c_int32), ("in_up2", C.c_int32),
        ("cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32), ("out_ld", C.c_int32),
        ("tile_n", C.c_int32), ("cin4", C.c_int32),
        ("act_slope", C.c_float), ("alpha", C.c_float), ("alpha2", C.c_float),
        ("res1_pre", C.c_int32), ("res1_ld", C.c_int32), ("res1_h", C.c_int32),
        ("res1_w", C.c_int32), ("res2_ld", C.c_int32), ("precision", C.c_int32),
        ("in_fmt", C.c_int32), ("out_fmt", C.c_int32), ("res1_fmt", C.c_int32), ("res2_fmt", C.c_int32),
        ("tile_m", C.c_int32), ("cin2", C.c_int32), ("in2_ld", C.c_int32), ("in2_h", C.c_int32),
        ("in2_w", C.c_int32), ("in2_stride", C.c_int32), ("flags", C.c_int32), ("cu_budget", C.c_int32),
        ("band_top", C.c_int32), ("band_bottom", C.c_int32),
    ]


CONV_FLAT_ADDR = 1      # fcp_conv_desc.flags: FCP_CONV_FLAT_ADDR
CONV_BALANCE_TAIL = 2   # fcp_conv_desc.flags: FCP_CONV_BALANCE_TAIL
CHAIN_OUT_EVEN_ONLY = 1  # fcp_chain_desc.flags: FCP_CHAIN_OUT_EVEN_ONLY


class ChainDesc(C.Structure):
    """Mirror of ``fcp_chain_desc``."""
    _fields_ = [(k, C.c_void_p) for k in ("t1", "w2", "ws2", "b2", "w3", "ws3", "b3", "res", "out", "w1n", "ws1n",
                                          "b1n", "t1n", "t1b")] + \
               [(k, C.c_int32) for k in ("n", "h", "w", "c", "cn", "t1_ld", "res_ld", "out_ld", "t1n_ld", "nout", "tile_m", "flags",
                                         "cb", "t1b_ld", "t1b_h", "t1b_w", "t1b_stride")]


# name -> argtypes; every function returns int (0 = ok)
_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
SIGNATURES = {
    "fcp_conv2d_nhwc_f32": [C.POINTER(ConvDesc), _P],
    "fcp_bottleneck_chain_f16x3": [C.POINTER(ChainDesc), _P],
    "fcp_u8_to_nhwc4_f32": [_P, _P, _L, C.POINTER(C.c_float), _F, _P],
    "fcp_f32nchw_to_nhwc4_f32": [_P, _P, _I, _I, _I, C.POINTER(C.c_float), _F, _P],
    "fcp_maxpool3x3s2_nhwc_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "fcp_maxpool3x3s2_split32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "fcp_stem7x7s2_relu_pool_u8": [_P, _I, _I, _I, C.POINTER(C.c_int32), _P, _P, _P, _P, _I, _I, _P],
    "fcp_stem7x7s2_relu_pool_conv1_u8": [_P, _I, _I, _I, C.POINTER(C.c_int32), _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P],
    "fcp_stem7x7s2_relu_pool_f32": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P],
    "fcp_f32_to_split32": [_P, _P, _L, _I, _P],
    "fcp_split32_to_f32": [_P, _P, _L, _I, _P],
    "fcp_absmax_nhwc": [_P, _L, _I, _I, _I, _P, _P],
    "fcp_retina_decode": [_P, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "fcp_retina_nms_select": [_P, _P, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P],
    "fcp_retina_gather_faces": [_P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _P],
    "fcp_estimate_transform": [_P, _P, _I, _I, _I, _P, _P, _P],
    "fcp_estimate_transform_counted": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    "fcp_bise_preprocess_u8": [_P, _I, _I, _I, _P, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P],
    "fcp_avgpool_nhwc_f32": [_P, _I, _I, _I, _I, _P, _P],
    "fcp_fc_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "fcp_scale_add_nhwc_f32": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P],
    "fcp_parse_tail": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "fcp_label_mask_u8": [_P, _L, C.c_uint32, _P, _P],
    "fcp_bicubic_down4_u8": [_P, _I, _I, _I, _P, _P],
    "fcp_warp_affine_u8": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "fcp_build_batch_u8": [_P, _L, _P, _P, _I, _I, _I, _I, _P, _P],
}
EXPORTS = ["fcp_abi_version", "fcp_last_error", "fcp_retina_nms_workspace_bytes"] + list(SIGNATURES)


def lib():
    """Load (once) and return the shared library; raise loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"HIP extension not built: {LIB_PATH} is missing. Run "
            f"`python face-crop-plus_amd/build_native.py` (needs hipcc). There is no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    l.fcp_abi_version.restype = C.c_int
    l.fcp_last_error.restype = C.c_char_p
    if l.fcp_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libfcp_hip.so ABI {l.fcp_abi_version()} != expected {ABI_VERSION}; rebuild")
    l.fcp_retina_nms_workspace_bytes.argtypes = [C.c_int, C.c_int]
    l.fcp_retina_nms_workspace_bytes.restype = C.c_int64
    for name, args in SIGNATURES.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = l
    return l


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().fcp_last_error().decode(errors="replace")
        raise RuntimeError(f"{what or 'fcp_hip'} failed ({rc}): {msg}")


def stream_ptr():
    """Current torch HIP stream as a raw hipStream_t."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, byte_offset: int = 0):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr() + byte_offset)
