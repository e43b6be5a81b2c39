"""Compile csrc/*.hip for gfx950 into csrc/libfcp_hip.so (in-tree, so the built
library travels with the repo snapshot to the GPU box).

    python face-crop-plus_amd/build_native.py [--force]

hipcc cross-compiles without a GPU.  ``-ffp-contract=off`` keeps every float /
double operation separately rounded, which the decode / NMS / warpAffine
kernels rely on to reproduce the reference's op-by-op arithmetic.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libfcp_hip.so")

ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}"]
FLAGS += [f"-D{d}" for d in os.environ.get("FCP_BUILD_DEFINES", "").split()]   # experiment builds on the GPU box only
FLAGS += os.environ.get("FCP_BUILD_FLAGS", "").split()                          # e.g. -include tools/probes/fcp_no_mfma.h


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isfile(c) or c == "hipcc"):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, deps):
    if not os.path.isfile(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if not force and _newer(obj, [src] + headers):
            continue
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for: {failed}")
    if force or procs or not _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # The TORCH_LIBRARY veneer is optional (default boundary: see torch_ops.py): a host without a matching g++ / torch
    # headers must still get the C-ABI library; torch_ops.load() raises lazily when the file is missing.
    try:
        build_torch_ops(force, verbose)
    except Exception as e:                                  # noqa: BLE001 - compiler / header / ABI problems alike
        import warnings
        # a veneer left over from an earlier build would become the default boundary (torch_ops auto mode) against a
        # library whose structs may have changed: remove it, so that auto mode falls back to ctypes with its warning
        if os.path.isfile(TORCH_LIB):
            os.remove(TORCH_LIB)
        warnings.warn(f"libfcp_torch.so (torch.ops.fcp veneer) was not built: {e}")
    return LIB


TORCH_LIB = os.path.join(CSRC, "libfcp_torch.so")
TORCH_SRC = os.path.join(CSRC, "torch_ops", "fcp_torch_ops.cpp")


def build_torch_ops(force: bool = False, verbose: bool = True) -> str:
    """csrc/torch_ops/fcp_torch_ops.cpp -> csrc/libfcp_torch.so: the TORCH_LIBRARY(fcp) registration of the ops over the
    C ABI (host C++ only: g++ against the installed torch headers / libraries, linked to libfcp_hip.so via $ORIGIN)."""
    headers = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    if not force and _newer(TORCH_LIB, [TORCH_SRC, LIB] + headers):
        return TORCH_LIB
    import torch
    tl = os.path.dirname(torch.__file__)
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}", f"-I{tl}/include", f"-I{tl}/include/torch/csrc/api/include", f"-I{rocm}/include",
           f"-I{INCLUDE}", TORCH_SRC, "-o", TORCH_LIB, f"-L{tl}/lib", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip",
           f"-L{CSRC}", "-lfcp_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tl}/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
