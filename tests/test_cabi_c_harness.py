"""The C ABI consumed from plain C (tests/cabi/align_crop_check.c): no Python, no torch on the calling side.

CPU: the header is valid C11 and the program links against libfcp_hip.so.  GPU: the program uploads a seeded case,
runs estimate + warp on its own HIP stream and must reproduce the oracle's ok flags / matrices / crop bytes."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import align_ref as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "align_crop_check.c")


def _build(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    from face_crop_plus_amd import _native as N
    libdir = os.path.dirname(N.LIB_PATH)
    exe = str(tmp_path / "align_crop_check")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "no C compiler"
    cmd = [cc, "-std=c11", "-Wall", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include",
           f"-I{ROOT}/include", SRC, f"-L{libdir}", "-lfcp_hip", f"-L{rocm}/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_program_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libfcp_hip.so" in out and "not found" not in out.split("libfcp_hip.so")[1].splitlines()[0]


def _case(path, rng, n, h, w, f, out_size, border, skew, with_pad):
    imgs = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    pads = np.zeros((n, 4), np.int32)
    if with_pad:
        pads[1] = (6, 5, 0, 0)
        pads[2 % n] = (0, 0, 9, 10)
    tgt = A.landmarks_target(out_size, 0.65)
    idx = rng.integers(0, n, f).astype(np.int32)
    lms = []
    for _ in range(f):
        th, s = rng.uniform(-1.0, 1.0), rng.uniform(0.2, 2.0)
        rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]) * s
        lms.append(tgt @ rot.T + rng.uniform(-40, 90, 2) + rng.normal(0, 1.5, (5, 2)))
    lms = np.stack(lms).astype(np.float32)
    lms[3] = 11.0                                            # degenerate face: ok = 0
    ok = np.zeros(f, np.int32)
    mats = np.zeros((f, 6), np.float64)
    crops = np.zeros((f, out_size[1], out_size[0], 3), np.uint8)
    for i in range(f):
        m = A.estimate_transform(lms[i], tgt, skew)
        if m is None:
            continue
        ok[i], mats[i] = 1, m.reshape(6)
        t, b, l, r = pads[idx[i]]
        crops[i] = A.warp_affine(imgs[idx[i]][t:h - b, l:w - r], m, out_size, A.BORDER[border])
    from face_crop_plus_amd import align
    with open(path, "wb") as fp:
        fp.write(np.array([n, h, w, f, 5, out_size[1], out_size[0], align.border_code(border), int(skew), int(with_pad)],
                          np.int32).tobytes())
        for a in (imgs, idx, lms, tgt.astype(np.float32), pads, ok, crops, mats):
            fp.write(np.ascontiguousarray(a).tobytes())
    return int(ok.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("border,skew,with_pad,out_size", [("constant", False, True, (256, 256)), ("reflect_101", True, False, (120, 90)),
                                                          ("replicate", False, True, (64, 200))])
def test_c_program_reproduces_oracle_bytes(tmp_path, border, skew, with_pad, out_size):
    exe = _build(tmp_path)
    rng = np.random.default_rng(17)
    case = str(tmp_path / "case.bin")
    live = _case(case, rng, 3, 150, 210, 14, out_size, border, skew, with_pad)
    r = subprocess.run([exe, case], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith(f"OK faces=14 live={live} ")
