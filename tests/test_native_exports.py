"""The C-ABI library builds, loads and exports every symbol include/fcp_hip.h declares
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    import __graft_entry__ as ge
    ge.build()
    from face_crop_plus_amd import _native as N
    assert os.path.isfile(N.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "fcp_hip.h")).read()
    declared = set(re.findall(r"\b(fcp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fcp_last_error"} - declared
    lib = ctypes.CDLL(N.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"symbols declared in fcp_hip.h but not exported: {missing}"
    assert set(N.EXPORTS) <= declared | {"fcp_abi_version", "fcp_last_error"}
    lib.fcp_abi_version.restype = ctypes.c_int
    assert lib.fcp_abi_version() == N.ABI_VERSION


def test_conv_desc_layout_matches_header():
    """ctypes mirror and the C struct must agree field by field."""
    from face_crop_plus_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "fcp_hip.h")).read()
    body = hdr[hdr.index("typedef struct fcp_conv_desc {"):hdr.index("} fcp_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = re.sub(r"^(const\s+)?(float|int32_t|void)\s*\*?", "", decl)
        fields += [n.strip().lstrip("*") for n in names.split(",")]
    mine = [f[0].rstrip("_") for f in N.ConvDesc._fields_]
    assert fields == mine
    assert ctypes.sizeof(N.ConvDesc) == 8 * 8 + 40 * 4      # 8 pointers + 39 int32 / float fields (+ 4 bytes of tail padding)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from face_crop_plus_amd import _native as N
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        N.lib()


def test_graft_entry_build_in_a_fresh_interpreter():
    """build() must work when __graft_entry__ is the first thing imported (no stray importlib.util import)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); print('ok')"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: importing every product module (incl. the CLI) must not pull it in, and
    no product source may mention it."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, importlib\n"
            "for m in ('cropper','retinaface','rrdb','bise','align','batch','utils','engine','weights','dist','__main__','_native'):\n"
            "    importlib.import_module('face_crop_plus_amd.' + m)\n"
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]\n"
            "assert not bad, bad\nprint('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
    pkg = os.path.join(root, "face-crop-plus_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
