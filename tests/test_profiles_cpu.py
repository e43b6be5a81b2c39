"""Evidence housekeeping: a `roofline.traffic` figure quoted from profiles/ must have been recorded at the
ABI version of the library it is quoted for, and bench.py refuses (names, but does not quote) one that was not."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _newest(key):
    pdir = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(pdir) if f.endswith(f"_{key}.json"))
    return os.path.join(pdir, files[-1]) if files else None


def test_committed_traffic_files_match_the_current_abi():
    import bench
    from face_crop_plus_amd._native import ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "fcp_hip.h")).read()
    assert f"#define FCP_ABI_VERSION {ABI_VERSION}" in hdr
    missing = []
    for key in bench.TRAFFIC_KEYS:
        path = _newest(key)
        assert path is not None, f"profiles/ holds no *_{key}.json (bash tools/profile_round.sh <round> regenerates it)"
        rec = json.load(open(path))
        if rec.get("abi_version") != ABI_VERSION:
            missing.append(f"{os.path.basename(path)}: recorded at ABI {rec.get('abi_version', 'unknown')}, library is {ABI_VERSION}")
        else:
            assert rec["launches_per_step"] > 0 and rec["hbm_bytes_per_launch"] > 0
    assert not missing, "stale traffic evidence (run `bash tools/profile_round.sh <round>` on the GPU box and commit the files): " + "; ".join(missing)


def test_bench_refuses_to_quote_stale_traffic(tmp_path, monkeypatch):
    import bench
    from face_crop_plus_amd._native import ABI_VERSION
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec = {"hbm_bytes_per_launch": 1.0e9, "abi_version": ABI_VERSION, "launches_per_step": 52}
    (prof / "r09_c3det_pmc.json").write_text(json.dumps(rec))
    assert bench._pmc_traffic("c3det_pmc", 52) == (1000000000, "profiles/r09_c3det_pmc.json", None)
    t, src, why = bench._pmc_traffic("c3det_pmc", 50)                      # the step has changed since
    assert t is None and src == "profiles/r09_c3det_pmc.json" and "52 conv launches" in why
    (prof / "r10_c3det_pmc.json").write_text(json.dumps(dict(rec, abi_version=ABI_VERSION - 1)))   # newer file, older build
    t, src, why = bench._pmc_traffic("c3det_pmc", 52)
    assert t is None and src.endswith("r10_c3det_pmc.json") and f"ABI {ABI_VERSION - 1}" in why
    (prof / "r11_c3det_pmc.json").write_text(json.dumps({"hbm_bytes_per_launch": 5.0}))              # pre-round-6 file: no stamp
    assert bench._pmc_traffic("c3det_pmc", 52)[2].startswith("recorded at ABI unknown")
    assert bench._pmc_traffic("nothing_pmc", 1) == (None, None, None)
