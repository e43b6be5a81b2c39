"""Range guard of the fp16x3 (split binary16) path: the reference computes in fp32 and accepts any checkpoint
(_layers.py:16-35); binary16 hi parts saturate at 65504, so weights that drive an activation past 2^15 must be
refused loudly instead of producing silently clipped landmarks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_absmax_kernel_matches_torch(device):
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(5)
    x = (torch.randn((3, 17, 19, 96), generator=g) * 40).to(device)
    x[1, 5, 7, 40] = -1234.5
    a = E.Act(x.contiguous())
    for view in (a, a.slice(32, 64)):
        for t in (view, None):
            if t is None:                                        # the same view in split32 format
                full = E.f32_to_split32(a)
                t = full if view is a else full.slice(32, 64)
            with E.RangeMonitor() as mon:
                mon.see("t", t)
            want = float(x[..., view.c0:view.c0 + view.c].abs().max().item())
            got = mon.report()[0][1]
            assert abs(got - want) <= 1e-3 * want                # split32 keeps ~22 bits
    y = x.clone()
    y[0, 0, 0, 3] = float("nan")
    with E.RangeMonitor() as mon:
        mon.see("nan", E.Act(y))
    assert mon.report()[0][1] == float("inf")                    # NaN counts as out of range


def test_selfcheck_passes_on_generated_weights_and_reports(device):
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.retinaface import RetinaFace
    sd = weights.generate_state_dict("retinaface")
    det = RetinaFace("largest", 0.6).load(device, sd)
    rep = det.selfcheck(sd)
    rows = rep["launch_absmax"]
    assert len(rows) >= 50 and all(0 < v < rep["limit"] for _, v in rows)
    assert max(rep["head_rel_diff"]) < 1e-4
    assert any(lbl.startswith("chain") for lbl, _ in rows) and any(lbl.startswith("stem") for lbl, _ in rows)


def test_selfcheck_trips_on_out_of_range_weights(device, monkeypatch):
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.retinaface import RetinaFace
    sd = weights.generate_state_dict("retinaface")
    for scale in (8.0, 3e4):
        bad = dict(sd)
        bad["body.layer2.1.bn3.weight"] = sd["body.layer2.1.bn3.weight"] * scale
        bad["body.layer2.1.bn3.bias"] = sd["body.layer2.1.bn3.bias"] * scale
        det = RetinaFace("largest", 0.6).load(device, bad)       # a state dict in memory: no automatic check
        if scale < 100:
            assert max(v for _, v in det.selfcheck(bad)["launch_absmax"]) < 32768.0     # large but representable: passes
            continue
        with pytest.raises(FloatingPointError, match=r"2\^15.*precision='f32'"):
            det.selfcheck(bad)
        monkeypatch.setenv("FCP_SELFCHECK", "1")                 # ... and at load time when asked for
        with pytest.raises(FloatingPointError, match="512->128|128->512"):
            RetinaFace("largest", 0.6).load(device, bad)
        monkeypatch.setenv("FCP_SELFCHECK", "0")
        RetinaFace("largest", 0.6).load(device, bad)
        # the exact-fp32 path takes the same weights without complaint
        det32 = RetinaFace("largest", 0.6).load(device, bad, "f32")
        assert "skipped" in det32.selfcheck(bad)


def test_selfcheck_runs_automatically_for_checkpoint_files(device, tmp_path, monkeypatch):
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.retinaface import RetinaFace
    monkeypatch.delenv("FCP_SELFCHECK", raising=False)
    sd = weights.generate_state_dict("retinaface")
    torch.save(sd, tmp_path / "retinaface_detector.pth")
    det = RetinaFace("largest", 0.6).load(device, str(tmp_path / "retinaface_detector.pth"))
    assert max(det.selfcheck_report["head_rel_diff"]) < 1e-4
    assert not hasattr(RetinaFace("largest", 0.6).load(device, sd), "selfcheck_report")


def test_selfcheck_bisenet_and_rrdb(device):
    """The same guard on the other two networks: generated weights pass (ranges far below 2^15, fp16x3 == exact fp32 to
    1e-4 of the output's largest value); a blown-up BatchNorm scale / conv weight trips it."""
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.bise import BiSeNet
    from face_crop_plus_amd.rrdb import RRDBNet
    sdb = weights.generate_state_dict("bisenet")
    par = BiSeNet({"glasses": [6]}, None, 8).load(device, sdb)
    rep = par.selfcheck(sdb)
    assert len(rep["launch_absmax"]) >= 25 and all(0 < v < rep["limit"] for _, v in rep["launch_absmax"])
    assert rep["logit_rel_diff"] < 1e-4
    bad = dict(sdb)
    bad["cp.resnet.layer3.0.bn2.weight"] = sdb["cp.resnet.layer3.0.bn2.weight"] * 1e5
    with pytest.raises(FloatingPointError, match=r"BiSeNet.*2\^15"):
        BiSeNet(None, {"eyes": [4, 5]}, 8).load(device, bad).selfcheck(bad)
    sde = weights.generate_state_dict("rrdb")
    enh = RRDBNet(0.001).load(device, sde)
    rep = enh.selfcheck(sde)
    assert len(rep["launch_absmax"]) >= 351 and all(v < rep["limit"] for _, v in rep["launch_absmax"])
    assert rep["output_rel_diff"] < 1e-4
    bad = dict(sde)
    bad["RRDB_trunk.3.RDB2.conv3.weight"] = sde["RRDB_trunk.3.RDB2.conv3.weight"] * 1e6
    with pytest.raises(FloatingPointError, match=r"RRDBNet.*2\^15"):
        RRDBNet(0.001).load(device, bad).selfcheck(bad)
