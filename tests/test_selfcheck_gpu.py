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


def test_selfcheck_compare_bands(monkeypatch):
    """<= tol silent; (tol, hard] warns (raises under FCP_SELFCHECK=1); > hard or non-finite raises in every mode."""
    import warnings
    from face_crop_plus_amd import engine as E
    monkeypatch.delenv("FCP_SELFCHECK", raising=False)
    ref = torch.tensor([1.0, -2.0, 0.5])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert E.selfcheck_compare("t", ref + 1e-5, ref, 1e-4) < 1e-4
    with pytest.warns(UserWarning, match="disagree"):
        d = E.selfcheck_compare("t", ref + 1e-3, ref, 1e-4)
    assert 1e-4 < d <= E.SELFCHECK_HARD_TOL
    with pytest.raises(FloatingPointError, match="hard limit"):
        E.selfcheck_compare("t", ref + 0.1, ref, 1e-4)
    with pytest.raises(FloatingPointError):
        E.selfcheck_compare("t", ref * float("nan"), ref, 1e-4)
    monkeypatch.setenv("FCP_SELFCHECK", "1")
    with pytest.raises(FloatingPointError, match="disagree"):
        E.selfcheck_compare("t", ref + 1e-3, ref, 1e-4)


def _oracle_vs(det, sd, images_u8):
    from oracle import retinaface_ref as R
    lm_ref, idx_ref = R.predict(images_u8.permute(0, 3, 1, 2).float(), sd, "largest", 0.6)
    lm, idx = det.predict(images_u8.to(det.device))
    assert list(idx) == list(idx_ref) and len(idx) > 0
    return float(np.abs(lm - lm_ref).max())


def test_trained_like_checkpoint_rehearsal(device, tmp_path, monkeypatch):
    """What a real checkpoint would do to the load path, without the checkpoint (no network here): the generated weights
    re-parametrised to trained statistics (``weights.trained_like_retinaface``: BatchNorm running_var over six decades,
    heavy-tailed filters, residual streams two to three decades above O(1)) go through ``load`` FROM A FILE, i.e. with the
    automatic self-check.  In range (peak ~1e4 < 2^15): silent, fp16x3 kept, landmarks / indices equal the oracle's on
    the same weights.  Out of range (a stream past 2^15): a default-precision load warns and lands on the exact-fp32
    path by itself — oracle-equal results again — while an explicit precision="f16x3" and the strict mode raise."""
    import warnings
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.retinaface import RetinaFace
    monkeypatch.delenv("FCP_SELFCHECK", raising=False)
    sd = weights.generate_state_dict("retinaface")
    g = torch.Generator().manual_seed(44)
    imgs = torch.randint(0, 256, (3, 160, 192, 3), generator=g, dtype=torch.uint8)
    # ---- in range
    ok = weights.trained_like_retinaface(sd, 1, stream_gain=(10.0, 30.0, 500.0, 10.0))
    rv = torch.cat([v.flatten() for k, v in ok.items() if k.endswith("running_var") and k != "body.bn1.running_var"])
    assert float(rv.min()) < 2e-3 and float(rv.max()) > 5e2
    torch.save(ok, tmp_path / "ok.pth")
    with warnings.catch_warnings():
        warnings.simplefilter("error")                           # the guard stays silent
        det = RetinaFace("largest", 0.6).load(device, str(tmp_path / "ok.pth"))
    rep = det.selfcheck_report
    peak = max(v for _, v in rep["launch_absmax"])
    print("trained-like in range: peak |x|", peak, "head rel diff", rep["head_rel_diff"])
    assert det.precision == 1 and 3e3 < peak < rep["limit"] and max(rep["head_rel_diff"]) < 1e-4
    err = _oracle_vs(det, ok, imgs)
    print("landmark err vs oracle (fp16x3, trained-like)", err)
    assert err < 1e-3
    # ---- out of range: layer3's stream at 4000x
    big = weights.trained_like_retinaface(sd, 1, stream_gain=(10.0, 30.0, 4000.0, 10.0), uniform_gain=True)
    torch.save(big, tmp_path / "big.pth")
    with pytest.warns(RuntimeWarning, match=r"2\^15.*Falling back to precision='f32'"):
        det32 = RetinaFace("largest", 0.6).load(device, str(tmp_path / "big.pth"))
    assert det32.precision == 0 and det32.selfcheck_report["fallback"] == "f32" and "2^15" in det32.selfcheck_report["reason"]
    err = _oracle_vs(det32, big, imgs)
    print("landmark err vs oracle (fallback f32, out-of-range weights)", err)
    assert err < 1e-3
    with pytest.raises(FloatingPointError, match=r"2\^15"):      # asked for by name: no silent change of arithmetic
        RetinaFace("largest", 0.6).load(device, str(tmp_path / "big.pth"), "f16x3")
    monkeypatch.setenv("FCP_SELFCHECK", "1")
    with pytest.raises(FloatingPointError, match=r"2\^15"):      # strict mode: an error, not a fallback
        RetinaFace("largest", 0.6).load(device, str(tmp_path / "big.pth"))


def test_trained_like_rehearsal_bisenet_and_rrdb(device, tmp_path, monkeypatch):
    """The same rehearsal for the other two networks (``weights.trained_like_statistics``: BatchNorm variance over six decades where
    there is BatchNorm, heavy-tailed filters everywhere), through ``load`` from a file: the guard stays silent, BiSeNet's label maps
    differ from the oracle's on the same weights only inside fp32 summation-order noise, RRDB's enhanced bytes only at rounding
    boundaries."""
    import warnings
    import torch.nn.functional as F
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.bise import BiSeNet
    from face_crop_plus_amd.rrdb import RRDBNet
    from oracle import bisenet_ref as B, rrdb_ref as RR
    monkeypatch.delenv("FCP_SELFCHECK", raising=False)
    sb = weights.trained_like_statistics(weights.generate_state_dict("bisenet"), 2)
    rv = torch.cat([v.flatten() for k, v in sb.items() if k.endswith("running_var")])
    assert float(rv.min()) < 2e-3 and float(rv.max()) > 5e2
    torch.save(sb, tmp_path / "bise.pth")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        par = BiSeNet(None, None, 4).load(device, str(tmp_path / "bise.pth"))
    rep = par.selfcheck_report
    print("BiSeNet trained-like: peak |x|", max(v for _, v in rep["launch_absmax"]), "logit rel diff", rep["logit_rel_diff"])
    assert par.precision == 1 and rep["logit_rel_diff"] < 1e-4
    g = torch.Generator().manual_seed(12)
    faces = torch.randint(0, 256, (2, 160, 192, 3), generator=g, dtype=torch.uint8)
    labels, _ = par.parse(faces.to(device))
    with torch.no_grad():
        x = faces.permute(0, 3, 1, 2).float()
        logits = F.interpolate(B.forward(B.preprocess(x), sb), size=x.shape[2:], mode="nearest")
    top2 = logits.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]).numpy()
    mism = labels.cpu().numpy() != logits.argmax(1).numpy()
    scale = float(logits.abs().max())
    print("trained-like label mismatches:", int(mism.sum()), "of", mism.size, "largest oracle margin at a mismatch:",
          float(gap[mism].max()) if mism.any() else 0.0, "logit scale", scale)
    assert (gap[mism] < 1e-4 * max(1.0, scale)).all() and mism.mean() < 2e-3
    se = weights.trained_like_statistics(weights.generate_state_dict("rrdb"), 3)
    torch.save(se, tmp_path / "rrdb.pth")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        enh = RRDBNet(1.0).load(device, str(tmp_path / "rrdb.pth"))
    rep = enh.selfcheck_report
    print("RRDB heavy-tailed: peak |x|", max(v for _, v in rep["launch_absmax"]), "output rel diff", rep["output_rel_diff"])
    assert enh.precision == 1 and rep["output_rel_diff"] < 1e-4
    img = torch.randint(0, 256, (1, 96, 80, 3), generator=g, dtype=torch.uint8)
    ref = RR.predict(img.permute(0, 3, 1, 2).float(), se, None, None).permute(0, 2, 3, 1).numpy()
    got = enh.predict(img.to(device), None, None).cpu().numpy()
    diff = np.abs(got.astype(int) - ref.astype(int))
    print("RRDB heavy-tailed vs oracle: max", diff.max(), "differing bytes", float((diff > 0).mean()))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3
