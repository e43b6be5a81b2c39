"""Pin the BiSeNet / RRDB oracles against golden vectors produced by the reference's own code."""
import hashlib
import os

import numpy as np
import torch

from oracle import bisenet_ref as B, rrdb_ref as RR

G = os.path.join(os.path.dirname(__file__), "golden")
ATTR = {"hair_and_hat": [17, 14], "no_cloth": [-16], "hair_only": [17, -14], "neck": [12]}
MASK = {"hair": [17], "neck_or_hat": [12, 14], "eyes": [4, 5]}


def _digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].numpy()).tobytes())
    return h.hexdigest()


def test_bisenet_labels_and_groups_match_reference():
    from face_crop_plus_amd import weights
    d = np.load(os.path.join(G, "bisenet.npz"))
    sd = weights.generate_state_dict("bisenet")
    assert _digest(sd) == str(d["sd_digest"])
    x = torch.from_numpy(d["faces"]).permute(0, 3, 1, 2).float()
    labels = B.parse_labels(x, sd, 2)
    mism = labels != d["labels"]
    # same ATen kernels: identical except (possibly) where the decision margin is at float-noise level
    assert mism.mean() < 1e-3 and (d["top2_gap"][mism] < 1e-4).all()
    ag, mg = B.predict(x, sd, ATTR, MASK, 2)
    assert sorted(ag) == d["attr_keys"].tolist() and sorted(mg) == d["mask_keys"].tolist()
    for k in ag:
        assert ag[k] == d[f"attr_{k}"].tolist()
    for k in mg:
        assert mg[k][0] == d[f"mask_{k}_idx"].tolist()
        assert (mg[k][1] != d[f"mask_{k}"]).mean() < 1e-3


def test_group_rules_thresholds():
    lab = np.zeros((2, 8, 8), np.int64)
    lab[0, 0, :6] = 6            # 6 pixels of class 6  (> 5  -> present)
    lab[1, 0, :5] = 6            # 5 pixels             (<= 5 -> absent)
    lab[0, 1:3, :] = 4           # 16 px (> 10 -> mask kept)
    lab[1, 1, :] = 4
    lab[1, 2, :2] = 4            # 10 px (not > 10 -> mask dropped)
    ag = B.group_by_attributes(lab, {"glasses": [6], "no_glasses": [-6]})
    assert ag == {"glasses": [0], "no_glasses": [1]}
    mg = B.group_by_masks(lab, {"eye": [4]})
    assert mg["eye"][0] == [0] and mg["eye"][1].dtype == np.uint8 and mg["eye"][1].max() == 255
    assert mg["eye"][1].sum() == 16 * 255


def test_rrdb_predict_matches_reference():
    from face_crop_plus_amd import weights
    d = np.load(os.path.join(G, "rrdb.npz"))
    sd = weights.generate_state_dict("rrdb")
    assert _digest(sd) == str(d["sd_digest"])
    x = torch.from_numpy(d["image"]).permute(0, 3, 1, 2).float()
    with torch.no_grad():
        x4 = RR.forward(x[:1].div(255), sd)
    np.testing.assert_allclose(x4.numpy(), d["x4_image0"], rtol=1e-4, atol=1e-5)
    res = RR.predict(x, sd, d["landmarks"], d["indices"].tolist(), 0.02).numpy()
    assert np.array_equal(res[1:], d["pred"][1:]) and np.array_equal(res[1:], x.numpy()[1:])
    assert np.abs(res[0] - d["pred"][0]).max() <= 1.0          # at most a rounding flip
    assert (res[0] != d["pred"][0]).mean() < 1e-3
    assert RR.gate(d["landmarks"], d["indices"].tolist(), 3, 20, 24, 0.02) == [True, False, False]
    assert RR.gate(None, None, 3, 20, 24, 0.02) == [True, True, True]
