"""Pin the oracle (oracle/retinaface_ref.py) against golden vectors produced by
the reference's own code (tests/golden/make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import retinaface_ref as R

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_priors_match_reference():
    d = load("retina_priors.npz")
    for (h, w) in [(64, 96), (100, 75)]:
        assert np.array_equal(R.prior_box(h, w), d[f"priors_{h}x{w}"])
    for (h, w) in [(640, 640), (1024, 1024), (576, 1024)]:
        p = R.prior_box(h, w)
        assert p.shape[0] == int(d[f"count_{h}x{w}"])
        assert hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() == str(d[f"digest_{h}x{w}"])
    assert R.prior_box(1024, 1024).shape[0] == 43008 and R.prior_box(640, 640).shape[0] == 16800


def test_decode_matches_reference():
    d = load("retina_postprocess.npz")
    h, w = int(d["h"]), int(d["w"])
    boxes, landms = R.decode(d["scores"], d["loc"], d["ldm"], R.prior_box(h, w), h, w)
    # exp() may differ in the last ulp between numpy and ATen; everything else is exact
    np.testing.assert_allclose(boxes, d["boxes"], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(landms, d["landms"], rtol=0, atol=0)


@pytest.mark.parametrize("fixture", ["retina_postprocess.npz", "retina_nms.npz"])
def test_filter_and_strategy_indices_bit_exact(fixture):
    d = load(fixture)
    fl, fb, sidx, _ = R.filter_preds(d["scores"], d["boxes"], d["landms"], 0.6, 0.4)
    assert sidx == d["filt_idx"].tolist()
    assert np.array_equal(fl, d["filt_landms"])
    assert np.array_equal(fb, d["filt_boxes"])
    for strat in ("all", "best", "largest"):
        lm, idx, _ = R.take_by_strategy(fl, fb, sidx, strat)
        assert idx == d[f"{strat}_idx"].tolist()
        assert np.array_equal(lm, d[f"{strat}_landms"])


def test_iou_exactly_at_threshold_survives():
    """retinaface.py:292 keeps `ovr <= 0.4`: the engineered pair in retina_nms.npz
    (image 1, candidates 0 and 1, IoU == 0.4 in float32) must both be kept."""
    d = load("retina_nms.npz")
    keep = R.nms_single(d["boxes"][1, :2], d["scores"][1, :2], 0.4)
    assert keep == [0, 1]


def test_bad_strategy_raises():
    with pytest.raises(ValueError):
        R.take_by_strategy(np.zeros((1, 10), np.float32), np.zeros((1, 4), np.float32), [0], "biggest")


def test_full_forward_and_predict_match_reference():
    from face_crop_plus_amd import weights
    d = load("retina_full.npz")
    sd = weights.generate_state_dict("retinaface")
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].numpy()).tobytes())
    assert h.hexdigest() == str(d["sd_digest"]), "weight generator drifted from the golden fixtures"
    x = torch.from_numpy(d["image"]).permute(0, 3, 1, 2).float()
    with torch.no_grad():
        prob, loc, ldm = R.forward(R.preprocess(x), sd)
    # same ATen kernels as the reference -> tight tolerance (not bit-exact across CPU ISAs)
    np.testing.assert_allclose(prob.numpy(), d["prob"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(loc.numpy(), d["loc"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ldm.numpy(), d["ldm"], rtol=1e-4, atol=1e-4)
    for strat, thr in (("all", 0.6), ("best", 0.6), ("largest", 0.6), ("all", 0.5)):
        lm, idx = R.predict(x, sd, strat, thr)
        assert idx == d[f"pred_{strat}_{thr}_indices"].tolist()
        np.testing.assert_allclose(lm, d[f"pred_{strat}_{thr}_landmarks"], rtol=0, atol=1e-3)


def test_resnet50_body_matches_transformers_resnet():
    """Third-party pin of the body topology (row a3): tests/golden/hf_resnet50.npz holds the stage 2-4 outputs of Hugging Face
    transformers' ResNetModel (an independent ResNet-50 v1.5, `downsample_in_bottleneck=False`) on the build's generated `body.*`
    weights (make_golden_hf_resnet.py).  The oracle's body must reproduce them — and the classic wrong topologies must NOT:
    the stride on conv1 instead of conv2 (ResNet v1), or a stem pool without padding."""
    import torch.nn.functional as F
    from face_crop_plus_amd import weights
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hf_resnet50.npz"))
    print("fixture from transformers", z["transformers_version"], "torch", z["torch_version"], str(z["config"]))
    sd = weights.generate_state_dict("retinaface")
    x = torch.from_numpy(z["x"])
    with torch.no_grad():
        feats = R.body(x, sd)
    for k, f in enumerate(feats, 1):
        ref = z[f"feat{k}"]
        assert tuple(f.shape) == ref.shape
        err = float(np.abs(f.numpy() - ref).max()) / float(np.abs(ref).max())
        assert err < 1e-5, (k, err)

    def body_variant(stride_on_conv1, pool_pad):
        t = F.relu(R._bn(R._conv(x, sd, "body.conv1", 2, 3), sd, "body.bn1"))
        t = F.max_pool2d(t, 3, 2, pool_pad)
        out = []
        for li, blocks in enumerate((3, 4, 6, 3), 1):
            for b in range(blocks):
                p, s = f"body.layer{li}.{b}", 2 if (b == 0 and li > 1) else 1
                s1, s2 = (s, 1) if stride_on_conv1 else (1, s)
                o = F.relu(R._bn(R._conv(t, sd, p + ".conv1", s1), sd, p + ".bn1"))
                o = F.relu(R._bn(R._conv(o, sd, p + ".conv2", s2, 1), sd, p + ".bn2"))
                o = R._bn(R._conv(o, sd, p + ".conv3"), sd, p + ".bn3")
                if (p + ".downsample.0.weight") in sd:
                    t = R._bn(R._conv(t, sd, p + ".downsample.0", s), sd, p + ".downsample.1")
                t = F.relu(o + t)
            if li >= 2:
                out.append(t)
        return out

    with torch.no_grad():
        same = body_variant(False, 1)
        assert all(torch.equal(a, b) for a, b in zip(same, feats))                      # the helper restates R.body
        v1 = body_variant(True, 1)                                                       # ResNet v1: stride on the first 1x1
        assert all(a.shape == b.shape for a, b in zip(v1, feats))                        # every shape survives ...
        assert float(np.abs(v1[0].numpy() - z["feat1"]).max()) / float(np.abs(z["feat1"]).max()) > 1e-2   # ... the pin does not
