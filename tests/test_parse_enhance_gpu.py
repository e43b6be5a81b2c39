"""HIP BiSeNet / RRDB paths vs the reference's golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bisenet_ref as B, rrdb_ref as RR

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ATTR = {"hair_and_hat": [17, 14], "no_cloth": [-16], "hair_only": [17, -14], "neck": [12]}
MASK = {"hair": [17], "neck_or_hat": [12, 14], "eyes": [4, 5]}


def test_glue_kernels_vs_torch(device):
    from face_crop_plus_amd.bise import BiSeNet
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 128, 16, 16, generator=g)
    a = E.Act(x.permute(0, 2, 3, 1).contiguous().to(device))
    avg = BiSeNet._avgpool(a)
    ref = F.avg_pool2d(x, 16).flatten(1)
    assert (avg.cpu() - ref).abs().max() < 1e-6
    w = torch.randn(64, 128, generator=g) / 11
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    got = BiSeNet._fc(avg, (w.to(device), sc.to(device), sh.to(device)), 2)
    exp = torch.sigmoid((ref @ w.t()) * sc + sh)
    assert (got.cpu() - exp).abs().max() < 1e-6
    s = torch.rand(3, 128, generator=g)
    addv = torch.randn(3, 128, generator=g)
    addt = torch.randn(3, 128, 16, 16, generator=g)
    out = BiSeNet._scale_add(a, s.to(device), add_nc=addv.to(device),
                             add_t=E.Act(addt.permute(0, 2, 3, 1).contiguous().to(device)))
    exp = x * s[:, :, None, None] + addv[:, :, None, None] + addt
    assert torch.equal(out.nchw().cpu(), exp)


def test_bise_preprocess_vs_torch(device):
    from face_crop_plus_amd import _native as N, engine as E
    import ctypes as C
    g = torch.Generator().manual_seed(1)
    faces = torch.randint(0, 256, (2, 100, 72, 3), generator=g, dtype=torch.uint8)
    out = E.Act.empty(2, 512, 512, 4, device)
    mean, std = (C.c_float * 3)(*B.MEAN), (C.c_float * 3)(*B.STD)
    N.check(N.lib().fcp_bise_preprocess_u8(N.ptr(faces.to(device)), 2, 100, 72, out.ptr(), 512, 512, mean, std,
                                           N.stream_ptr()))
    ref = B.preprocess(faces.permute(0, 3, 1, 2).float())
    got = out.nchw().cpu()[:, :3]
    assert (got - ref).abs().max() < 2e-6


def test_bisenet_labels_groups_vs_reference_golden(device):
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.bise import BiSeNet
    d = np.load(os.path.join(G, "bisenet.npz"))
    m = BiSeNet(ATTR, MASK, 2).load(device, weights.generate_state_dict("bisenet"))
    faces = torch.from_numpy(d["faces"]).to(device)
    labels, counts = m.parse(faces)
    labels = labels.cpu().numpy()
    mism = labels != d["labels"]
    print("label mismatches:", int(mism.sum()), "of", mism.size,
          "max margin at mismatch:", float(d["top2_gap"][mism].max()) if mism.any() else 0.0)
    # bit-exact wherever the reference's own decision margin exceeds fp32 summation-order noise
    assert (d["top2_gap"][mism] < 1e-4).all() and mism.mean() < 2e-3
    assert np.array_equal(counts.cpu().numpy(), np.stack([np.bincount(l.ravel(), minlength=19) for l in labels]))
    ag, mg = m.predict(faces)
    assert sorted(ag) == d["attr_keys"].tolist() and sorted(mg) == d["mask_keys"].tolist()
    for k in ag:
        assert ag[k] == d[f"attr_{k}"].tolist()
    for k in mg:
        assert mg[k][0] == d[f"mask_{k}_idx"].tolist()
        assert mg[k][1].dtype == np.uint8 and mg[k][1].shape == d[f"mask_{k}"].shape
        assert (mg[k][1] != d[f"mask_{k}"]).mean() < 2e-3
    # reference float-NCHW signature gives the same answer
    ag2, mg2 = m.predict(torch.from_numpy(d["faces"]).permute(0, 3, 1, 2).float())
    assert ag2 == ag and sorted(mg2) == sorted(mg)


def test_bisenet_logits_vs_oracle(device):
    from face_crop_plus_amd import weights, engine as E, _native as N
    from face_crop_plus_amd.bise import BiSeNet
    import ctypes as C
    sd = weights.generate_state_dict("bisenet")
    m = BiSeNet(None, None, 4).load(device, sd)
    g = torch.Generator().manual_seed(2)
    faces = torch.randint(0, 256, (2, 256, 256, 3), generator=g, dtype=torch.uint8)
    x4 = E.Act.empty(2, 512, 512, 4, device)
    N.check(N.lib().fcp_bise_preprocess_u8(N.ptr(faces.to(device)), 2, 256, 256, x4.ptr(), 512, 512,
                                           (C.c_float * 3)(*B.MEAN), (C.c_float * 3)(*B.STD), N.stream_ptr()))
    lg = m.forward_logits8(x4).nchw().cpu()
    with torch.no_grad():
        ref = B.forward_logits8(B.preprocess(faces.permute(0, 3, 1, 2).float()), sd)
    err = (lg - ref).abs().max().item()
    print("logit err", err, "scale", ref.abs().max().item())
    assert err < 1e-4 * max(1.0, ref.abs().max().item())


def test_rrdb_vs_reference_golden(device):
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.rrdb import RRDBNet
    d = np.load(os.path.join(G, "rrdb.npz"))
    m = RRDBNet(0.02).load(device, weights.generate_state_dict("rrdb"))
    img = torch.from_numpy(d["image"]).to(device)
    from face_crop_plus_amd import engine as E
    y = m.forward(E.u8_to_nhwc4(img[:1], div=255.0)).nchw().cpu()[:, :3]
    err = np.abs(y.numpy() - d["x4_image0"]).max()
    print("x4 err", err)
    assert err < 5e-5
    out = m.predict(img.clone(), d["landmarks"], d["indices"].tolist())
    got = out.permute(0, 3, 1, 2).float().cpu().numpy()
    assert np.array_equal(got[1:], d["pred"][1:])                 # gated off: untouched
    diff = np.abs(got[0] - d["pred"][0])
    assert diff.max() <= 1.0 and (diff > 0).mean() < 2e-3          # only rounding-boundary flips
    out_all = m.predict(img.clone(), None, None).permute(0, 3, 1, 2).float().cpu().numpy()
    diff = np.abs(out_all - d["pred_all"])
    assert diff.max() <= 1.0 and (diff > 0).mean() < 2e-3
    # reference float-NCHW signature
    xf = torch.from_numpy(d["image"]).permute(0, 3, 1, 2).float()
    outf = m.predict(xf, d["landmarks"], d["indices"].tolist())
    assert outf.dtype == torch.float32 and np.array_equal(outf.numpy(), got)


def test_bicubic_tail_exact_on_smooth_input(device):
    """clamp / *255 / round-half-even tail against torch on a hand-made x4 image."""
    from face_crop_plus_amd import _native as N
    g = torch.Generator().manual_seed(3)
    x4 = torch.rand(1, 3, 64, 48, generator=g) * 1.4 - 0.2
    ref = F.interpolate(x4, None, 0.25, "bicubic").clamp(0, 1).mul(255).round()[0].permute(1, 2, 0)
    src = torch.zeros(1, 64, 48, 4)
    src[..., :3] = x4.permute(0, 2, 3, 1)
    out = torch.empty((16, 12, 3), dtype=torch.uint8, device=device)
    N.check(N.lib().fcp_bicubic_down4_u8(N.ptr(src.to(device)), 16, 12, 4, N.ptr(out), N.stream_ptr()))
    assert np.abs(out.cpu().numpy().astype(int) - ref.numpy().astype(int)).max() <= 1
    assert (out.cpu().numpy() != ref.numpy().astype(np.uint8)).mean() < 0.01
