"""HIP decode / NMS / strategy kernels vs the oracle and the reference's golden vectors.
Index outputs must be bit-exact; floats within expf ulp noise."""
import os

import numpy as np
import pytest
import torch

from oracle import retinaface_ref as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _heads_from_raw(logits, loc, ldm, h, w, device):
    """Pack (n,P,2)/(n,P,4)/(n,P,10) reference-order predictions into the three fused
    32-channel NHWC head maps the decode kernel consumes."""
    n = logits.shape[0]
    heads, start = [], 0
    for s in (8, 16, 32):
        hl, wl = -(-h // s), -(-w // s)
        cnt = 2 * hl * wl
        sl = slice(start, start + cnt)
        t = np.concatenate([logits[:, sl].reshape(n, hl, wl, 4), loc[:, sl].reshape(n, hl, wl, 8),
                            ldm[:, sl].reshape(n, hl, wl, 20)], -1).astype(np.float32)
        heads.append(torch.from_numpy(np.ascontiguousarray(t)).to(device))
        start += cnt
    return heads


def _decode(heads, n, h, w, thr, device):
    from face_crop_plus_amd import _native as N
    P = sum(2 * (-(-h // s)) * (-(-w // s)) for s in (8, 16, 32))
    f32, i32 = torch.float32, torch.int32
    cs = torch.empty((n, P), dtype=f32, device=device); cb = torch.empty((n, P, 4), dtype=f32, device=device)
    cl = torch.empty((n, P, 10), dtype=f32, device=device); cp = torch.empty((n, P), dtype=i32, device=device)
    cc = torch.empty((n,), dtype=i32, device=device)
    ds = torch.empty((n, P), dtype=f32, device=device); db = torch.empty((n, P, 4), dtype=f32, device=device)
    dl = torch.empty((n, P, 10), dtype=f32, device=device)
    N.check(N.lib().fcp_retina_decode(N.ptr(heads[0]), N.ptr(heads[1]), N.ptr(heads[2]), n, h, w, thr, 0.1, 0.2,
                                      N.ptr(cs), N.ptr(cb), N.ptr(cl), N.ptr(cp), N.ptr(cc), N.ptr(ds), N.ptr(db),
                                      N.ptr(dl), N.stream_ptr()))
    torch.cuda.synchronize()
    return cs, cb, cl, cp, cc, ds, db, dl


def test_decode_vs_reference_golden(device):
    d = np.load(os.path.join(G, "retina_postprocess.npz"))
    h, w = int(d["h"]), int(d["w"])
    n = d["logits"].shape[0]
    heads = _heads_from_raw(d["logits"], d["loc"], d["ldm"], h, w, device)
    cs, cb, cl, cp, cc, ds, db, dl = _decode(heads, n, h, w, 0.6, device)
    np.testing.assert_allclose(ds.cpu().numpy(), d["scores"], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(db.cpu().numpy(), d["boxes"], rtol=3e-6, atol=2e-4)
    np.testing.assert_allclose(dl.cpu().numpy(), d["landms"], rtol=1e-6, atol=1e-5)
    # compaction: ascending prior order, strict '>' — compare where the margin to the threshold is unambiguous
    sc = d["scores"]
    for i in range(n):
        exp_idx = np.nonzero(sc[i] > np.float32(0.6))[0]
        assert np.abs(sc[i] - 0.6).min() > 1e-5
        k = int(cc[i].item())
        assert k == len(exp_idx)
        assert np.array_equal(cp[i, :k].cpu().numpy(), exp_idx)
        np.testing.assert_allclose(cs[i, :k].cpu().numpy(), sc[i, exp_idx], rtol=3e-6)
        np.testing.assert_allclose(cl[i, :k].cpu().numpy(), d["landms"][i, exp_idx], rtol=1e-6, atol=1e-5)


def test_priors_analytic_match_reference(device):
    """Zero regressions => boxes are exactly the priors' corners: checks the in-kernel PriorBox."""
    h, w = 100, 75
    pri = np.load(os.path.join(G, "retina_priors.npz"))[f"priors_{h}x{w}"]
    P = pri.shape[0]
    z = np.zeros((1, P, 1), np.float32)
    heads = _heads_from_raw(np.concatenate([z, z + 5], -1), np.repeat(z, 4, -1), np.repeat(z, 10, -1), h, w, device)
    *_, ds, db, dl = _decode(heads, 1, h, w, 0.6, device)
    f = np.float32
    x1 = (pri[:, 0] - pri[:, 2] / f(2)); y1 = (pri[:, 1] - pri[:, 3] / f(2))
    exp = np.stack([x1 * f(w), y1 * f(h), (pri[:, 2] + x1) * f(w), (pri[:, 3] + y1) * f(h)], -1)
    assert np.array_equal(db.cpu().numpy()[0], exp)
    assert np.array_equal(dl.cpu().numpy()[0, :, 0], pri[:, 0] * f(w))
    assert np.array_equal(dl.cpu().numpy()[0, :, 1], pri[:, 1] * f(h))


@pytest.mark.parametrize("h,w,n", [(1600, 1200, 2), (2048, 2048, 1)])
def test_decode_frames_above_65536_priors(h, w, n, device):
    """Frames whose prior count needs more than one 64-round span of the decode kernel (78 800 / 172 032 priors):
    dense decode vs the oracle, compaction in ascending prior order, and the NMS stage on that capacity."""
    rng = np.random.default_rng(h + w)
    pri = R.prior_box(h, w)
    P = pri.shape[0]
    assert P > 65536
    logits = rng.normal(0, 2.0, (n, P, 2)).astype(np.float32)
    loc = rng.normal(0, 1.0, (n, P, 4)).astype(np.float32)
    ldm = rng.normal(0, 1.0, (n, P, 10)).astype(np.float32)
    heads = _heads_from_raw(logits, loc, ldm, h, w, device)
    thr = 0.9
    cs, cb, cl, cp, cc, ds, db, dl = _decode(heads, n, h, w, thr, device)
    eb, el = R.decode(None, loc, ldm, pri, h, w)
    es = torch.softmax(torch.from_numpy(logits), -1)[..., 1].numpy()
    np.testing.assert_allclose(ds.cpu().numpy(), es, rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(db.cpu().numpy(), eb, rtol=3e-6, atol=6e-4)
    np.testing.assert_allclose(dl.cpu().numpy(), el, rtol=1e-6, atol=1e-4)
    got_s = ds.cpu().numpy()
    for i in range(n):
        idx = np.nonzero(got_s[i] > np.float32(thr))[0]      # the kernel's own scores: no threshold ambiguity
        k = int(cc[i].item())
        assert k == len(idx) and k > 1000
        assert np.array_equal(cp[i, :k].cpu().numpy(), idx)
        assert np.array_equal(cs[i, :k].cpu().numpy(), got_s[i, idx])
        assert np.array_equal(cb[i, :k].cpu().numpy(), db.cpu().numpy()[i, idx])
        assert np.array_equal(cl[i, :k].cpu().numpy(), dl.cpu().numpy()[i, idx])
    from face_crop_plus_amd.retinaface import nms_select
    out = nms_select(cs, cb, cc, 0.4, "all")
    torch.cuda.synchronize()
    for i in range(n):
        k = int(cc[i].item())
        keep = R.nms_single(cb[i, :k].cpu().numpy(), cs[i, :k].cpu().numpy(), 0.4)
        kc = int(out["keep_count"][i].item())
        assert kc == len(keep)
        assert out["keep_pos"][i, :kc].cpu().tolist() == keep


def _run_nms(scores, boxes, thr, strategy, device, vis=0.6):
    """scores (n,P), boxes (n,P,4): threshold+compact on host (exactly like the mask gather), NMS on device."""
    from face_crop_plus_amd.retinaface import nms_select
    n, P = scores.shape
    cs = np.zeros((n, P), np.float32); cb = np.zeros((n, P, 4), np.float32); cc = np.zeros((n,), np.int32)
    idxs = []
    for i in range(n):
        idx = np.nonzero(scores[i] > np.float32(vis))[0]
        idxs.append(idx)
        cs[i, :len(idx)] = scores[i, idx]; cb[i, :len(idx)] = boxes[i, idx]; cc[i] = len(idx)
    out = nms_select(torch.from_numpy(cs).to(device), torch.from_numpy(cb).to(device),
                     torch.from_numpy(cc).to(device), thr, strategy)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}, idxs


@pytest.mark.parametrize("fixture", ["retina_postprocess.npz", "retina_nms.npz"])
def test_nms_and_strategy_bit_exact_vs_reference_golden(fixture, device):
    d = np.load(os.path.join(G, fixture))
    scores, boxes, landms = d["scores"], d["boxes"], d["landms"]
    n = scores.shape[0]
    out, idxs = _run_nms(scores, boxes, 0.4, "all", device)
    got_l, got_idx = [], []
    for i in range(n):
        k = out["keep_count"][i]
        pri = idxs[i][out["keep_pos"][i, :k]]
        got_l.append(landms[i, pri]); got_idx += [i] * k
    assert got_idx == d["filt_idx"].tolist()
    assert np.array_equal(np.concatenate(got_l), d["filt_landms"])
    for strat in ("best", "largest"):
        out, idxs = _run_nms(scores, boxes, 0.4, strat, device)
        sel_l, sel_i = [], []
        for i in range(n):
            if out["sel_count"][i]:
                assert out["sel_count"][i] == 1
                sel_l.append(landms[i, idxs[i][out["sel_pos"][i, 0]]]); sel_i.append(i)
        assert sel_i == d[f"{strat}_idx"].tolist()
        assert np.array_equal(np.stack(sel_l), d[f"{strat}_landms"])


@pytest.mark.parametrize("K,spread", [(1, 100), (63, 40), (64, 40), (65, 300), (700, 150), (5000, 500), (9000, 900),
                                      (20000, 2500)])
def test_nms_random_vs_oracle(K, spread, device):
    """Sizes straddling the wave tile (64), the LDS sort limit (8192) and the global-memory sort path."""
    rng = np.random.default_rng(K)
    cxy = rng.uniform(0, spread, (K, 2)).astype(np.float32)
    wh = rng.uniform(4, 60, (K, 2)).astype(np.float32)
    boxes = np.concatenate([cxy - wh / 2, cxy + wh / 2], -1)[None]
    scores = rng.uniform(0.61, 1.0, (1, K)).astype(np.float32)
    scores[0, rng.integers(0, K, K // 10)] = scores[0, 0]          # ties
    out, idxs = _run_nms(scores, boxes, 0.4, "largest", device)
    keep = R.nms_single(boxes[0], scores[0], 0.4)
    k = out["keep_count"][0]
    assert k == len(keep)
    assert out["keep_pos"][0, :k].tolist() == keep
    kb = boxes[0, keep]
    areas = (kb[:, 2] - kb[:, 0] + np.float32(1)) * (kb[:, 3] - kb[:, 1] + np.float32(1))
    assert out["sel_pos"][0, 0] == keep[int(np.argmax(areas))]


def test_nms_empty_image_and_bad_strategy(device):
    from face_crop_plus_amd.retinaface import nms_select
    cs = torch.zeros((2, 128), device=device); cb = torch.zeros((2, 128, 4), device=device)
    cc = torch.tensor([0, 0], dtype=torch.int32, device=device)
    out = nms_select(cs, cb, cc, 0.4, "all")
    assert out["keep_count"].tolist() == [0, 0] and out["sel_count"].tolist() == [0, 0]
    with pytest.raises(ValueError):
        nms_select(cs, cb, cc, 0.4, "biggest")


@pytest.mark.parametrize("n,max_faces", [(1, 1), (5, 3), (7, 40), (64, 64), (200, 300)])
def test_gather_faces_one_block_per_image(n, max_faces, device):
    """``fcp_retina_gather_faces`` (one workgroup per image): image-major order, exclusive-prefix offsets, un-padding
    (cropper.py:822), truncation at ``max_faces`` and a ZEROED tail — against a plain numpy restatement."""
    from face_crop_plus_amd import _native as N
    rng = np.random.default_rng(n * 131 + max_faces)
    cap = 50
    ldm = rng.normal(0, 100, (n, cap, 10)).astype(np.float32)
    cnt = rng.integers(0, 4, n).astype(np.int32)
    if n > 2:
        cnt[1] = 0
    pos = np.stack([rng.permutation(cap) for _ in range(n)]).astype(np.int32)
    pads = rng.integers(0, 9, (n, 4)).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(device)
    d_ldm, d_pos, d_cnt, d_pad = t(ldm), t(pos), t(cnt), t(pads)
    off = torch.full((n + 1,), -7, dtype=torch.int32, device=device)
    out_l = torch.full((max_faces, 5, 2), float("nan"), device=device)
    out_i = torch.full((max_faces,), -7, dtype=torch.int32, device=device)
    N.check(N.lib().fcp_retina_gather_faces(N.ptr(d_ldm), N.ptr(d_pos), N.ptr(d_cnt), n, cap, N.ptr(d_pad), max_faces,
                                            N.ptr(off), N.ptr(out_l), N.ptr(out_i), N.stream_ptr()))
    torch.cuda.synchronize()
    exp_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    exp_l = np.zeros((max_faces, 5, 2), np.float32)
    exp_i = np.zeros((max_faces,), np.int32)
    f = 0
    for i in range(n):
        for k in range(cnt[i]):
            if f < max_faces:
                exp_l[f] = ldm[i, pos[i, k]].reshape(5, 2) - pads[i, [2, 0]].astype(np.float32)
                exp_i[f] = i
            f += 1
    assert np.array_equal(off.cpu().numpy(), exp_off)
    assert np.array_equal(out_l.cpu().numpy(), exp_l) and np.array_equal(out_i.cpu().numpy(), exp_i)


def test_estimate_transform_counted(device):
    """``fcp_estimate_transform_counted``: rows beyond the device-side face count get ok = 0 / a zero matrix, live rows
    equal the plain entry point bit for bit, and the number of kept faces is ADDED to the device accumulator."""
    from face_crop_plus_amd import align
    from face_crop_plus_amd.cropper import landmarks_target
    g = torch.Generator().manual_seed(2)
    f = 150
    tgt = torch.from_numpy(landmarks_target((128, 128), 0.65)).to(device)
    lm = (tgt.cpu()[None] * 1.5 + 40 + torch.rand(f, 5, 2, generator=g) * 30).to(device)
    lm[3] = 7.0                                     # five identical points: degenerate, dropped (cropper.py:529-531)
    lm[77, 2, 1] = float("nan")
    mat0, ok0 = align.estimate_transform(lm, tgt)
    total = torch.full((), 5, dtype=torch.int64, device=device)
    for live in (0, 1, 64, 100, 150, 400):
        fc = torch.tensor([9, live], dtype=torch.int32, device=device)[1:]       # a view with a storage offset
        before = int(total.item())
        mat, ok = align.estimate_transform(lm, tgt, False, fc, total)
        torch.cuda.synchronize()
        k = min(live, f)
        assert torch.equal(mat[:k], mat0[:k]) and torch.equal(ok[:k], ok0[:k])
        assert int(ok[k:].sum()) == 0 and float(mat[k:].abs().sum()) == 0.0
        assert int(total.item()) - before == int(ok0[:k].sum())
    assert int(ok0.sum()) == f - 2
