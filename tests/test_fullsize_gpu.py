"""BASELINE.json full-size configurations, checked through size-independent properties
(the oracle would need minutes per image at these sizes):

* batch independence — every image's result inside the big batch equals the result of running
  that image alone (the path has no cross-image coupling; K order is fixed, so this is bit-exact);
* permutation equivariance and run-to-run determinism;
* power-of-two linearity of the conv engine;
* crops of the full pipeline are byte-identical to warping the same image with the same landmarks alone.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det(device):
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.retinaface import RetinaFace
    return RetinaFace("largest", 0.6).load(device, weights.generate_state_dict("retinaface"))


def _detect(det, imgs):
    res = det.detect(imgs, max_faces=imgs.shape[0])
    torch.cuda.synchronize()
    nf = int(res["face_offset"][-1].item())
    return res["landmarks"][:nf].cpu().numpy(), res["img_idx"][:nf].cpu().numpy(), res


@pytest.mark.parametrize("batch,size", [(64, 640), (32, 1024)])
def test_configs_batch_independence_and_determinism(batch, size, det, device):
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randint(0, 256, (batch, size, size, 3), generator=g, dtype=torch.uint8).to(device)
    lm, idx, res = _detect(det, imgs)
    assert len(idx) == batch and idx.tolist() == list(range(batch))        # one "largest" face per image
    assert np.isfinite(lm).all() and lm.min() > -size and lm.max() < 2 * size
    lm2, idx2, _ = _detect(det, imgs)
    assert np.array_equal(lm, lm2) and np.array_equal(idx, idx2)            # deterministic
    for i in (0, batch // 2, batch - 1):                                     # image alone == image in the batch
        lmi, idxi, _ = _detect(det, imgs[i:i + 1].contiguous())
        assert np.array_equal(lmi[0], lm[i]), f"image {i} differs when run alone"
    perm = torch.randperm(batch, generator=g)
    lmp, idxp, _ = _detect(det, imgs[perm.to(device)].contiguous())
    assert np.array_equal(lmp, lm[perm.numpy()])                            # permutation equivariance
    # candidate bookkeeping invariants at full size
    cc = res["cand_count"].cpu().numpy()
    kc = res["keep_count"].cpu().numpy()
    assert (kc <= cc).all() and (kc >= 1).all() and (res["sel_count"].cpu().numpy() == 1).all()
    P = res["cand_score"].shape[1]
    assert P == sum(2 * (-(-size // s)) ** 2 for s in (8, 16, 32))
    for i in (0, batch - 1):
        sc = res["cand_score"][i, :cc[i]].cpu().numpy()
        pr = res["cand_prior"][i, :cc[i]].cpu().numpy()
        assert (sc > 0.6).all() and (np.diff(pr) > 0).all()                  # strict threshold, ascending prior order
        kp = res["keep_pos"][i, :kc[i]].cpu().numpy()
        assert (np.diff(sc[kp]) <= 0).all()                                  # kept boxes come in score order


@pytest.mark.parametrize("batch,size,sample", [(64, 640, (0, 21, 42, 63)), (32, 1024, (0, 31))])
def test_c2_c3_sample_vs_oracle(batch, size, sample, det, device):
    """The benchmark workloads themselves (BASELINE configs[1] / the north-star geometry, seed-1234 batch as in
    bench.py) against the oracle, on a sample of their images: the GPU results of the FULL batch at those images must
    give the oracle's indices, landmarks within 1e-3 px (north_star), and crops byte-equal to the oracle's estimate +
    warp of the GPU's landmarks (retinaface.py:449-470, cropper.py:514-547)."""
    from face_crop_plus_amd import align, weights
    from face_crop_plus_amd.cropper import landmarks_target
    from oracle import retinaface_ref as R, align_ref as A
    g = torch.Generator(device="cpu").manual_seed(1234)
    imgs_h = torch.randint(0, 256, (batch, size, size, 3), generator=g, dtype=torch.uint8)
    imgs = imgs_h.to(device)
    lm, idx, res = _detect(det, imgs)
    tgt = landmarks_target((256, 256), 0.65)
    crops, ok, _ = align.crop_align(imgs, res["img_idx"], res["landmarks"], tgt, (256, 256), 0)
    crops, ok = crops.cpu().numpy(), ok.cpu().numpy()
    sd = weights.generate_state_dict("retinaface")
    sub = imgs_h[list(sample)]
    lm_ref, idx_ref = R.predict(sub.permute(0, 3, 1, 2).float(), sd, "largest", 0.6)
    assert idx_ref == list(range(len(sample)))                              # one face per sampled image
    rows = [int(np.nonzero(idx == i)[0][0]) for i in sample]
    assert all((idx == i).sum() == 1 for i in sample)
    err = np.abs(lm[rows] - lm_ref).max()
    assert err < 1e-3, f"landmarks {err} px from the oracle"
    ref_crops = A.crop_align(sub.numpy(), None, idx_ref, lm[rows], tgt, (256, 256), "constant")
    assert ok[rows].all() and np.array_equal(crops[rows], ref_crops)


def test_full_size_crops_match_single_image_warp(det, device):
    from face_crop_plus_amd import align
    from face_crop_plus_amd.cropper import landmarks_target
    g = torch.Generator().manual_seed(7)
    imgs = torch.randint(0, 256, (32, 1024, 1024, 3), generator=g, dtype=torch.uint8).to(device)
    lm, idx, res = _detect(det, imgs)
    tgt = landmarks_target((256, 256), 0.65)
    crops, ok, mat = align.crop_align(imgs, res["img_idx"], res["landmarks"], tgt, (256, 256), 0)
    assert crops.shape == (32, 256, 256, 3) and bool(ok.all())
    for i in (0, 17, 31):
        single, ok1, _ = align.crop_align(imgs[i:i + 1].contiguous(), torch.zeros(1, dtype=torch.int32),
                                          res["landmarks"][i:i + 1], tgt, (256, 256), 0)
        assert torch.equal(single[0], crops[i])
    # a similarity transform has a = d, b = -c
    m = mat.cpu().numpy().reshape(-1, 2, 3)
    assert np.allclose(m[:, 0, 0], m[:, 1, 1]) and np.allclose(m[:, 0, 1], -m[:, 1, 0])


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_conv_power_of_two_linearity_full_size(precision, device):
    """conv(4x) == 4 conv(x): exact on the fp32 path (a power-of-two scale commutes with every rounding
    step); on the fp16x3 path only up to the lo parts' binary16 underflow, i.e. ~1e-6 of the output scale."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 80, 80, 256, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) / 48
    with E.default_precision(precision):
        pc = E.pack_conv(w, None, None, 1, 1, device)
        a = E.conv(pc, E.Act(x.to(device)))
        b = E.conv(pc, E.Act((x * 4).to(device)))
    if precision == "f32":
        assert torch.equal(a.buf * 4, b.buf)
    else:
        assert (a.buf * 4 - b.buf).abs().max().item() <= 4e-6 * b.buf.abs().max().item()


def test_c5_4k_frames_strategy_all(tmp_path, device):
    """SURVEY C5 at full geometry: 3840x2160 frames -> GPU batch builder (INTER_AREA to 1024x576 + 224 px pads)
    -> RetinaFace with strategy "all" (43 008 priors, ~100 faces) -> un-pad -> warp.  Checked end to end against
    the oracle chain (batch_ref -> retinaface_ref -> align_ref) on the same frames."""
    from face_crop_plus_amd import Cropper, weights
    from oracle import retinaface_ref as R, align_ref as A, batch_ref as B
    rng = np.random.default_rng(21)
    base = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    frames = [np.kron(base, np.ones((8, 8, 1), np.uint8)), rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)]
    sd = weights.generate_state_dict("retinaface")
    c = Cropper(output_size=128, resize_size=1024, strategy="all", det_threshold=0.55, device="cuda:0",
                weights={"retinaface": sd})
    from face_crop_plus_amd.batch import build_batch
    dev_batch, _, pads = build_batch(frames, c.resize_size, "constant", c.device)
    batch, _, epads = B.as_batch(frames, 1024)
    assert pads.tolist() == epads.tolist() == [[224, 224, 0, 0]] * 2
    assert np.array_equal(dev_batch.cpu().numpy(), batch)
    x_ref = torch.from_numpy(batch).permute(0, 3, 1, 2).float()
    lm_ref, idx_ref, ex = R.predict(x_ref, sd, "all", 0.55, return_all=True)
    lm, idx = c.det_model.predict(dev_batch)
    assert list(idx) == list(idx_ref) and len(idx) > 50                     # same faces per image, same order
    # Inside the zero padding the input is translation invariant: whole rows of priors tie (or differ by one ulp
    # of position-dependent summation order), so WHICH of them survives NMS is arithmetic noise on either side.
    # Faces on image content must agree to the usual tolerance; faces in the bands only in count and row.
    on_image = (lm_ref[:, :, 1].max(1) >= 224) & (lm_ref[:, :, 1].min(1) < 800)
    assert on_image.sum() >= 30
    err = np.abs(lm - lm_ref)
    # north_star's 1e-3 px against a float64 evaluation of the same faces (oracle/retinaface_ref.landmarks_fp64): the GPU may be
    # as far from the exact landmarks as the float32 oracle itself is, plus 1e-3, never more — and within 1e-3 of the oracle
    l64 = R.landmarks_fp64(x_ref, sd, ex)
    e_gpu, e_ora = float(np.abs(lm - l64)[on_image].max()), float(np.abs(lm_ref - l64)[on_image].max())
    print(f"C5 on-image faces: |gpu - fp64| {e_gpu:.3g} px, |oracle - fp64| {e_ora:.3g} px, |gpu - oracle| {float(err[on_image].max()):.3g} px; "
          f"band faces, rows: {float(np.abs(lm[..., 1] - lm_ref[..., 1]).max()):.3g} px")
    assert e_gpu <= e_ora + 1e-3 and err[on_image].max() < 1e-3, (e_gpu, e_ora, float(err[on_image].max()))
    assert np.abs(lm[..., 1] - lm_ref[..., 1]).max() < 1e-3                 # band faces: same rows
    # crops: byte-equal to the oracle's estimate + warp OF THE GPU'S OWN LANDMARKS (cropper.py:514-547) — the landmark
    # noise above (1e-4 px) flips isolated fixed-point roundings between the two landmark sets, so comparing against the
    # warp of the oracle's landmarks would need a byte budget; this comparison needs none
    un = lm - epads[idx_ref][:, None, [2, 0]].astype(np.float32)
    ref_crops = A.crop_align(batch, epads, idx_ref, un, A.landmarks_target((128, 128), 0.65), (128, 128), "constant")
    got = c.crop_align(batch, pads, list(idx), lm - pads[idx][:, None, [2, 0]].astype(np.float32))
    assert got.shape == ref_crops.shape and np.array_equal(got, ref_crops)


def _photo_like(h, w, seed):
    """Smooth structure + noise (an i.i.d. noise image exercises only the clamp of the RRDB tail)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 90 * np.sin(xx / (17 + 9 * c) + c) * np.cos(yy / (23 - 5 * c)) for c in range(3)], -1)
    img += rng.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_rrdb_c3_geometry_1024(device):
    """BASELINE configs[2] geometry for the enhancer: one real 1024x1024 image through RRDBNet.  At this size the
    x4-resolution tail (upconv2 / HRconv / conv_last, 64 channels at 4096x4096 = 4 GiB per tensor) runs in bands of
    256 output rows on the fp16x3 kernels (``RRDBNet._tail``; bit-identical to one band over the whole image,
    ``test_parse_enhance_gpu.py::test_rrdb_banded_tail_same_bits``), so no tensor of the pass reaches the 4 GiB limit of
    buffer addressing.  The oracle needs ~90 s per image here, so the check is by properties: (i) the
    fp16x3 trunk agrees with the all-fp32 path within one rounding flip on < 0.2 % of the bytes, (ii) two runs are
    bit-identical, (iii) a 256x256 corner crop run alone agrees with the big run away from the crop's border
    (the network is translation equivariant; receptive field < 120 px)."""
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.rrdb import RRDBNet
    sd = weights.generate_state_dict("rrdb")
    img = torch.from_numpy(_photo_like(1024, 1024, 5))[None].to(device)
    m16 = RRDBNet(1.0).load(device, sd, "f16x3")
    a = m16.predict(img.clone(), None, None)
    b = m16.predict(img.clone(), None, None)
    torch.cuda.synchronize()
    assert torch.equal(a, b), "RRDB at 1024^2 is not deterministic"
    assert not torch.equal(a, img), "enhancement left the image untouched"
    m32 = RRDBNet(1.0).load(device, sd, "f32")
    c = m32.predict(img.clone(), None, None)
    d = (a.int() - c.int()).abs()
    print("f16x3 vs f32 at 1024^2: max", int(d.max()), "differing bytes", float((d > 0).float().mean()))
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-3
    del m32, c
    crop = img[:, :256, :256].contiguous()
    e = m16.predict(crop.clone(), None, None)
    inner = (a[:, :128, :128].int() - e[:, :128, :128].int()).abs()
    assert int(inner.max()) <= 1 and float((inner > 0).float().mean()) < 2e-3


def test_rrdb_256_vs_oracle(device):
    """Largest size the torch-CPU oracle finishes in well under a minute on the GPU box's host cores."""
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.rrdb import RRDBNet
    from oracle import rrdb_ref as RR
    sd = weights.generate_state_dict("rrdb")
    img = torch.from_numpy(_photo_like(256, 256, 6))[None]
    ref = RR.predict(img.permute(0, 3, 1, 2).float(), sd, None, None).permute(0, 2, 3, 1).numpy()
    for prec in ("f16x3", "f32"):
        got = RRDBNet(1.0).load(device, sd, prec).predict(img.to(device), None, None).cpu().numpy()
        diff = np.abs(got.astype(int) - ref.astype(int))
        print(prec, "vs oracle at 256^2: max", diff.max(), "differing bytes", (diff > 0).mean())
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-3


def test_c3_full_pipeline_with_enhance(tmp_path, device):
    """BASELINE configs[2] end to end in one Cropper: detect -> RRDB gate + enhance -> align -> BiSeNet parse on
    1024x1024 inputs (batch of 2 to keep the test short: the enhancer costs ~0.2 s per image).  The gate decision,
    the enhanced batch and the final crops are checked against the same stages run one by one."""
    from PIL import Image
    from face_crop_plus_amd import Cropper, weights
    from face_crop_plus_amd.batch import build_batch
    src = tmp_path / "in"
    src.mkdir()
    rng = np.random.default_rng(30)          # i.i.d. noise: what the seeded random-init detector fires on
    imgs = [rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8) for _ in range(2)]
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(src / f"im{i}.png")
    w = {k: weights.generate_state_dict(k) for k in ("retinaface", "rrdb", "bisenet")}
    c = Cropper(output_size=256, resize_size=1024, strategy="largest", det_threshold=0.6, enh_threshold=1.0,
                attr_groups={"any": [1]}, mask_groups={"skin": [1]}, batch_size=2, device="cuda:0", weights=w)
    c.par_model.attr_threshold = -1            # every face lands in the group whatever the random-weight labels are
    c.par_model.mask_threshold = -1
    c.process_dir(str(src), str(tmp_path / "out"), desc=None)
    # the same stages one by one
    batch, _, pads = build_batch(imgs, c.resize_size, "constant", c.device)
    lm, idx = c.det_model.predict(batch)
    assert sorted(idx) == [0, 1]
    todo = c.enh_model.gate(2, 1024, 1024, lm, idx)
    assert todo == [0, 1]                       # threshold 1.0: every image with a face is enhanced
    enhanced = c.enh_model.predict(batch.clone(), lm, idx)
    assert not torch.equal(enhanced, batch)
    crops = c.crop_align(enhanced.cpu().numpy(), pads, list(idx), lm - pads[idx][:, None, [2, 0]].astype(np.float32))
    out_dir = tmp_path / "out" / "any" / "skin"
    files = sorted(p.name for p in out_dir.iterdir())
    assert files == ["im0.png", "im1.png"], files
    for k, i in enumerate(idx):
        got = np.asarray(Image.open(out_dir / f"im{i}.png").convert("RGB"))
        assert np.array_equal(got, crops[k]), f"crop of image {i} differs from the stage-by-stage result"
    masks = sorted(p.name for p in (tmp_path / "out" / "any" / "skin_mask").iterdir())
    assert masks == files


def test_c5_rrdb_gated_leg(tmp_path, device):
    """BASELINE configs[4]'s enhancement leg on letter-boxed 4K frames, strategy "all": the gate decision of the pipeline
    equals the oracle's gate (rrdb.py:124-140) on the same landmarks — with a threshold placed BETWEEN the two frames'
    face factors, so that exactly one frame is enhanced —, the batch the pipeline enhanced equals ``enh_model`` run stage
    by stage, and the crops the pipeline hands to the writer are the oracle's estimate + warp of the enhanced bytes."""
    from face_crop_plus_amd import Cropper, weights
    from face_crop_plus_amd.batch import build_batch
    from oracle import align_ref as A, batch_ref as B, rrdb_ref as RR
    rng = np.random.default_rng(22)
    base = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    frames = [np.kron(base, np.ones((8, 8, 1), np.uint8)), rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)]
    w = {k: weights.generate_state_dict(k) for k in ("retinaface", "rrdb")}
    c = Cropper(output_size=128, resize_size=1024, strategy="all", det_threshold=0.55, enh_threshold=0.001, device="cuda:0",
                weights=w)
    dev_batch, _, pads = build_batch(frames, c.resize_size, "constant", c.device)
    _, _, epads = B.as_batch(frames, 1024)
    assert pads.tolist() == epads.tolist() == [[224, 224, 0, 0]] * 2
    for vis in np.linspace(0.55, 0.999, 90):          # K calibration as in bench.py's Pipeline4K, in finer steps
        c.det_model.vis_threshold = float(vis)
        lm, idx = c.det_model.predict(dev_batch)
        if len(idx) <= 40:
            break
    assert set(idx) == {0, 1} and 2 <= len(idx) <= 40, (vis, len(idx))
    un = lm - pads[idx][:, None, [2, 0]].astype(np.float32)
    idx = list(idx)
    fac = [float(((un[np.array(idx) == i][:, 4, 0] - un[np.array(idx) == i][:, 0, 0]) *
                  (un[np.array(idx) == i][:, 4, 1] - un[np.array(idx) == i][:, 0, 1]) / (1024 * 1024)).mean()) for i in (0, 1)]
    assert fac[0] != fac[1]
    thr = (fac[0] + fac[1]) / 2
    c.enh_model.min_face_factor = thr
    expect = [i for i, on in enumerate(RR.gate(un, idx, 2, 1024, 1024, thr)) if on]      # the oracle returns one flag per image
    assert c.enh_model.gate(2, 1024, 1024, un, idx) == expect and len(expect) == 1, (fac, expect)
    # the pipeline itself (decoded frames in, writer intercepted)
    seen = {}
    enh_predict = c.enh_model.predict

    def recording_predict(images, landmarks, indices):
        out = enh_predict(images, landmarks, indices)
        seen["enhanced"], seen["landmarks"], seen["indices"] = out.clone(), np.array(landmarks), list(indices)
        return out
    c.enh_model.predict = recording_predict
    c.save_groups = lambda faces, names, out_dir, *groups: seen.update(faces=np.array(faces), names=list(names))
    c._process_images(frames, np.array(["f0.png", "f1.png"]), str(tmp_path / "out"))
    assert seen["indices"] == idx and np.array_equal(seen["landmarks"], un)
    # stage by stage
    staged = dev_batch.clone()
    c.enh_model.enhance_u8(staged, expect)
    torch.cuda.synchronize()
    assert torch.equal(seen["enhanced"], staged), "the pipeline's enhanced batch differs from the stage-by-stage one"
    other = 1 - expect[0]
    assert torch.equal(staged[other], dev_batch[other]) and not torch.equal(staged[expect[0]], dev_batch[expect[0]])
    ref = A.crop_align(staged.cpu().numpy(), epads, idx, un, A.landmarks_target((128, 128), 0.65), (128, 128), "constant")
    assert seen["faces"].shape == ref.shape and np.array_equal(seen["faces"], ref), "crops != oracle warp of the enhanced bytes"
    assert seen["names"] == [f"f{i}.png" for i in idx]
