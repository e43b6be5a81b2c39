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
