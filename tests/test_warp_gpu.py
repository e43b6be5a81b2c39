"""HIP estimate_transform / warpAffine vs the oracle's OpenCV restatement (bit-exact bytes)."""
import numpy as np
import pytest
import torch

from oracle import align_ref as A

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("skew", [False, True])
def test_estimate_transform_matches_oracle(skew, device):
    from face_crop_plus_amd import align
    rng = np.random.default_rng(0)
    tgt = A.landmarks_target((256, 256), 0.65)
    src = rng.uniform(0, 640, (64, 5, 2)).astype(np.float32)
    src[5] = 7.0                 # degenerate: all points identical
    src[9, 3, 0] = np.inf        # non-finite
    mat, ok = align.estimate_transform(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), skew)
    mat, ok = mat.cpu().numpy(), ok.cpu().numpy()
    for i in range(64):
        ref = A.estimate_transform(src[i], tgt, skew)
        if ref is None:
            assert ok[i] == 0
        else:
            assert ok[i] == 1
            np.testing.assert_allclose(mat[i].reshape(2, 3), ref, rtol=1e-12, atol=1e-12)
    assert ok[5] == 0 and ok[9] == 0


@pytest.mark.parametrize("border", ["constant", "replicate", "reflect", "wrap", "reflect_101"])
@pytest.mark.parametrize("out_size", [(256, 256), (200, 300), (37, 29)])
def test_warp_affine_bit_exact(border, out_size, device):
    from face_crop_plus_amd import align
    rng = np.random.default_rng(hash((border, out_size)) & 0xFFFF)
    n, h, w = 3, 96, 128
    imgs = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    pads = np.array([[0, 0, 0, 0], [8, 9, 0, 0], [0, 0, 13, 12]], np.int32)
    tgt = A.landmarks_target(out_size, 0.65)
    f = 12
    idx = rng.integers(0, n, f).astype(np.int32)
    lms = []
    for k in range(f):       # faces of assorted scale / rotation / position, some far outside the image
        th, s = rng.uniform(-1.2, 1.2), rng.uniform(0.15, 2.5)
        Rm = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]) * s
        lms.append(tgt @ Rm.T + rng.uniform(-60, 120, 2) + rng.normal(0, 1.5, (5, 2)))
    lms = np.stack(lms).astype(np.float32)
    crops, ok, mat = align.crop_align(torch.from_numpy(imgs).to(device), torch.from_numpy(idx), torch.from_numpy(lms),
                                      tgt, out_size, align.border_code(border), False, torch.from_numpy(pads))
    crops, ok, mat = crops.cpu().numpy(), ok.cpu().numpy(), mat.cpu().numpy()
    assert ok.all()
    for k in range(f):
        t, b, l, r = pads[idx[k]]
        src = imgs[idx[k]][t:h - b, l:w - r]
        ref = A.warp_affine(src, mat[k].reshape(2, 3), out_size, A.BORDER[border])
        assert crops[k].shape == ref.shape == (out_size[1], out_size[0], 3)
        assert np.array_equal(crops[k], ref), f"face {k}: {np.abs(crops[k].astype(int) - ref).max()}"


def test_identity_warp_reproduces_source(device):
    from face_crop_plus_amd import align
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (1, 64, 64, 3), dtype=np.uint8)
    mat = torch.tensor([[1.0, 0, 0, 0, 1.0, 0]], dtype=torch.float64, device=device)
    out = align.warp_affine(torch.from_numpy(img).to(device), torch.zeros(1, dtype=torch.int32, device=device), mat,
                            None, None, (64, 64), 0)
    assert np.array_equal(out.cpu().numpy()[0], img[0])


def test_degenerate_face_is_flagged(device):
    from face_crop_plus_amd import align
    img = torch.zeros((1, 32, 32, 3), dtype=torch.uint8, device=device)
    lm = torch.ones((2, 5, 2))
    lm[1] = torch.from_numpy(A.landmarks_target((16, 16), 0.65))
    crops, ok, _ = align.crop_align(img, torch.zeros(2, dtype=torch.int32), lm, A.landmarks_target((16, 16), 0.65),
                                    (16, 16))
    assert ok.cpu().tolist() == [0, 1]


def test_estimate_transform_matches_scikit_image_umeyama(device):
    """The kernel's closed form against scikit-image's Umeyama estimate (fixture: tests/golden/make_golden_skimage.py)."""
    import os
    from face_crop_plus_amd import align
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_similarity.npz"))
    mat, ok = align.estimate_transform(torch.from_numpy(z["est_src"]).to(device), torch.from_numpy(z["est_dst"]).to(device), False)
    mat, ok = mat.cpu().numpy().reshape(-1, 2, 3), ok.cpu().numpy()
    assert ok.all()
    assert np.abs(mat - z["est_mat"]).max() < 1e-9


def test_warp_kernel_geometry_matches_scikit_image(device):
    """The kernel on the scikit-image geometry fixture (tests/test_align_oracle.py::test_warp_geometry_matches_scikit_image):
    byte-equal to the oracle there, hence within one grey level of an independent bilinear resampler in the interior."""
    import importlib.util, os
    from face_crop_plus_amd import _native as N
    spec = importlib.util.spec_from_file_location("_align_oracle_tests", os.path.join(os.path.dirname(__file__), "test_align_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _skimage_warp_cases = mod._skimage_warp_cases
    for img, M, (ow, oh), ref, interior in _skimage_warp_cases():
        images = torch.from_numpy(img[None].copy()).to(device)
        mat = torch.from_numpy(M.reshape(1, 6).copy()).to(device)
        idx = torch.zeros(1, dtype=torch.int32, device=device)
        out = torch.empty((1, oh, ow, 3), dtype=torch.uint8, device=device)
        N.check(N.lib().fcp_warp_affine_u8(N.ptr(images), 1, img.shape[0], img.shape[1], N.ptr(idx), N.ptr(mat), None, None, 1, oh, ow, 0,
                                           N.ptr(out), N.stream_ptr()), "fcp_warp_affine_u8")
        got = out.cpu().numpy()[0]
        assert np.array_equal(got, A.warp_affine(img, M, (ow, oh), 0))
        assert np.abs(got.astype(np.float64) - ref)[interior].max() <= 1.0
