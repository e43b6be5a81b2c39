"""CPU: self-consistency of the batch-builder oracle (oracle/batch_ref.py).  OpenCV is absent, so the
restated cv2.resize / copyMakeBorder arithmetic is *parity unpinned*; these properties pin everything that
does not need the library (geometry against the reference docstring, exact cases, <= 1 LSB from float64
evaluations of the documented filters, documented border patterns)."""
import numpy as np
import pytest

from oracle import batch_ref as B


def test_geometry_docstring_example_and_paddings():
    # utils.py:288-291: 1280x720 into (512, 256) -> resized to 455x256, width padded on both sides
    ww, hh, pad, unscale, interp = B.geometry(720, 1280, (512, 256))
    assert (ww, hh) == (455, 256) and pad == [0, 0, 28, 29] and interp == "area"
    assert unscale == 256 / 720
    assert B.geometry(50, 80, (64, 64))[:3] == (64, 40, [12, 12, 0, 0])
    assert B.geometry(90, 40, (64, 64))[:3] == (28, 64, [0, 0, 18, 18])
    assert B.geometry(50, 80, (64, 64))[4] == "area" and B.geometry(50, 60, 64)[4] == "cubic"
    # 4K frame into 1024^2: 1024x576 + 224 px top / bottom
    assert B.geometry(2160, 3840, 1024)[:3] == (1024, 576, [224, 224, 0, 0])


def test_same_size_is_a_copy():
    img = np.random.default_rng(0).integers(0, 256, (17, 23, 3), dtype=np.uint8)
    for interp in ("area", "cubic"):
        assert np.array_equal(B.resize_u8(img, 23, 17, interp), img)


def test_area_integral_scales():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (24, 36, 3), dtype=np.uint8)
    got = B.resize_area_u8(img, 18, 12)                             # 2x2: (a+b+c+d+2)>>2
    cells = img.reshape(12, 2, 18, 2, 3).astype(np.int64).sum((1, 3))
    assert np.array_equal(got, (cells + 2) >> 2)
    got = B.resize_area_u8(img, 12, 8)                              # 3x3: cvRound(sum * (1.f/9))
    mean = img.reshape(8, 3, 12, 3, 3).astype(np.float64).mean((1, 3))
    assert np.abs(got.astype(np.float64) - mean).max() <= 0.5 + 1e-4
    got = B.resize_area_u8(img, 9, 12)                              # 4 (x) by 2 (y)
    mean = img.reshape(12, 2, 9, 4, 3).astype(np.float64).mean((1, 3))
    assert np.abs(got.astype(np.float64) - mean).max() <= 0.5 + 1e-4


def _exact_area(img, dw, dh):
    """float64 box filter: every destination pixel = mean of the source over its (scale_x x scale_y) cell."""
    sh, sw = img.shape[:2]

    def weights(ss, ds):
        scale = ss / ds
        W = np.zeros((ds, ss))
        for d in range(ds):
            a, b = d * scale, min((d + 1) * scale, ss)
            for s in range(int(np.floor(a)), int(np.ceil(b))):
                W[d, s] = max(0.0, min(b, s + 1) - max(a, s))
            W[d] /= W[d].sum()
        return W
    Wy, Wx = weights(sh, dh), weights(sw, dw)
    return np.einsum("ys,stc,xt->yxc", Wy, img.astype(np.float64), Wx)


@pytest.mark.parametrize("shape,dst", [((45, 80), (64, 36)), ((100, 37), (23, 64)), ((135, 240), (192, 108)),
                                       ((61, 61), (60, 60))])
def test_area_general_close_to_exact_box_filter(shape, dst):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    got = B.resize_area_u8(img, *dst)
    assert got.shape == (dst[1], dst[0], 3)
    exact = _exact_area(img, *dst)
    assert np.abs(got.astype(np.float64) - exact).max() <= 0.5 + 2e-3      # float32 tables vs float64
    for v in (0, 1, 128, 255):                                      # the float32 weights of a cell sum to 1
        const = np.full(shape + (3,), v, np.uint8)
        assert np.array_equal(B.resize_area_u8(const, *dst), np.full((dst[1], dst[0], 3), v, np.uint8))


def _exact_cubic(img, dw, dh):
    """float64 Keys cubic convolution (A=-0.75), half-pixel centres, replicated edges."""
    def kern(t):
        t = np.abs(t)
        A = -0.75
        return np.where(t <= 1, ((A + 2) * t - (A + 3)) * t * t + 1,
                        np.where(t < 2, ((A * t - 5 * A) * t + 8 * A) * t - 4 * A, 0.0))

    def weights(ss, ds):
        scale = ss / ds
        W = np.zeros((ds, ss))
        for d in range(ds):
            f = (d + 0.5) * scale - 0.5
            s0 = int(np.floor(f))
            for j in range(-1, 3):
                W[d, min(max(s0 + j, 0), ss - 1)] += kern(f - (s0 + j))
        return W
    Wy, Wx = weights(img.shape[0], dh), weights(img.shape[1], dw)
    return np.einsum("ys,stc,xt->yxc", Wy, img.astype(np.float64), Wx)


@pytest.mark.parametrize("shape,dst", [((20, 30), (64, 42)), ((50, 60), (64, 53)), ((40, 64), (64, 40)),
                                       ((33, 7), (13, 64))])
def test_cubic_close_to_float64_convolution(shape, dst):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    got = B.resize_cubic_u8(img, *dst)
    exact = np.clip(_exact_cubic(img, *dst), 0, 255)
    # 11-bit coefficients: each of the two passes is within ~255 * 4 * 2^-12 of the exact filter
    assert np.abs(got.astype(np.float64) - exact).max() <= 0.5 + 0.6
    assert np.abs(got.astype(np.float64) - exact).mean() < 0.3
    for v in (0, 37, 255):
        const = np.full(shape + (3,), v, np.uint8)
        assert np.array_equal(B.resize_cubic_u8(const, *dst), np.full((dst[1], dst[0], 3), v, np.uint8))


def test_cubic_coefficients_partition_unity_and_known_values():
    c = B._cubic_coeffs(np.array([0.0, 0.5, 0.25], np.float32))
    assert np.allclose(c[0], [0, 1, 0, 0], atol=1e-7)
    assert np.allclose(c[1], [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-7)   # the (-3,19,19,-3)/32 taps
    assert np.allclose(c.sum(1), 1, atol=1e-6)


def test_border_patterns():
    a = np.arange(8)
    pat = lambda mode: "".join("abcdefgh"[i] for i in B.border_index(np.arange(-6, 15), 8, mode))
    assert pat("replicate") == "aaaaaa" + "abcdefgh" + "hhhhhhh"
    assert pat("reflect") == "fedcba" + "abcdefgh" + "hgfedcb"
    assert pat("reflect_101") == "gfedcb" + "abcdefgh" + "gfedcba"
    assert pat("wrap") == "cdefgh" + "abcdefgh" + "abcdefg"
    img = np.random.default_rng(4).integers(0, 256, (5, 4, 3), dtype=np.uint8)
    out = B.copy_make_border(img, 2, 3, 1, 0, "constant")
    assert out.shape == (10, 5, 3) and np.array_equal(out[2:7, 1:], img) and out[:2].sum() == 0 and out[7:].sum() == 0
    out = B.copy_make_border(img, 7, 0, 0, 9, "reflect_101")        # pad wider than the image
    assert np.array_equal(out[7:, :4], img) and np.array_equal(out[6, :4], img[1])
    assert a.sum() == 28


def test_as_batch_shapes_and_unscales():
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in ((50, 80), (90, 40), (64, 64), (20, 30))]
    batch, unscales, pads = B.as_batch(imgs, (64, 64))
    assert batch.shape == (4, 64, 64, 3) and batch.dtype == np.uint8
    assert pads.tolist() == [[12, 12, 0, 0], [0, 0, 18, 18], [0, 0, 0, 0], [11, 11, 0, 0]]
    assert np.allclose(unscales, [0.8, 64 / 90, 1.0, 64 / 30])
    assert np.array_equal(batch[2], imgs[2])                        # same size: untouched
    assert batch[0, :12].sum() == 0 and batch[0, 52:].sum() == 0 and batch[1, :, :18].sum() == 0


def test_area_integral_ratios_match_scikit_image_block_means():
    """Third-party check of row f1 at integral ratios: scikit-image 0.18.3 `downscale_local_mean` (fixture from
    tests/golden/make_golden_skimage_area.py; OpenCV is not in the image).  INTER_AREA is the rounded block mean there:
    (s + 2) >> 2 for 2 x 2 (ties up), cvRound(sum * (1.f / area)) otherwise (ties to even, float32 product)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_area.npz"))
    for k in range(int(z["cases"])):
        img, f = z[f"img{k}"], int(z[f"factor{k}"])
        mean = z[f"mean_x64_{k}"].astype(np.float64) / 64
        got = B.resize_area_u8(img, img.shape[1] // f, img.shape[0] // f).astype(np.float64)
        assert got.shape == mean.shape
        assert np.abs(got - mean).max() <= 0.5 + 1 / 64, (f, np.abs(got - mean).max())      # a rounding of the same mean
        off_tie = np.abs(mean - np.floor(mean) - 0.5) > 1 / 32
        assert np.array_equal(got[off_tie], np.rint(mean)[off_tie])                              # identical away from exact ties


@pytest.mark.parametrize("shape,dst", [((40, 64), (160, 100)), ((33, 21), (64, 100)), ((64, 48), (160, 213)), ((7, 9), (30, 23))])
def test_cubic_upscale_matches_torch_bicubic(shape, dst):
    """Third-party check of row f1's INTER_CUBIC leg (the reference only up-scales with it, utils.py:316-320): PyTorch's
    `F.interpolate(mode="bicubic", align_corners=False)` is an independently written implementation of the same filter —
    Keys cubic with A = -0.75, half-pixel centres, source indices clamped at the border — evaluated in float32.  The oracle's
    OpenCV restatement (int16 coefficients with 11 fractional bits, two fixed-point passes, saturating round) must agree with
    it within the rounding of those coefficients; a different A (Pillow's -0.5), a different centre convention or a different
    border rule would miss by tens of grey levels.  OpenCV's own rounding stays unpinned (no cv2 in the image)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    dw, dh = dst
    got = B.resize_cubic_u8(img, dw, dh).astype(np.float64)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
    ref = F.interpolate(t, size=(dh, dw), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).clamp(0, 255).numpy().astype(np.float64)
    d = np.abs(got - ref)
    print(f"{shape} -> {(dh, dw)}: max |oracle - torch bicubic| = {d.max():.3f} grey levels, mean {d.mean():.3f}")
    assert d.max() <= 0.5 + 0.6 and d.mean() < 0.3
    # the same comparison with the filter Pillow / many others use (A = -0.5) shows what the check excludes
    from oracle import batch_ref
    if hasattr(batch_ref, "_cubic_coeffs"):
        assert abs(float(batch_ref._cubic_coeffs(np.array([0.5], np.float32))[0][0]) - (-0.09375)) < 1e-7     # A = -0.75 -> -3/32
