"""Self-consistency of the OpenCV restatement (oracle/align_ref.py).  OpenCV is
not available anywhere in this build ("parity unpinned", see the module header):
these properties pin the restated algorithm instead."""
import numpy as np
import pytest

from oracle import align_ref as A


def test_landmarks_target_defaults():
    t = A.landmarks_target((256, 256), 0.65)
    ref = np.float32([[97.31, 121.61], [158.39, 121.61], [128.04, 151.38], [102.95, 182.03], [153.53, 182.03]])
    assert t.dtype == np.float32
    np.testing.assert_allclose(t, ref, atol=0.006)


@pytest.mark.parametrize("skew", [False, True])
def test_transform_recovers_known_map(skew):
    rng = np.random.default_rng(0)
    src = rng.uniform(50, 400, (5, 2)).astype(np.float32)
    if skew:
        M = np.array([[0.8, 0.3, 12.0], [-0.2, 1.1, -7.0]])
    else:
        a, b = 0.7 * np.cos(0.4), 0.7 * np.sin(0.4)
        M = np.array([[a, -b, 31.0], [b, a, -5.5]])
    dst = (src.astype(np.float64) @ M[:, :2].T + M[:, 2]).astype(np.float32)
    got = A.estimate_transform(src, dst, skew)
    np.testing.assert_allclose(got, M, atol=2e-5)


def test_similarity_is_lsq_optimal():
    rng = np.random.default_rng(1)
    src = rng.uniform(0, 300, (5, 2)).astype(np.float32)
    dst = A.landmarks_target((256, 256), 0.65)
    M = A.estimate_transform(src, dst, False)
    assert abs(M[0, 0] - M[1, 1]) < 1e-15 and abs(M[0, 1] + M[1, 0]) < 1e-15

    def cost(p):
        a, b, tx, ty = p
        x, y = src[:, 0].astype(np.float64), src[:, 1].astype(np.float64)
        return ((a * x - b * y + tx - dst[:, 0]) ** 2 + (b * x + a * y + ty - dst[:, 1]) ** 2).sum()
    p0 = np.array([M[0, 0], M[1, 0], M[0, 2], M[1, 2]])
    c0 = cost(p0)
    for i in range(4):                      # normal equations: gradient vanishes
        e = np.zeros(4); e[i] = 1e-4
        assert cost(p0 + e) >= c0 - 1e-9 and cost(p0 - e) >= c0 - 1e-9


def test_degenerate_and_nonfinite_are_dropped():
    dst = A.landmarks_target((256, 256), 0.65)
    assert A.estimate_transform(np.ones((5, 2), np.float32), dst) is None
    bad = np.ones((5, 2), np.float32); bad[2, 1] = np.nan
    assert A.estimate_transform(bad, dst) is None


def _img(h=37, w=53, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_identity_and_integer_translation_are_bit_exact():
    img = _img()
    I = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(A.warp_affine(img, I, (53, 37), 0), img)
    T = np.array([[1.0, 0, 5], [0, 1.0, 3]])     # dst(x,y) = src(x-5, y-3)
    out = A.warp_affine(img, T, (53, 37), 0)
    assert np.array_equal(out[3:, 5:], img[:-3, :-5])
    assert not out[:3].any() and not out[:, :5].any()


def test_border_patterns_follow_opencv_documentation():
    # 1-D check through borderInterpolate: gfedcb|abcdefgh|gfedcba etc.
    n = 8
    p = np.arange(-6, n + 7)
    assert A.border_interpolate(p, n, 1).tolist() == [0] * 6 + list(range(8)) + [7] * 7            # aaaaaa|abcdefgh|hhhhhhh
    assert A.border_interpolate(p, n, 2).tolist() == [5, 4, 3, 2, 1, 0] + list(range(8)) + [7, 6, 5, 4, 3, 2, 1]  # fedcba|abcdefgh|hgfedcb
    assert A.border_interpolate(p, n, 4).tolist() == [6, 5, 4, 3, 2, 1] + list(range(8)) + [6, 5, 4, 3, 2, 1, 0]  # gfedcb|abcdefgh|gfedcba
    assert A.border_interpolate(p, n, 3).tolist() == [2, 3, 4, 5, 6, 7] + list(range(8)) + [0, 1, 2, 3, 4, 5, 6]  # cdefgh|abcdefgh|abcdefg
    assert A.border_interpolate(p, n, 0).tolist() == [-1] * 6 + list(range(8)) + [-1] * 7


@pytest.mark.parametrize("border", [0, 1, 2, 3, 4])
def test_translation_with_border_modes(border):
    img = _img(20, 24, 3)
    T = np.array([[1.0, 0, -4], [0, 1.0, 6]])    # dst(x,y) = src(x+4, y-6)
    out = A.warp_affine(img, T, (24, 20), border)
    ys = A.border_interpolate(np.arange(20) - 6, 20, border)
    xs = A.border_interpolate(np.arange(24) + 4, 24, border)
    exp = img[np.maximum(ys, 0)][:, np.maximum(xs, 0)].copy()
    exp[ys < 0] = 0
    exp[:, xs < 0] = 0
    assert np.array_equal(out, exp)


def test_within_one_lsb_of_float64_bilinear():
    img = _img(64, 80, 5)
    th = 0.3
    M = np.array([[1.3 * np.cos(th), -1.3 * np.sin(th), 7.3], [1.3 * np.sin(th), 1.3 * np.cos(th), -11.7]])
    out = A.warp_affine(img, M, (96, 96), 1).astype(np.float64)
    Minv = np.linalg.inv(np.vstack([M, [0, 0, 1]]))[:2]
    ys, xs = np.mgrid[0:96, 0:96].astype(np.float64)
    sx = Minv[0, 0] * xs + Minv[0, 1] * ys + Minv[0, 2]
    sy = Minv[1, 0] * xs + Minv[1, 1] * ys + Minv[1, 2]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    c = lambda v, n: np.clip(v, 0, n - 1)
    f = img.astype(np.float64)
    ref = ((1 - fy) * (1 - fx) * f[c(y0, 64), c(x0, 80)] + (1 - fy) * fx * f[c(y0, 64), c(x0 + 1, 80)]
           + fy * (1 - fx) * f[c(y0 + 1, 64), c(x0, 80)] + fy * fx * f[c(y0 + 1, 64), c(x0 + 1, 80)])
    # coordinates are quantised to 1/32 px, so allow the gradient * 1/32 plus one rounding LSB
    gx = np.abs(np.diff(f, axis=1)).max()
    assert np.abs(out - ref).max() <= 1.0 + 2 * gx / 32 + 1e-9
    assert np.abs(out - ref).mean() < 1.5


def test_crop_align_drops_degenerate_faces_and_unpads():
    imgs = np.stack([_img(64, 64, 7), _img(64, 64, 8)])
    tgt = A.landmarks_target((32, 32), 0.65)
    pad = np.array([[4, 4, 0, 0], [0, 0, 6, 6]])
    lm = np.stack([tgt * 1.5 + 3, np.ones((5, 2), np.float32), tgt * 1.2 + 5]).astype(np.float32)
    out = A.crop_align(imgs, pad, [0, 0, 1], lm, tgt, (32, 32), "reflect")
    assert out.shape == (2, 32, 32, 3) and out.dtype == np.uint8
    M = A.estimate_transform(lm[2], tgt)
    assert np.array_equal(out[1], A.warp_affine(imgs[1][:, 6:58], M, (32, 32), 2))
