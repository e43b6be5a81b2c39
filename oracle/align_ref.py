"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement (numpy, integer/fixed-point exact) of the align + crop stage,
reference ``cropper.py:392-439`` (_init_landmarks_target) and ``:441-552``
(crop_align), whose arithmetic lives in OpenCV.

PARITY UNPINNED for the OpenCV parts: ``opencv-python`` (version unpinned in
the reference's setup.py:39) is not installed here, not vendored in
/root/reference, and the reference has no tests or golden crops.  The two
functions below restate OpenCV's published algorithms:

* ``estimate_transform``  — cv::estimateAffinePartial2D / estimateAffine2D
  (calib3d/ptsetreg.cpp) with ransacReprojThreshold=inf: RANSAC's first minimal
  sample is accepted with every point an inlier, then 10 Levenberg-Marquardt
  iterations on all points of a residual that is *linear* in the parameters
  => the linear least-squares similarity (a,-b,tx; b,a,ty) / affine, float64.
* ``warp_affine``         — cv::warpAffine (imgproc/imgwarp.cpp), INTER_LINEAR,
  uint8: inverse map in double; per-column adelta/bdelta = cvRound(M*x*1024);
  per-row X0 = cvRound((M1*y+M2)*1024) + 16; X = (X0+adelta)>>5; integer part
  X>>5 (saturated to int16), 5-bit fraction; bilinear weights from
  BilinearTab_i = (32-fy)(32-fx)*32 ... (sum 32768; the saturated (0,0) entry
  {32767,0,0,1} gives the same pixel as {32768,0,0,0}); result
  (sum + 2^14) >> 15; borders via cv::borderInterpolate.

They are checked by self-consistency properties in tests/test_align_oracle.py
(LSQ optimality, exact recovery of known similarities, identity / integer
translations reproducing source bytes, documented border patterns, <= 1 LSB
from a float64 bilinear evaluation).

``estimate_transform_cv_sequence`` restates the SEQUENCE OpenCV actually runs (minimal-sample seed, then
``refineIters`` = 10 Levenberg-Marquardt iterations of calib3d's LMSolver) instead of its fixed point, so that
the distance between the two is a measured number rather than an argument: tests/test_align_oracle.py bounds
max |dM| over 10^4 random 5-point sets and counts flipped crop bytes (DESIGN.md §4 quotes both).  It stays
"parity unpinned" — the restatement of LMSolver is from OpenCV's published source, not from a cv2 run.
"""
from __future__ import annotations

import numpy as np

STANDARD_LANDMARKS_5 = np.float32([
    [0.31556875000000000, 0.4615741071428571],
    [0.68262291666666670, 0.4615741071428571],
    [0.50026249999999990, 0.6405053571428571],
    [0.34947187500000004, 0.8246919642857142],
    [0.65343645833333330, 0.8246919642857142],
])

BORDER = {"constant": 0, "replicate": 1, "reflect": 2, "wrap": 3, "reflect_101": 4, "reflect101": 4,
          "default": 4}


def landmarks_target(output_size, face_factor):
    """cropper.py:423-439 — float32 in-place arithmetic on the float32 table."""
    std = STANDARD_LANDMARKS_5.copy()
    std[:, 0] *= output_size[0] * face_factor
    std[:, 1] *= output_size[1] * face_factor
    std[:, 0] += (1 - face_factor) * output_size[0] / 2
    std[:, 1] += (1 - face_factor) * output_size[1] / 2
    return std


def estimate_transform(src, dst, allow_skew=False):
    """src, dst: (k,2) -> 2x3 float64 or None (degenerate / non-finite)."""
    s = np.asarray(src, np.float32).astype(np.float64)
    d = np.asarray(dst, np.float32).astype(np.float64)
    if not np.isfinite(s).all():
        return None
    k = len(s)
    mx = my = MX = MY = 0.0
    for p in range(k):           # same accumulation order as the kernel
        mx += s[p, 0]; my += s[p, 1]; MX += d[p, 0]; MY += d[p, 1]
    mx /= k; my /= k; MX /= k; MY /= k
    if not allow_skew:
        sxx = sa = sb = 0.0
        for p in range(k):
            x, y = s[p, 0] - mx, s[p, 1] - my
            X, Y = d[p, 0] - MX, d[p, 1] - MY
            sxx += x * x + y * y
            sa += x * X + y * Y
            sb += x * Y - y * X
        if not sxx > 0.0:
            return None
        a, b = sa / sxx, sb / sxx
        m = np.array([[a, -b, MX - a * mx + b * my], [b, a, MY - b * mx - a * my]])
    else:
        sxx = sxy = syy = sxX = syX = sxY = syY = 0.0
        for p in range(k):
            x, y = s[p, 0] - mx, s[p, 1] - my
            X, Y = d[p, 0] - MX, d[p, 1] - MY
            sxx += x * x; sxy += x * y; syy += y * y
            sxX += x * X; syX += y * X; sxY += x * Y; syY += y * Y
        det = sxx * syy - sxy * sxy
        if not abs(det) > 1e-12 * (sxx * syy + 1e-300):
            return None
        a = (sxX * syy - syX * sxy) / det; b = (syX * sxx - sxX * sxy) / det
        c = (sxY * syy - syY * sxy) / det; e = (syY * sxx - sxY * sxy) / det
        m = np.array([[a, b, MX - a * mx - b * my], [c, e, MY - c * mx - e * my]])
    return m if np.isfinite(m).all() else None


def _seed_model(s, d, idx, allow_skew):
    """Minimal-sample model: 2 points -> similarity (AffinePartial2DEstimatorCallback::runKernel), 3 points ->
    affine (Affine2DEstimatorCallback::runKernel).  Returns the parameter vector LM refines:
    (a, b, tx, ty) for [[a, -b, tx], [b, a, ty]]  or  (h0..h5) for [[h0, h1, h2], [h3, h4, h5]]."""
    if not allow_skew:
        z = complex(*s[idx[0]]) - complex(*s[idx[1]])
        Z = complex(*d[idx[0]]) - complex(*d[idx[1]])
        if z == 0:
            return None
        q = Z / z                                           # a + ib
        t = complex(*d[idx[0]]) - q * complex(*s[idx[0]])
        return np.array([q.real, q.imag, t.real, t.imag])
    A = np.array([[s[i, 0], s[i, 1], 1.0] for i in idx])
    if abs(np.linalg.det(A)) < 1e-12:
        return None
    return np.concatenate([np.linalg.solve(A, d[list(idx), 0]), np.linalg.solve(A, d[list(idx), 1])])


def _residual_jacobian(h, s, d, allow_skew):
    """AffinePartial2DRefineCallback / Affine2DRefineCallback::compute: reprojection error of every inlier
    (interleaved x, y) and its Jacobian — both are LINEAR in h."""
    x, y = s[:, 0], s[:, 1]
    k = len(s)
    if not allow_skew:
        J = np.zeros((2 * k, 4))
        J[0::2] = np.stack([x, -y, np.ones(k), np.zeros(k)], 1)
        J[1::2] = np.stack([y, x, np.zeros(k), np.ones(k)], 1)
    else:
        J = np.zeros((2 * k, 6))
        J[0::2, 0], J[0::2, 1], J[0::2, 2] = x, y, 1.0
        J[1::2, 3], J[1::2, 4], J[1::2, 5] = x, y, 1.0
    r = J @ h - d.reshape(-1)
    return r, J


def _lm_solver(h, s, d, allow_skew, max_iters=10, eps=float(np.finfo(np.float32).eps)):
    """cv::LMSolver (calib3d/levmarq.cpp, LMSolverImpl::run) restated: lambda starts at 1 on diag(JtJ), is halved
    (and dropped to 0 below 0.75) when the gain ratio R > 0.75, raised by nu in [2, 10] when R < 0.25; a step is
    accepted iff it lowers the squared error; stops after max_iters or when |step|_inf or |residual|_inf < eps.
    Returns (h, iterations used, trace of the accepted squared errors)."""
    x = np.asarray(h, np.float64).copy()
    r, J = _residual_jacobian(x, s, d, allow_skew)
    S = float(r @ r)
    A, v = J.T @ J, J.T @ r
    D = np.diag(A).copy()
    lam, lc, it, trace = 1.0, 0.75, 0, [S]
    while True:
        Ap = A + np.diag(lam * D)
        try:
            step = np.linalg.solve(Ap, v)
        except np.linalg.LinAlgError:
            return None, it, trace
        xd = x - step
        rd, _ = _residual_jacobian(xd, s, d, allow_skew)
        Sd = float(rd @ rd)
        dS = float(step @ (2 * v - A @ step))
        R = (S - Sd) / (dS if abs(dS) > np.finfo(np.float64).eps else 1.0)
        if R > 0.75:
            lam *= 0.5
            if lam < lc:
                lam = 0.0
        elif R < 0.25:
            t = float(step @ v)
            nu = (Sd - S) / (t if abs(t) > np.finfo(np.float64).eps else 1.0) + 2
            nu = min(max(nu, 2.0), 10.0)
            if lam == 0:
                inv_diag = np.abs(np.diag(np.linalg.inv(A)))
                lam = lc = 1.0 / max(float(inv_diag.max()), np.finfo(np.float64).eps)
                nu *= 0.5
            lam *= nu
        if Sd < S:
            S, x = Sd, xd
            r, J = _residual_jacobian(x, s, d, allow_skew)
            A, v = J.T @ J, J.T @ r
            trace.append(S)
        it += 1
        if not (it < max_iters and np.abs(step).max() >= eps and np.abs(r).max() >= eps):
            break
    return x, it, trace


def estimate_transform_cv_sequence(src, dst, allow_skew=False, seed=None, refine_iters=10, return_info=False):
    """cv::estimateAffinePartial2D / estimateAffine2D as the SEQUENCE OpenCV runs for the reference's call
    (cropper.py:515-527: method RANSAC, ransacReprojThreshold = inf, refineIters = 10; calib3d/ptsetreg.cpp):

    1. RANSAC draws one minimal sample (2 points / 3 for the affine form).  With an infinite threshold every
       point is an inlier of that first model, so the iteration count collapses to zero and the sample's model
       is "the best".  ``seed``: the sample's point indices (default: the first non-degenerate one in index
       order — which sample cv::RNG draws is irrelevant, the result below does not depend on it; the tests run
       every possible sample).
    2. 10 iterations of LMSolver on the reprojection error of all inliers, starting from that model.

    -> 2x3 float64 (or None)."""
    s = np.asarray(src, np.float32).astype(np.float64)
    d = np.asarray(dst, np.float32).astype(np.float64)
    if not (np.isfinite(s).all() and np.isfinite(d).all()):
        return None
    import itertools
    m = 3 if allow_skew else 2
    samples = [tuple(seed)] if seed is not None else list(itertools.combinations(range(len(s)), m))
    h0 = None
    for idx in samples:                       # RANSAC re-draws while the sample is degenerate
        h0 = _seed_model(s, d, idx, allow_skew)
        if h0 is not None and np.isfinite(h0).all():
            break
        h0 = None
    if h0 is None:
        return None
    h, iters, trace = _lm_solver(h0, s, d, allow_skew, refine_iters)
    if h is None or not np.isfinite(h).all():
        return None
    M = np.array([[h[0], -h[1], h[2]], [h[1], h[0], h[3]]]) if not allow_skew else h.reshape(2, 3)
    return (M, iters, trace) if return_info else M


def _cv_round(v):
    v = np.asarray(v, np.float64)
    out = np.full(v.shape, -2147483648, np.int64)
    ok = (v >= -2147483648.0) & (v < 2147483648.0)
    out[ok] = np.rint(v[ok]).astype(np.int64)
    return out


def _wrap32(v):
    return ((np.asarray(v, np.int64) + 2 ** 31) % 2 ** 32) - 2 ** 31


def border_interpolate(p, length, border):
    """Vectorised cv::borderInterpolate. border: cv2.BORDER_* code."""
    p = np.asarray(p, np.int64).copy()
    if border == 0:
        return np.where((p >= 0) & (p < length), p, -1)
    if border == 1:
        return np.clip(p, 0, length - 1)
    if border in (2, 4):
        delta = 1 if border == 4 else 0
        if length == 1:
            return np.zeros_like(p)
        while True:
            bad = (p < 0) | (p >= length)
            if not bad.any():
                return p
            neg = p < 0
            p = np.where(neg, -p - 1 + delta, p)
            hi = (~neg) & (p >= length)
            p = np.where(hi, length - 1 - (p - length) - delta, p)
    if border == 3:
        return np.mod(p, length)
    raise ValueError(border)


def warp_affine_float32(image, M, dsize, border=0):
    """The OTHER family of cv2.warpAffine(INTER_LINEAR) implementations: source coordinates and bilinear weights in float32,
    one rounding to uint8 at the end (cvRound = round-half-even), as the SIMD "linear" warp kernels of recent OpenCV 4.x / 5.x
    releases evaluate it, instead of the classic 5-bit-fraction / 15-bit-weight fixed-point tables that ``warp_affine``
    restates (imgproc's WarpAffineInvoker + remapBilinear).  ``opencv-python`` is unpinned in the reference (setup.py:39), so
    either may be what a user's wheel runs.  RESTATED FROM MEMORY OF THE PUBLISHED SOURCE, NOT PINNED: this variant exists so that
    a fixture produced by such a wheel (tools/make_cv2_fixture.py records cv2.__version__ and the CPU dispatch) is recognisable
    as "the float family, within one grey level of the fixed-point result" rather than as a defect; the HIP kernel implements
    the classic algorithm.  Steps, all float32: inverse map as in ``warp_affine`` (float64, then rounded to float32);
    sx = M0 x + M1 y + M2; ix = floor(sx), a = sx - ix; row lerp v0 = p00 + a (p01 - p00), v1 likewise; v = v0 + b (v1 - v0)."""
    f = np.float32
    img = np.asarray(image, np.uint8)
    sh, sw = img.shape[:2]
    ow, oh = int(dsize[0]), int(dsize[1])
    m = np.asarray(M, np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    m = m.astype(f)
    xs, ys = np.arange(ow, dtype=f)[None, :], np.arange(oh, dtype=f)[:, None]
    sx = (xs * m[0] + (ys * m[1] + m[2]).astype(f)).astype(f)
    sy = (xs * m[3] + (ys * m[4] + m[5]).astype(f)).astype(f)
    ix, iy = np.floor(sx), np.floor(sy)
    ax, ay = (sx - ix).astype(f)[..., None], (sy - iy).astype(f)[..., None]
    ix, iy = np.clip(ix, -32768, 32767).astype(np.int64), np.clip(iy, -32768, 32767).astype(np.int64)
    if border == 1:
        ci = lambda p, n: np.clip(p, 0, n - 1)
        x0, x1, y0, y1 = ci(ix, sw), ci(ix + 1, sw), ci(iy, sh), ci(iy + 1, sh)
    else:
        x0 = border_interpolate(ix, sw, border); x1 = border_interpolate(ix + 1, sw, border)
        y0 = border_interpolate(iy, sh, border); y1 = border_interpolate(iy + 1, sh, border)

    def tap(yy, xx):
        okm = (yy >= 0) & (xx >= 0)
        v = img[np.where(okm, yy, 0), np.where(okm, xx, 0)].astype(f)
        return np.where(okm[..., None], v, f(0))

    p00, p01, p10, p11 = tap(y0, x0), tap(y0, x1), tap(y1, x0), tap(y1, x1)
    v0 = (p00 + ax * (p01 - p00)).astype(f)
    v1 = (p10 + ax * (p11 - p10)).astype(f)
    v = (v0 + ay * (v1 - v0)).astype(f)
    out = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    if border == 0:
        outside = (ix >= sw) | (ix + 1 < 0) | (iy >= sh) | (iy + 1 < 0)
        out[outside] = 0
    return out


def warp_affine(image, M, dsize, border=0, variant="fixed"):
    """image (h,w,3) uint8, M 2x3 forward transform, dsize=(width,height) -> (height,width,3) uint8.
    ``variant``: "fixed" = OpenCV's classic fixed-point algorithm (what the HIP kernel implements), "float32" = the float
    family of newer wheels (``warp_affine_float32``)."""
    if variant == "float32":
        return warp_affine_float32(image, M, dsize, border)
    if variant != "fixed":
        raise ValueError(f"unknown warp variant {variant!r}")
    img = np.asarray(image, np.uint8)
    sh, sw = img.shape[:2]
    ow, oh = int(dsize[0]), int(dsize[1])
    m = np.asarray(M, np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    xs = np.arange(ow, dtype=np.float64)
    ys = np.arange(oh, dtype=np.float64)
    adelta = _cv_round(m[0] * xs * 1024.0)
    bdelta = _cv_round(m[3] * xs * 1024.0)
    X0 = _wrap32(_cv_round((m[1] * ys + m[2]) * 1024.0) + 16)
    Y0 = _wrap32(_cv_round((m[4] * ys + m[5]) * 1024.0) + 16)
    X = _wrap32(X0[:, None] + adelta[None, :]) >> 5
    Y = _wrap32(Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w0 = (32 - fy) * (32 - fx) * 32
    w1 = (32 - fy) * fx * 32
    w2 = fy * (32 - fx) * 32
    w3 = fy * fx * 32

    if border == 1:
        def ci(p, n):
            return np.clip(p, 0, n - 1)
        sx0, sx1, sy0, sy1 = ci(sx, sw), ci(sx + 1, sw), ci(sy, sh), ci(sy + 1, sh)
    else:
        sx0 = border_interpolate(sx, sw, border); sx1 = border_interpolate(sx + 1, sw, border)
        sy0 = border_interpolate(sy, sh, border); sy1 = border_interpolate(sy + 1, sh, border)

    def tap(yy, xx):
        okm = (yy >= 0) & (xx >= 0)
        v = img[np.where(okm, yy, 0), np.where(okm, xx, 0)].astype(np.int64)
        return np.where(okm[..., None], v, 0)

    acc = (tap(sy0, sx0) * w0[..., None] + tap(sy0, sx1) * w1[..., None]
           + tap(sy1, sx0) * w2[..., None] + tap(sy1, sx1) * w3[..., None])
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    if border == 0:
        outside = (sx >= sw) | (sx + 1 < 0) | (sy >= sh) | (sy + 1 < 0)
        out[outside] = 0
    return out


def crop_align(images, padding, indices, landmarks_source, landmarks_tgt, output_size,
               border="constant", allow_skew=False):
    """cropper.py:510-552.  images: (N,H,W,3) uint8 or list; -> (F,Hout,Wout,3) uint8."""
    b = BORDER[border.lower()] if isinstance(border, str) else border
    outs = []
    for li, ii in enumerate(indices):
        M = estimate_transform(landmarks_source[li], landmarks_tgt, allow_skew)
        if M is None:
            continue
        image = images[ii]
        if padding is not None:
            t, bb, l, r = [int(v) for v in padding[ii]]
            image = image[t:image.shape[0] - bb, l:image.shape[1] - r]
        outs.append(warp_affine(image, M, output_size, b))
    return np.stack(outs) if outs else np.array(outs)
