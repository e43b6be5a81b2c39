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


def warp_affine(image, M, dsize, border=0):
    """image (h,w,3) uint8, M 2x3 forward transform, dsize=(width,height) -> (height,width,3) uint8."""
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
