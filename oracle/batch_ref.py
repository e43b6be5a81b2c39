"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement (numpy, integer / float32 step-exact) of the batch builder that
precedes the detector: reference ``utils.py:273-342`` (``as_batch``), whose pixel
arithmetic lives in OpenCV (``cv2.resize`` INTER_AREA / INTER_CUBIC on uint8 and
``cv2.copyMakeBorder``, call sites ``utils.py:334-335``).

PARITY UNPINNED for the OpenCV parts: ``opencv-python`` (version unpinned in the
reference's setup.py:39) is not installed here, not vendored in /root/reference,
and the reference has no tests.  What is restated is OpenCV 4.x's portable C++
path of ``cv::resize`` (imgproc/resize.cpp):

* same-size input: plain copy.
* INTER_CUBIC, 8U: ``scale = 1./((double)dsize/ssize)``; per destination
  column ``fx = (float)((dx+0.5)*scale-0.5)``, ``sx = floor(fx)``, the four
  Keys coefficients (A = -0.75) evaluated in float32 by ``interpolateCubic``
  and rounded to int16 with scale 2^11; horizontal pass in int32 with
  replicated edge columns (HResizeCubic), vertical pass over rows clipped to
  the image, ``(v + 2^21) >> 22`` saturated (FixedPtCast).  (OpenCV's SIMD
  VResizeCubicVec_32s8u evaluates the vertical pass in float32 instead; the two
  agree except when the exact value lies within ~1e-5 of a half.)
* INTER_AREA with both scales >= 1:
  - both scales integral (|scale - round(scale)| < DBL_EPSILON): box sums,
    ``(s+2)>>2`` for 2x2 (ResizeAreaFastVec), else
    ``cvRound(float(sum) * (1.f/area))`` (resizeAreaFast_Invoker);
  - otherwise ``computeResizeAreaTab`` (double) -> float32 alpha tables, a
    float32 row accumulation ``buf += S*alpha`` in table order, a float32
    column accumulation ``sum (+)= beta*buf`` in table order, ``cvRound``
    (ResizeArea_Invoker).
* ``copyMakeBorder`` through ``borderInterpolate``'s index patterns.

Checked by the self-consistency properties of tests/test_batch_oracle.py.
"""
from __future__ import annotations

import math

import numpy as np

DBL_EPSILON = 2.220446049250313e-16
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS
F32 = np.float32


# ------------------------------------------------------------------ geometry
def geometry(h: int, w: int, size):
    """utils.py:316-331 -> (ww, hh, [t,b,l,r], unscale, interpolation 'area'|'cubic')."""
    size = (size, size) if isinstance(size, int) else tuple(size)
    m = max(h, w)
    interp = "area" if m > max(size) else "cubic"
    ratio_w, ratio_h = size[0] / w, size[1] / h
    if ratio_w < ratio_h:
        unscale = ratio_w
        ww, hh = size[0], int(h * ratio_w)
        padding = [(size[1] - hh) // 2, (size[1] - hh + 1) // 2, 0, 0]
    else:
        unscale = ratio_h
        ww, hh = int(w * ratio_h), size[1]
        padding = [0, 0, (size[0] - ww) // 2, (size[0] - ww + 1) // 2]
    return ww, hh, padding, unscale, interp


def _scale(ssize: int, dsize: int) -> float:
    inv = float(dsize) / float(ssize)          # inv_scale_x = (double)dsize.width / ssize.width
    return 1.0 / inv                           # scale_x = 1. / inv_scale_x


# --------------------------------------------------------------------- cubic
def _cubic_coeffs(x: np.ndarray) -> np.ndarray:
    """interpolateCubic in float32, operation by operation."""
    x = x.astype(F32)
    A = F32(-0.75)
    x1 = x + F32(1)
    c0 = ((A * x1 - F32(5) * A) * x1 + F32(8) * A) * x1 - F32(4) * A
    c1 = ((A + F32(2)) * x - (A + F32(3))) * x * x + F32(1)
    y = F32(1) - x
    c2 = ((A + F32(2)) * y - (A + F32(3))) * y * y + F32(1)
    c3 = F32(1) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], 1).astype(F32)


def _cubic_tab(ssize: int, dsize: int):
    scale = _scale(ssize, dsize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(F32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(F32)
    coef = _cubic_coeffs(f) * F32(COEF_SCALE)
    icoef = np.clip(np.rint(coef), -32768, 32767).astype(np.int64)     # saturate_cast<short>(float): cvRound
    return s, icoef


def resize_cubic_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    sh, sw = src.shape[:2]
    sx, ia = _cubic_tab(sw, dw)
    sy, ib = _cubic_tab(sh, dh)
    S = src.astype(np.int64)
    H = np.zeros((sh, dw) + src.shape[2:], np.int64)
    for j in range(4):
        cols = np.clip(sx - 1 + j, 0, sw - 1)
        H += S[:, cols] * ia[:, j].reshape((1, dw) + (1,) * (src.ndim - 2))
    V = np.zeros((dh, dw) + src.shape[2:], np.int64)
    for k in range(4):
        rows = np.clip(sy - 1 + k, 0, sh - 1)
        V += H[rows] * ib[:, k].reshape((dh,) + (1,) * (src.ndim - 1))
    assert np.abs(V).max() < 2 ** 31                                    # OpenCV accumulates in int32
    return np.clip((V + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------- area
def area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab -> per destination index the list of (si, alpha float32)."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        ent = []
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, F32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, F32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, F32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def _accumulate(S: np.ndarray, tab, axis_len: int) -> np.ndarray:
    """out[d] = sequential float32 sum over the entries of tab[d] of S[si]*alpha (first axis)."""
    out = np.zeros((axis_len,) + S.shape[1:], F32)
    maxn = max(len(e) for e in tab)
    for j in range(maxn):
        ds = np.array([d for d, e in enumerate(tab) if len(e) > j], np.int64)
        si = np.array([tab[d][j][0] for d in ds], np.int64)
        al = np.array([tab[d][j][1] for d in ds], F32).reshape((-1,) + (1,) * (S.ndim - 1))
        prod = (S[si] * al).astype(F32)
        out[ds] = prod if j == 0 else (out[ds] + prod).astype(F32)
    return out


def resize_area_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    sh, sw = src.shape[:2]
    scale_x, scale_y = _scale(sw, dw), _scale(sh, dh)
    assert scale_x >= 1 and scale_y >= 1, "as_batch only decimates with INTER_AREA"
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))
    if abs(scale_x - isx) < DBL_EPSILON and abs(scale_y - isy) < DBL_EPSILON:
        assert dw * isx == sw and dh * isy == sh
        cells = src.reshape((dh, isy, dw, isx) + src.shape[2:]).astype(np.int64).sum((1, 3))
        if isx == 2 and isy == 2:
            return ((cells + 2) >> 2).astype(np.uint8)
        inv_area = F32(1.0) / F32(isx * isy)
        return np.clip(np.rint(cells.astype(F32) * inv_area), 0, 255).astype(np.uint8)
    xtab, ytab = area_tab(sw, dw, scale_x), area_tab(sh, dh, scale_y)
    Sf = src.astype(F32)
    buf = _accumulate(np.swapaxes(Sf, 0, 1), xtab, dw)                  # (dw, sh, c): row pass, table order
    buf = np.swapaxes(buf, 0, 1)                                        # (sh, dw, c)
    out = _accumulate(buf, ytab, dh)                                    # (dh, dw, c): column pass, table order
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_u8(src: np.ndarray, dw: int, dh: int, interp: str) -> np.ndarray:
    """cv2.resize(src, (dw, dh), interpolation=INTER_AREA|INTER_CUBIC) on uint8."""
    if (src.shape[1], src.shape[0]) == (dw, dh):
        return src.copy()
    return resize_area_u8(src, dw, dh) if interp == "area" else resize_cubic_u8(src, dw, dh)


# -------------------------------------------------------------------- border
def border_index(p: np.ndarray, n: int, mode: str) -> np.ndarray:
    """cv::borderInterpolate for replicate / reflect / reflect_101 / wrap."""
    p = np.asarray(p, np.int64)
    if mode == "replicate":
        return np.clip(p, 0, n - 1)
    if mode == "wrap":
        return np.mod(p, n)
    if mode in ("reflect", "reflect_101", "reflect101", "default"):
        if n == 1:
            return np.zeros_like(p)
        delta = 1 if mode != "reflect" else 0
        q = p.copy()
        for _ in range(64):
            bad = (q < 0) | (q >= n)
            if not bad.any():
                break
            q = np.where(q < 0, -q - 1 + delta, q)
            q = np.where(q >= n, n - 1 - (q - n) - delta, q)
        return q
    raise ValueError(mode)


def copy_make_border(img: np.ndarray, t: int, b: int, l: int, r: int, mode: str = "constant") -> np.ndarray:
    h, w = img.shape[:2]
    mode = mode.lower()
    if mode == "constant":
        out = np.zeros((h + t + b, w + l + r) + img.shape[2:], img.dtype)
        out[t:t + h, l:l + w] = img
        return out
    ys = border_index(np.arange(-t, h + b), h, mode)
    xs = border_index(np.arange(-l, w + r), w, mode)
    return img[ys][:, xs]


def as_batch(images, size=512, padding_mode: str = "constant"):
    """utils.py:273-342 -> (batch (N,H,W,3) u8, unscales (N,), paddings (N,4) int64 [t,b,l,r])."""
    batch, unscales, paddings = [], [], []
    for image in images:
        h, w = image.shape[:2]
        ww, hh, padding, unscale, interp = geometry(h, w, size)
        image = resize_u8(image, ww, hh, interp)
        batch.append(copy_make_border(image, *padding, mode=padding_mode))
        unscales.append(np.array(unscale))
        paddings.append(np.array(padding))
    return np.stack(batch), np.stack(unscales), np.stack(paddings)
