"""Third-party check of the warp's GEOMETRY (reference cropper.py:542-547): scikit-image 0.18.3's
`transform.warp(order=1)` — an independent bilinear resampler with the same conventions as cv2.warpAffine without
WARP_INVERSE_MAP (M maps source to destination, integer coordinates are pixel centres, constant border = blend with 0) — on
smooth photo-like images.  It evaluates the bilinear filter in float64; OpenCV's fixed-point path (1/32-pixel coordinates,
15-bit weights) that the oracle / kernel restate must agree with it to about one grey level on such images, and a wrong
convention (half-pixel offset, inverted map, swapped axes) would not.  OpenCV itself is not in the image: this pins the
geometry, not the rounding.

    /opt/conda/bin/python3.9 tests/golden/make_golden_skimage_warp.py     # writes tests/golden/skimage_warp.npz"""
import os
import warnings

import numpy as np

warnings.simplefilter("ignore")
import skimage
from skimage.transform import warp

HERE = os.path.dirname(os.path.abspath(__file__))


def photo_like(h, w, seed):                      # smooth structure, gradients of a few grey levels per pixel, no noise
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 90 * np.sin(xx / (17 + 9 * c) + c + seed) * np.cos(yy / (23 - 5 * c)) for c in range(3)], -1)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


rng = np.random.default_rng(77)
out = {"skimage_version": np.array(skimage.__version__)}
k = 0
for (h, w), (ow, oh) in (((97, 131), (64, 48)), ((180, 240), (96, 96)), ((200, 160), (112, 128))):
    img = photo_like(h, w, k)
    mats, res = [], []
    for j in range(3):
        ang, sc = rng.uniform(-0.5, 0.5), rng.uniform(0.5, 2.0)       # M: source -> destination (similarity + shift)
        a, b = sc * np.cos(ang), sc * np.sin(ang)
        centre = np.array([w, h]) * rng.uniform(0.3, 0.7, 2)
        tx, ty = ow / 2 - (a * centre[0] - b * centre[1]), oh / 2 - (b * centre[0] + a * centre[1])
        if j == 2:
            tx += ow * 0.6                                              # part of the output looks outside the image
        M = np.array([[a, -b, tx], [b, a, ty]], np.float64)
        inv = np.linalg.inv(np.vstack([M, [0, 0, 1]]))                  # skimage wants output -> input coordinates
        r = warp(img.astype(np.float64), inv, output_shape=(oh, ow), order=1, mode="constant", cval=0.0, preserve_range=True)
        mats.append(M); res.append(np.rint(r * 8).astype(np.uint16))       # eighths of a grey level
    out[f"img{k}"], out[f"mat{k}"], out[f"dsize{k}"], out[f"x8_{k}"] = img, np.stack(mats), np.array([ow, oh]), np.stack(res)
    k += 1
out["cases"] = np.array(k)
np.savez_compressed(os.path.join(HERE, "skimage_warp.npz"), **out)
print("wrote skimage_warp.npz", k, "images x 3 matrices", skimage.__version__)
