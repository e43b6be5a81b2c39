"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement (torch CPU fp32 + numpy) of the reference's RRDBNet (BSRGAN x4)
enhancer.  Pinned against the reference module by tests/golden/make_golden.py.

Follows (paths relative to /root/reference/src/face_crop_plus):
  forward      models/rrdb.py:64-81, models/_layers.py:168-200
  predict      models/rrdb.py:124-146 (face-area gate, x4 SR, bicubic x0.25, clamp, round)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _c(x, sd, p):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], 1, 1)


def _rdb(x, sd, p):
    l = lambda t: F.leaky_relu(t, 0.2)
    x1 = l(_c(x, sd, p + ".conv1"))
    x2 = l(_c(torch.cat((x, x1), 1), sd, p + ".conv2"))
    x3 = l(_c(torch.cat((x, x1, x2), 1), sd, p + ".conv3"))
    x4 = l(_c(torch.cat((x, x1, x2, x3), 1), sd, p + ".conv4"))
    x5 = _c(torch.cat((x, x1, x2, x3, x4), 1), sd, p + ".conv5")
    return x5 * 0.2 + x


def forward(x, sd, n_blocks=23):
    fea0 = _c(x, sd, "conv_first")
    t = fea0
    for b in range(n_blocks):
        o = t
        for r in (1, 2, 3):
            o = _rdb(o, sd, f"RRDB_trunk.{b}.RDB{r}")
        t = o * 0.2 + t
    fea = fea0 + _c(t, sd, "trunk_conv")
    l = lambda v: F.leaky_relu(v, 0.2)
    fea = l(_c(F.interpolate(fea, scale_factor=2), sd, "upconv1"))
    fea = l(_c(F.interpolate(fea, scale_factor=2), sd, "upconv2"))
    return _c(l(_c(fea, sd, "HRconv")), sd, "conv_last")


def gate(landmarks, indices, n_images, h, w, min_face_factor):
    """Which images get enhanced (rrdb.py:124-140)."""
    out = []
    for i in range(n_images):
        if landmarks is None or indices is None:
            ff = np.array([min_face_factor])
        else:
            lm = landmarks[[idx == i for idx in indices]]
            if len(lm) == 0:
                out.append(False)
                continue
            wv, hv = (lm[:, 4] - lm[:, 0]).T
            ff = wv * hv / (h * w)
        out.append(bool(ff.mean() <= min_face_factor))
    return out


@torch.no_grad()
def predict(images, sd, landmarks, indices, min_face_factor=0.001, n_blocks=23):
    """images: torch (N,3,H,W) float 0..255 (modified copy returned)."""
    images = images.clone()
    g = gate(landmarks, indices, len(images), images[0].shape[1], images[0].shape[2], min_face_factor)
    for i, do in enumerate(g):
        if do:
            x4 = forward(images[i].unsqueeze(0).div(255), sd, n_blocks)
            x1 = F.interpolate(x4, None, 0.25, "bicubic")
            images[i] = x1.clamp(0, 1).mul(255).round()[0]
    return images
