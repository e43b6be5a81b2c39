"""Golden vectors for the reference's OpenCV *call plumbing* (build container only).

    python tests/golden/make_golden_plumbing.py

cv2 is not installed, so the pixel arithmetic of resize / warpAffine / imwrite cannot be pinned.  What CAN be pinned
is everything the reference decides around those calls: this script compiles ``as_batch`` (utils.py),
``Cropper.crop_align``, ``Cropper.save_group`` and ``Cropper.save_groups`` (cropper.py) out of the reference's
syntax trees and runs them against a recording stand-in for the ``cv2`` module.  The recorded arguments (target
sizes, interpolation and border constants, paddings, un-padded slices, dsize order, skipped faces, output paths
and their order) go to ``plumbing.json``.  No reference text is written to the repo.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_host import REF, extract  # noqa: E402


class FakeCv2:
    """Records calls; returns arrays of the right shape whose content encodes the call."""
    INTER_AREA, INTER_CUBIC = 3, 2
    BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
    COLOR_RGB2BGR = 4

    def __init__(self):
        self.calls = []

    def resize(self, image, dsize, interpolation=None):
        self.calls.append(["resize", list(image.shape), [int(dsize[0]), int(dsize[1])], int(interpolation)])
        return np.zeros((dsize[1], dsize[0], 3), np.uint8)

    def copyMakeBorder(self, image, t, b, l, r, borderType=None):
        self.calls.append(["copyMakeBorder", list(image.shape), [int(t), int(b), int(l), int(r)], int(borderType)])
        return np.zeros((image.shape[0] + t + b, image.shape[1] + l + r, 3), np.uint8)

    def estimateAffinePartial2D(self, src, dst, ransacReprojThreshold=None):
        self.calls.append(["estimateAffinePartial2D", np.asarray(src).tolist(), bool(np.isinf(ransacReprojThreshold))])
        m = None if np.asarray(src)[0, 0] < 0 else np.array([[1.0, 0, float(np.asarray(src)[0, 0])], [0, 1.0, 0]])
        return m, None

    def estimateAffine2D(self, src, dst, ransacReprojThreshold=None):
        self.calls.append(["estimateAffine2D", np.asarray(src).tolist(), bool(np.isinf(ransacReprojThreshold))])
        return np.array([[1.0, 0, 0], [0, 1.0, 0]]), None

    def warpAffine(self, image, m, dsize, borderMode=None):
        self.calls.append(["warpAffine", list(image.shape), int(image[0, 0, 0]), np.asarray(m).tolist(),
                           [int(dsize[0]), int(dsize[1])], int(borderMode)])
        return np.full((dsize[1], dsize[0], 3), image[0, 0, 0], np.uint8)

    def cvtColor(self, img, code):
        self.calls.append(["cvtColor", list(img.shape), int(code)])
        return img

    def imwrite(self, path, img):
        self.calls.append(["imwrite", path, list(img.shape)])
        return True


def main():
    out = {}
    # ---- as_batch: geometry + constants
    cv = FakeCv2()
    ns = {"np": np, "cv2": cv}
    extract(os.path.join(REF, "utils.py"), {"as_batch"}, ns)
    shapes = [(720, 1280), (50, 80), (90, 40), (64, 64), (20, 30), (2160, 3840), (1024, 1024), (1023, 1025), (1, 7), (600, 800)]
    for size in (512, (512, 256), 1024, (64, 64), (192, 128)):
        cv.calls.clear()
        imgs = [np.zeros(s + (3,), np.uint8) for s in shapes]
        batch, unscales, paddings = ns["as_batch"](imgs, size)
        out[f"as_batch_{size}"] = {"shapes": shapes, "calls": list(cv.calls), "batch_shape": list(batch.shape),
                                   "unscales": [float(u) for u in unscales], "paddings": np.asarray(paddings).tolist()}
    cv.calls.clear()
    ns["as_batch"]([np.zeros((30, 50, 3), np.uint8)], 64, padding_mode="reflect_101")
    out["as_batch_border_mode"] = list(cv.calls)

    # ---- Cropper.crop_align / save_group / save_groups
    cv = FakeCv2()
    ns = {"np": np, "cv2": cv, "os": os, "defaultdict": defaultdict}
    extract(os.path.join(REF, "cropper.py"), {"Cropper.crop_align", "Cropper.save_group", "Cropper.save_groups"}, ns)

    class Self:
        pass
    s = Self()
    s.padding, s.allow_skew, s.output_size = "reflect", False, (96, 112)
    s.landmarks_target = np.zeros((5, 2), np.float32)
    images = np.stack([np.full((40, 60, 3), v, np.uint8) for v in (10, 20, 30)])
    paddings = np.array([[0, 0, 0, 0], [3, 4, 0, 0], [0, 0, 5, 6]])
    indices = [0, 1, 1, 2, 2]
    lms = np.arange(50, dtype=np.float32).reshape(5, 5, 2)
    lms[2, 0, 0] = -1.0                                            # the stand-in estimator returns None for this face
    res = ns["crop_align"](s, images, paddings, indices, lms)
    out["crop_align"] = {"calls": list(cv.calls), "result_shape": list(res.shape), "result_values": res[:, 0, 0, 0].tolist()}
    cv.calls.clear()
    s.allow_skew, s.padding = True, "constant"
    res = ns["crop_align"](s, [np.full((8, 9, 3), 7, np.uint8)], None, [0], lms[:1])
    out["crop_align_skew_list_nopad"] = {"calls": list(cv.calls), "result_shape": list(res.shape)}
    cv.calls.clear()
    res = ns["crop_align"](s, images, paddings, [], lms[:0])
    out["crop_align_empty"] = {"result_shape": list(np.asarray(res).shape)}

    with tempfile.TemporaryDirectory() as d:
        faces = [np.zeros((4, 4, 3), np.uint8)] * 5 + [np.zeros((4, 4), np.uint8)]
        names = np.array(["a.jpg", "a.jpg", "b.png", "a.jpg", "c.jpeg", "b.png"])
        for strategy, fmt in (("all", None), ("largest", None), ("all", "png"), ("best", "jpg")):
            cv.calls.clear()
            s.strategy, s.output_format = strategy, fmt
            ns["save_group"](s, faces, names, os.path.join(d, "o"))
            out[f"save_group_{strategy}_{fmt}"] = [[c[0], os.path.relpath(c[1], d)] if c[0] == "imwrite" else c for c in cv.calls]
        cv.calls.clear()
        s.strategy, s.output_format = "all", None
        s.save_group = lambda f, n, o: ns["save_group"](s, f, n, o)
        attr = {"glasses": [0, 2, 4], "no_glasses": [1, 3]}
        masks = {"eyes": ([0, 1, 4], np.zeros((3, 4, 4), np.uint8)), "hair": ([2], np.zeros((1, 4, 4), np.uint8))}
        ns["save_groups"](s, faces[:5], names[:5], os.path.join(d, "g"), attr, masks)
        out["save_groups_attr_mask"] = [os.path.relpath(c[1], d) for c in cv.calls if c[0] == "imwrite"]
        cv.calls.clear()
        ns["save_groups"](s, faces[:5], names[:5], os.path.join(d, "n"), None, None)
        out["save_groups_none"] = [os.path.relpath(c[1], d) for c in cv.calls if c[0] == "imwrite"]
        cv.calls.clear()
        ns["save_groups"](s, faces[:5], names[:5], os.path.join(d, "m"), None, masks)
        out["save_groups_mask_only"] = [os.path.relpath(c[1], d) for c in cv.calls if c[0] == "imwrite"]
    json.dump(out, open(os.path.join(HERE, "plumbing.json"), "w"), indent=0)
    print("wrote plumbing.json:", sorted(out))


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
