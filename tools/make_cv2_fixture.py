"""Record what OpenCV / torchvision THEMSELVES compute on seeded inputs, so that the "parity unpinned" rows of
DESIGN.md section 4 (a13 estimateAffine*2D, a14 warpAffine, f1 resize / copyMakeBorder, a3 ResNet-50 body)
can be pinned on any machine that has the wheels.  The build container has neither (no network), so the files this
script writes are absent there and `tests/test_third_party_pins.py` skips with that reason; once they exist the same
tests compare BOTH the CPU oracle (`oracle/align_ref.py`, `oracle/batch_ref.py`, `oracle/retinaface_ref.py`) and the HIP
kernels to them.

    pip install opencv-python torchvision          # anywhere with a network
    python tools/make_cv2_fixture.py                # writes tests/golden/opencv_align.npz, opencv_batch.npz,
                                                    #        tests/golden/torchvision_resnet50.npz
    python -m pytest tests/test_third_party_pins.py            # oracle vs the pins (CPU)
    python -m pytest tests/test_third_party_pins.py -m gpu     # kernels vs the pins (MI355X)

Everything is called exactly the way the reference calls it:
  cropper.py:515-527   cv2.estimateAffinePartial2D / estimateAffine2D(src, dst, ransacReprojThreshold=np.inf)[0]
  cropper.py:542-547   cv2.warpAffine(image, M, dsize, borderMode=cv2.BORDER_*)          (flags default INTER_LINEAR)
  utils.py:320-335     cv2.resize(image, (ww, hh), interpolation=INTER_AREA | INTER_CUBIC) + cv2.copyMakeBorder
  retinaface.py:93-99  torchvision.models.resnet50() + _utils.IntermediateLayerGetter(layer2, layer3, layer4)
Only data is stored (inputs, outputs, library versions): no third-party source.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

BORDERS = ("constant", "replicate", "reflect", "wrap", "reflect_101")     # codes 0..4 of align.BORDER_MODES


def standard_target(out_w, out_h, face_factor=0.65):
    """The reference's 5-point target (cropper.py:392-439) via this repo's host code (pinned by host_logic.npz)."""
    from face_crop_plus_amd.cropper import landmarks_target
    return landmarks_target((out_w, out_h), face_factor)


def photo_like(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 90 * np.sin(xx / (17 + 9 * c) + c) * np.cos(yy / (23 - 5 * c)) for c in range(3)], -1)
    img += rng.normal(0, 25, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def describe_cv2(cv2):
    """What produced the numbers: the wheel's version, the CPU baseline / dispatch lines of its build, whether its optimised
    paths were on.  `opencv-python` is unpinned in the reference (setup.py:39) and its warp / resize kernels differ between
    releases and CPU dispatch levels: a fixture must say which algorithm family it records."""
    import platform
    info = cv2.getBuildInformation()
    keep = [ln.strip() for ln in info.splitlines()
            if any(k in ln for k in ("Version control", "CPU/HW features", "Baseline:", "Dispatched code generation", "requested:",
                                     "Use IPP", "Parallel framework", "Platform", "Host:"))
            or ln.strip().startswith(("SSE", "AVX", "NEON", "FP16", "VSX", "RVV"))]
    return {"cv2_version": np.array(cv2.__version__), "cv2_build_cpu": np.array(" | ".join(keep)),
            "cv2_use_optimized": np.array(bool(cv2.useOptimized())), "cv2_num_threads": np.array(int(cv2.getNumThreads())),
            "numpy_version": np.array(np.__version__), "python_version": np.array(platform.python_version()),
            "machine": np.array(platform.machine() + " " + (platform.processor() or ""))}


def make_align(cv2):
    rng = np.random.default_rng(101)
    out = describe_cv2(cv2)
    # ---- estimators: 64 five-point sets (similarity-like faces at several scales / rotations + jitter) and degenerate ones
    tgt = standard_target(256, 256)
    srcs = []
    for i in range(64):
        ang, sc = rng.uniform(-0.6, 0.6), rng.uniform(0.3, 4.0)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]) * sc
        srcs.append((tgt @ R.T + rng.uniform(0, 3000, 2) + rng.normal(0, 2.0 * sc, (5, 2))).astype(np.float32))
    srcs.append(np.full((5, 2), 7.0, np.float32))                         # all points equal
    srcs.append(np.stack([np.linspace(0, 40, 5), np.linspace(0, 40, 5)], 1).astype(np.float32))   # collinear
    srcs = np.stack(srcs)
    for name, fn in (("partial", cv2.estimateAffinePartial2D), ("affine", cv2.estimateAffine2D)):
        mats, ok = np.zeros((len(srcs), 2, 3), np.float64), np.zeros(len(srcs), np.int32)
        for i, s in enumerate(srcs):
            m = fn(s, tgt, ransacReprojThreshold=np.inf)[0]
            if m is not None:
                mats[i], ok[i] = m, 1
        out[f"est_{name}_mat"], out[f"est_{name}_ok"] = mats, ok
    out["est_src"], out["est_dst"] = srcs, tgt
    # ---- warpAffine: 3 source sizes x 5 borders x 4 matrices (incl. one that looks far outside the image)
    k = 0
    for (h, w), (ow, oh) in (((97, 131), (64, 48)), ((240, 320), (128, 128)), ((512, 384), (256, 256))):
        img = photo_like(h, w, 200 + k)
        t = standard_target(ow, oh)
        mats = []
        for j in range(4):
            ang, sc = rng.uniform(-0.5, 0.5), rng.uniform(0.4, 2.5)
            R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]) * sc
            centre = np.array([w, h]) * (rng.uniform(0.2, 0.8, 2) if j < 3 else np.array([1.1, -0.2]))
            s = ((t - t.mean(0)) @ R.T + centre).astype(np.float32)
            mats.append(cv2.estimateAffinePartial2D(s, t, ransacReprojThreshold=np.inf)[0])
        mats = np.stack(mats)
        out[f"warp{k}_img"], out[f"warp{k}_mat"], out[f"warp{k}_dsize"] = img, mats, np.array([ow, oh])
        for b in BORDERS:
            mode = getattr(cv2, f"BORDER_{b.upper()}")
            out[f"warp{k}_{b}"] = np.stack([cv2.warpAffine(img, m, (ow, oh), borderMode=mode) for m in mats])
            # the same calls with the wheel's optimised (IPP / HAL / SIMD-dispatch) paths switched off: the portable C++
            # implementation of the same release, i.e. the other algorithm family when the two differ
            cv2.setUseOptimized(False)
            try:
                out[f"warp{k}_{b}_noopt"] = np.stack([cv2.warpAffine(img, m, (ow, oh), borderMode=mode) for m in mats])
            finally:
                cv2.setUseOptimized(True)
        k += 1
    out["warp_cases"] = np.array(k)
    # which algorithm family this wheel's warpAffine belongs to, decided against the oracle's two variants exactly as
    # tests/test_third_party_pins.py::_warp_family does; recorded so that a later re-classification fails loudly
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import align_ref as A
    worst = {"fixed": 0, "float32": 0}
    for c in range(k):
        for b in BORDERS:
            for j, m in enumerate(out[f"warp{c}_mat"]):
                for variant in worst:
                    got = A.warp_affine(out[f"warp{c}_img"], m, tuple(int(v) for v in out[f"warp{c}_dsize"]), A.BORDER[b], variant=variant)
                    worst[variant] = max(worst[variant], int(np.abs(got.astype(int) - out[f"warp{c}_{b}"][j].astype(int)).max()))
    out["warp_family"] = np.array("fixed" if worst["fixed"] == 0 else ("float32" if worst["float32"] <= 1 else "unknown"))
    print("warpAffine family of this wheel:", out["warp_family"], worst)
    np.savez_compressed(os.path.join(GOLDEN, "opencv_align.npz"), **out)
    print("wrote opencv_align.npz", {n: v.shape for n, v in out.items() if n.startswith("est")})


def make_batch(cv2):
    """utils.py:320-335 for down-scaling (INTER_AREA) and up-scaling (INTER_CUBIC) inputs, integer and non-integer ratios,
    plus the five border types of copyMakeBorder."""
    out = describe_cv2(cv2)
    cases = [((270, 480), 128), ((135, 240), 256), ((600, 400), 200), ((301, 517), 224), ((64, 48), 160), ((540, 960), 256)]
    for k, ((h, w), size) in enumerate(cases):
        img = photo_like(h, w, 300 + k)
        interp = cv2.INTER_AREA if max(h, w) > size else cv2.INTER_CUBIC
        if (rw := size / w) < (rh := size / h):
            ww, hh = size, int(h * rw)
            pad = [(size - hh) // 2, (size - hh + 1) // 2, 0, 0]
        else:
            ww, hh = int(w * rh), size
            pad = [0, 0, (size - ww) // 2, (size - ww + 1) // 2]
        res = cv2.resize(img, (ww, hh), interpolation=interp)
        out[f"batch{k}_img"], out[f"batch{k}_size"], out[f"batch{k}_resized"] = img, np.array(size), res
        out[f"batch{k}_pad"] = np.array(pad)
        for b in BORDERS:
            out[f"batch{k}_{b}"] = cv2.copyMakeBorder(res, *pad, borderType=getattr(cv2, f"BORDER_{b.upper()}"))
    out["batch_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(GOLDEN, "opencv_batch.npz"), **out)
    print("wrote opencv_batch.npz")


def make_resnet():
    """retinaface.py:93-99 with the build's generated `body.*` weights: the three feature maps torchvision's own
    ResNet-50 + IntermediateLayerGetter return for one seeded 96x128 input."""
    import torch
    import torchvision
    from torchvision.models import _utils, resnet50
    from face_crop_plus_amd import weights
    sd = weights.generate_state_dict("retinaface")
    body = _utils.IntermediateLayerGetter(resnet50(), {"layer2": 1, "layer3": 2, "layer4": 3})
    own = {k[len("body."):]: v for k, v in sd.items() if k.startswith("body.")}
    missing, unexpected = body.load_state_dict(own, strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing, unexpected)
    body.eval()
    x = torch.from_numpy(np.random.default_rng(400).normal(0, 50, (1, 3, 96, 128)).astype(np.float32))
    with torch.no_grad():
        feats = body(x)
    np.savez_compressed(os.path.join(GOLDEN, "torchvision_resnet50.npz"), x=x.numpy(),
                        torchvision_version=np.array(torchvision.__version__), torch_version=np.array(torch.__version__),
                        **{f"feat{k}": v.numpy() for k, v in feats.items()})
    print("wrote torchvision_resnet50.npz", {k: tuple(v.shape) for k, v in feats.items()})


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--only", choices=["cv2", "torchvision"], default=None)
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    if args.only != "torchvision":
        try:
            import cv2
        except ImportError:
            print("cv2 is not installed: opencv_*.npz not written (pip install opencv-python)")
        else:
            make_align(cv2)
            make_batch(cv2)
    if args.only != "cv2":
        try:
            import torchvision  # noqa: F401
        except ImportError:
            print("torchvision is not installed: torchvision_resnet50.npz not written")
        else:
            make_resnet()


if __name__ == "__main__":
    main()
