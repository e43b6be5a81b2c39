"""Pins for the arithmetic that lives in un-vendored third-party code (SURVEY.md 8c "parity unpinned": OpenCV's
estimateAffine*2D / warpAffine / resize / copyMakeBorder, torchvision's ResNet-50 topology).

The fixtures are produced by ``tools/make_cv2_fixture.py`` on a machine that has ``opencv-python`` / ``torchvision``
(this build container has neither and no network).  When a fixture is present, the CPU oracle AND the HIP kernels are
held to it; when it is absent the tests SKIP and say so — they never pass vacuously."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
BORDERS = ("constant", "replicate", "reflect", "wrap", "reflect_101")
# |kernel - cv2| when the wheel is of the float family (see test_kernel_warp_vs_opencv): 255/32 + 2 roundings, and how many bytes
KERNEL_VS_FLOAT_WHEEL_MAX, KERNEL_VS_FLOAT_WHEEL_FRACTION = 10, 0.5
HOW = "run `python tools/make_cv2_fixture.py` where opencv-python / torchvision are installed and commit the file"


def _fixture(name):
    path = os.path.join(G, name)
    if not os.path.isfile(path):
        pytest.skip(f"tests/golden/{name} is absent (no cv2 / torchvision in the build container): parity of this row stays "
                    f"unpinned; {HOW}")
    z = np.load(path)
    versions = {k: str(z[k]) for k in z.files if k.endswith("_version") or k in ("cv2_build_cpu", "cv2_use_optimized", "machine")}
    print(f"tests/golden/{name}: generated with {versions}")                 # shown with `pytest -s` / `-rP`
    return z


def _diff(a, b):
    d = np.abs(a.astype(int) - b.astype(int))
    return int(d.max()), float((d > 0).mean())


def _warp_family(z):
    """Which algorithm family of cv2.warpAffine(INTER_LINEAR) the fixture records, decided on ALL its warps against the two
    oracle variants: "fixed" (classic 5-bit-fraction / 15-bit-weight tables: byte-exact) or "float32" (newer SIMD linear
    kernels).  Returns (family, report lines)."""
    from oracle import align_ref as A
    worst = {"fixed": [0, 0.0], "float32": [0, 0.0]}
    for k, img, mats, dsize in _warp_cases(z):
        for b in BORDERS:
            for j, m in enumerate(mats):
                for variant in worst:
                    mx, frac = _diff(A.warp_affine(img, m, dsize, A.BORDER[b], variant=variant), z[f"warp{k}_{b}"][j])
                    worst[variant] = [max(worst[variant][0], mx), max(worst[variant][1], frac)]
    # (the two families differ by up to several grey levels on noisy images: the fixed-point one quantises the source
    #  coordinate to 1/32 px — so "which one is byte-exact, or within one rounding" decides, not their mutual distance)
    family = "fixed" if worst["fixed"][0] == 0 else ("float32" if worst["float32"][0] <= 1 else "unknown")
    lines = [f"cv2 {z['cv2_version']} warpAffine vs oracle variant {v!r}: max |d| = {w[0]}, worst fraction of differing bytes = {w[1]:.2e}"
             for v, w in worst.items()]
    return family, lines, worst


# ------------------------------------------------------------------------------------------- a13: estimators
def _check_estimates(z, name, allow_skew, fn, tol):
    src, dst = z["est_src"], z["est_dst"]
    mat_cv, ok_cv = z[f"est_{name}_mat"], z[f"est_{name}_ok"]
    mats, oks = fn(src, dst, allow_skew)
    assert np.array_equal(oks != 0, ok_cv != 0), "set of faces OpenCV drops (cropper.py:529-531) differs"
    live = ok_cv != 0
    err = np.abs(mats[live] - mat_cv[live]).max()
    print(f"estimateAffine{'2D' if allow_skew else 'Partial2D'}: max |dM| = {err:.3e} (cv2 {z['cv2_version']})")
    assert err < tol, err


@pytest.mark.parametrize("name,allow_skew,tol", [("partial", False, 1e-5), ("affine", True, 1e-3)])
def test_oracle_estimators_vs_opencv(name, allow_skew, tol):
    """Tolerances: OpenCV runs RANSAC's first sample + 10 LM iterations in float64 on un-centred coordinates; DESIGN.md
    section 4 bounds its distance from the closed form at 4.7e-7 px (similarity) / 3.9e-5 px (affine)."""
    from oracle import align_ref as A
    z = _fixture("opencv_align.npz")

    def run(src, dst, skew):
        mats, oks = np.zeros((len(src), 2, 3)), np.zeros(len(src), np.int32)
        for i, s in enumerate(src):
            m = A.estimate_transform(s, dst, skew)
            if m is not None:
                mats[i], oks[i] = m, 1
        return mats, oks
    _check_estimates(z, name, allow_skew, run, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("name,allow_skew,tol", [("partial", False, 1e-5), ("affine", True, 1e-3)])
def test_kernel_estimators_vs_opencv(name, allow_skew, tol, device):
    from face_crop_plus_amd import align
    z = _fixture("opencv_align.npz")

    def run(src, dst, skew):
        mat, ok = align.estimate_transform(torch.from_numpy(src).to(device), torch.from_numpy(dst).to(device), skew)
        return mat.cpu().numpy().reshape(-1, 2, 3), ok.cpu().numpy()
    _check_estimates(z, name, allow_skew, run, tol)


# ------------------------------------------------------------------------------------------- a14: warpAffine
def _warp_cases(z):
    for k in range(int(z["warp_cases"])):
        yield k, z[f"warp{k}_img"], z[f"warp{k}_mat"], tuple(int(v) for v in z[f"warp{k}_dsize"])


def test_oracle_warp_vs_opencv():
    """The classic fixed-point bilinear warp (AB_BITS 10, INTER_BITS 5, (sum + 2^14) >> 15) has no platform freedom: a wheel
    that runs it must match ``warp_affine`` byte for byte.  A wheel of the float family (newer SIMD linear kernels; the
    reference leaves opencv-python unpinned) is recognised as such — reported with its version, CPU dispatch and the size of
    the difference — and must then lie within one grey level of the fixed-point result; anything else is a defect."""
    from oracle import align_ref as A
    z = _fixture("opencv_align.npz")
    family, lines, worst = _warp_family(z)
    print("\n".join(lines))
    print(f"=> this fixture records the {family!r} family")
    # "fixed": byte-exact by construction of _warp_family; "float32": within one rounding of the float restatement (which is
    # written from memory of the published source: an exact match is not claimed); anything else is a defect to look at
    assert family in ("fixed", "float32"), f"neither family explains this wheel's warpAffine: {lines}"
    # the fixture says which family it was generated as (tools/make_cv2_fixture.py records it): a regenerated fixture of the
    # other family, or an oracle change that re-classifies the committed one, fails loudly instead of switching tolerance
    if "warp_family" in z.files:
        assert str(z["warp_family"]) == family, f"fixture recorded as {z['warp_family']!r}, classified as {family!r}"
    # the wheel's own portable path (optimised code switched off), when recorded, is the classic algorithm
    if "warp0_constant_noopt" in z.files:
        for k, img, mats, dsize in _warp_cases(z):
            for b in BORDERS:
                for j, m in enumerate(mats):
                    mx, frac = _diff(A.warp_affine(img, m, dsize, A.BORDER[b]), z[f"warp{k}_{b}_noopt"][j])
                    assert mx == 0, f"setUseOptimized(False) warp, case {k}, border {b}, matrix {j}: max |d| {mx}, {frac:.2e} of bytes"


@pytest.mark.gpu
def test_kernel_warp_vs_opencv(device):
    from face_crop_plus_amd import align
    z = _fixture("opencv_align.npz")
    for k, img, mats, dsize in _warp_cases(z):
        dimg = torch.from_numpy(img)[None].to(device)
        idx = torch.zeros(len(mats), dtype=torch.int32, device=device)
        dm = torch.from_numpy(mats.reshape(-1, 6)).to(device)
        for b in BORDERS:
            got = align.warp_affine(dimg, idx, dm, None, None, dsize, align.border_code(b)).cpu().numpy()
            mx, frac = _diff(got, z[f"warp{k}_{b}"])
            if mx:                       # say HOW it differs: a float-family wheel is a known other algorithm, a bug is not
                from oracle import align_ref as A
                family, lines, _ = _warp_family(z)
                print(f"case {k}, border {b}: kernel vs cv2 max |d| = {mx}, {frac:.2e} of bytes; fixture family {family!r}; " + "; ".join(lines))
                assert family == "float32", f"case {k}, border {b}: kernel differs from a fixed-point wheel"
                # the kernel implements the classic algorithm: its distance from a float-family wheel must be exactly the
                # distance between the two algorithms, i.e. the kernel still equals the fixed-point restatement byte for byte
                fixed = np.stack([A.warp_affine(img, m, dsize, A.BORDER[b]) for m in mats])
                assert np.array_equal(got, fixed), f"case {k}, border {b}"
                # ... and that distance is BOUNDED against the wheel itself, not only explained: the fixed-point family
                # quantises the source coordinate to 1/32 px, so on an image of grey-level range R neighbouring pixels differ by
                # <= R and the two bilinear results by <= R/32 + 1 rounding each way (KERNEL_VS_FLOAT_WHEEL_MAX for uint8 noise);
                # a drift of the kernel beyond it fails here even though no fixed-point wheel is at hand
                assert mx <= KERNEL_VS_FLOAT_WHEEL_MAX and frac <= KERNEL_VS_FLOAT_WHEEL_FRACTION, \
                    f"case {k}, border {b}: kernel vs float-family cv2: max |d| {mx}, {frac:.2e} of bytes"
            if f"warp{k}_{b}_noopt" in z.files:
                assert np.array_equal(got, z[f"warp{k}_{b}_noopt"]), f"case {k}, border {b}: portable-path wheel output"


# ------------------------------------------------------------------------------------------- f1: resize + border
def test_oracle_batch_builder_vs_opencv():
    """cv2.resize INTER_AREA / INTER_CUBIC on uint8 + copyMakeBorder.  INTER_AREA is exact; OpenCV's SIMD build runs the
    vertical cubic pass in float32, which may differ from the fixed-point definition by one LSB on ~1e-5 of the pixels
    (DESIGN.md section 4): <= 1 LSB on < 1e-4 of the bytes is accepted for cubic, and reported."""
    from oracle import batch_ref as B
    z = _fixture("opencv_batch.npz")
    for k in range(int(z["batch_cases"])):
        img, size = z[f"batch{k}_img"], int(z[f"batch{k}_size"])
        ww, hh, pad, _, interp = B.geometry(img.shape[0], img.shape[1], size)
        assert list(pad) == z[f"batch{k}_pad"].tolist()
        res = B.resize_u8(img, ww, hh, interp)
        d = np.abs(res.astype(int) - z[f"batch{k}_resized"].astype(int))
        print(f"case {k} ({interp}): max diff {d.max()}, differing bytes {(d > 0).mean():.2e}")
        assert d.max() == 0 if interp == "area" else (d.max() <= 1 and (d > 0).mean() < 1e-4), (k, interp)
        for b in BORDERS:
            got = B.copy_make_border(z[f"batch{k}_resized"], *pad, mode=b)
            assert np.array_equal(got, z[f"batch{k}_{b}"]), (k, b)


@pytest.mark.gpu
def test_kernel_batch_builder_vs_opencv(device):
    from face_crop_plus_amd.batch import build_batch
    z = _fixture("opencv_batch.npz")
    for k in range(int(z["batch_cases"])):
        img, size = z[f"batch{k}_img"], int(z[f"batch{k}_size"])
        for b in BORDERS:
            got = build_batch([img], size, b, device)[0][0].cpu().numpy()
            d = np.abs(got.astype(int) - z[f"batch{k}_{b}"].astype(int))
            assert d.max() <= 1 and (d > 0).mean() < 1e-4, (k, b, int(d.max()), float((d > 0).mean()))


# ------------------------------------------------------------------------------------------- a3: ResNet-50 body
def test_oracle_body_vs_torchvision():
    """The oracle's ResNet-50 v1.5 body (stride on the 3x3, IntermediateLayerGetter order) against torchvision's own
    forward on the build's generated weights: same ops in the same order, so fp32 results agree to summation noise."""
    from face_crop_plus_amd import weights
    from oracle import retinaface_ref as R
    z = _fixture("torchvision_resnet50.npz")
    sd = weights.generate_state_dict("retinaface")
    with torch.no_grad():
        feats = R.body(torch.from_numpy(z["x"]), sd)
    for k, f in zip((1, 2, 3), feats):
        ref = z[f"feat{k}"]
        assert tuple(f.shape) == ref.shape
        err = float(np.abs(f.numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
        print(f"layer{k + 1}: relative error {err:.2e} (torchvision {z['torchvision_version']})")
        assert err < 1e-5


@pytest.mark.gpu
def test_kernel_body_vs_torchvision(device):
    from face_crop_plus_amd import weights, engine as E
    from face_crop_plus_amd.retinaface import RetinaFace
    z = _fixture("torchvision_resnet50.npz")
    det = RetinaFace("all", 0.6).load(device, weights.generate_state_dict("retinaface"))
    x = torch.from_numpy(z["x"]).to(device)
    # the detector's stem expects RGB - mean in NHWC4; torchvision's body saw `x` as is, with the BGR swap folded into the
    # stem filter here: feed x's channels reversed so that the filter's permutation restores the fixture's order
    x4 = E.f32nchw_to_nhwc4(x.flip(1).contiguous())
    det._debug_feats = feats = []
    det.forward_heads(x4)
    torch.cuda.synchronize()
    assert len(feats) == 3
    for k, f in zip((1, 2, 3), feats):
        ref = z[f"feat{k}"]
        err = float(np.abs(f.nchw().cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
        assert err < 2e-5, (k, err)
