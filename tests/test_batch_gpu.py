"""GPU: the HIP batch builder (fcp_build_batch_u8) against the oracle restatement of
cv2.resize + cv2.copyMakeBorder (utils.py:316-335) — bit-exact."""
import numpy as np
import pytest
import torch

from oracle import batch_ref as B

pytestmark = pytest.mark.gpu


def _imgs(shapes, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in shapes]


RAGGED = [(50, 80), (90, 40), (64, 64), (20, 30), (128, 128), (192, 192), (256, 128), (61, 61), (200, 37),
          (3, 200), (64, 10), (1, 1), (63, 64), (65, 64), (130, 64)]


@pytest.mark.parametrize("size", [(64, 64), (96, 48), (40, 72)])
def test_ragged_batch_matches_oracle(size, device):
    from face_crop_plus_amd.batch import build_batch
    imgs = [im for im in _imgs(RAGGED, 1) if min(B.geometry(im.shape[0], im.shape[1], size)[:2]) >= 1]
    got, unscales, pads = build_batch(imgs, size, "constant", device)
    exp, eu, ep = B.as_batch(imgs, size)
    assert got.shape == exp.shape and got.dtype == torch.uint8 and got.is_cuda
    assert np.array_equal(pads, ep) and np.array_equal(unscales, eu)
    got = got.cpu().numpy()
    for i in range(len(imgs)):
        assert np.array_equal(got[i], exp[i]), (i, imgs[i].shape, np.abs(got[i].astype(int) - exp[i]).max())


@pytest.mark.parametrize("mode", ["replicate", "reflect", "wrap", "reflect_101"])
def test_border_modes(mode, device):
    from face_crop_plus_amd import utils
    imgs = _imgs([(50, 80), (90, 40), (5, 64), (64, 3), (20, 30)], 2)
    got, _, _ = utils.as_batch(imgs, (64, 64), mode, device)          # reference signature: numpy out
    exp, _, _ = B.as_batch(imgs, (64, 64), mode)
    assert isinstance(got, np.ndarray) and np.array_equal(got, exp)


def test_interpolation_paths_individually(device):
    """copy / cubic up / cubic down / area 2x2 / area 3x3 / area 4x2 / general area, each checked alone."""
    from face_crop_plus_amd.batch import build_batch
    cases = {"copy": ((64, 48), (48, 64)), "cubic_up": ((20, 15), (48, 64)), "cubic_down": ((60, 45), (36, 48)),
             "area2": ((96, 128), (64, 48)), "area3": ((144, 192), (64, 48)), "area_gen": ((135, 240), (64, 36))}
    for name, (shape, size) in cases.items():
        img = _imgs([shape], 3)[0]
        ww, hh, pad, _, interp = B.geometry(shape[0], shape[1], size)
        got = build_batch([img], size, "constant", device)[0].cpu().numpy()[0]
        exp = B.as_batch([img], size)[0][0]
        assert np.array_equal(got, exp), name
    # anisotropic integral scales (4 in x, 2 in y) through the C ABI directly
    from face_crop_plus_amd import _native as N
    from face_crop_plus_amd.batch import ITEM_DTYPE
    img = _imgs([(24, 36)], 4)[0]
    items = np.zeros(1, ITEM_DTYPE)
    items[0] = (0, 24, 36, 12, 9, 0, 0, 1, 0)
    out = torch.empty((1, 12, 9, 3), dtype=torch.uint8, device=device)
    blob, items_dev = torch.from_numpy(img.reshape(-1)).to(device), torch.from_numpy(items.view(np.uint8)).to(device)
    N.check(N.lib().fcp_build_batch_u8(N.ptr(blob), img.size, items.ctypes.data, N.ptr(items_dev), 1, 12, 9, 0,
                                       N.ptr(out), N.stream_ptr()))
    assert np.array_equal(out.cpu().numpy()[0], B.resize_area_u8(img, 9, 12))


def test_4k_frames_into_1024(device):
    """3840x2160 frames -> 1024x576 + 224 px top/bottom (scale 3.75, general INTER_AREA)."""
    from face_crop_plus_amd.batch import build_batch
    imgs = _imgs([(2160, 3840), (2160, 3840)], 5)
    imgs[1][:] = (np.arange(3840)[None, :, None] // 15).astype(np.uint8)       # smooth ramp
    got, _, pads = build_batch(imgs, 1024, "constant", device)
    assert pads.tolist() == [[224, 224, 0, 0]] * 2
    exp, _, _ = B.as_batch(imgs, 1024)
    assert np.array_equal(got.cpu().numpy(), exp)


def test_bad_items_are_rejected(device):
    from face_crop_plus_amd import _native as N
    from face_crop_plus_amd.batch import ITEM_DTYPE, build_batch
    items = np.zeros(1, ITEM_DTYPE)
    items[0] = (0, 8, 8, 16, 16, 0, 0, 1, 0)                            # INTER_AREA asked to enlarge
    out = torch.empty((1, 16, 16, 3), dtype=torch.uint8, device=device)
    blob, items_dev = torch.zeros(192, dtype=torch.uint8, device=device), torch.from_numpy(items.view(np.uint8)).to(device)
    with pytest.raises(RuntimeError, match="decimation"):
        N.check(N.lib().fcp_build_batch_u8(N.ptr(blob), 192, items.ctypes.data, N.ptr(items_dev), 1, 16, 16, 0,
                                           N.ptr(out), N.stream_ptr()))
    items[0] = (64, 8, 8, 16, 16, 0, 0, 0, 0)                           # runs past the blob
    with pytest.raises(RuntimeError, match="outside the source blob"):
        N.check(N.lib().fcp_build_batch_u8(N.ptr(blob), 192, items.ctypes.data, N.ptr(items_dev), 1, 16, 16, 0,
                                           N.ptr(out), N.stream_ptr()))
    with pytest.raises(ValueError):
        build_batch([np.zeros((4, 4), np.uint8)], 16, "constant", device)
    empty, u, p = build_batch([], 16, "constant", device)
    assert empty.shape == (0, 16, 16, 3) and p.shape == (0, 4)


def test_build_batch_direct_upload_from_registered_rings(tmp_path, device):
    """process_dir's decode workers hand over images inside page-locked shared-memory rings; build_batch uploads those
    straight from the ring.  All-pinned, mixed and all-staged batches must give the same bytes."""
    from PIL import Image
    from face_crop_plus_amd import Cropper
    from face_crop_plus_amd._io_pool import IOProcesses
    from face_crop_plus_amd.batch import build_batch
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (int(rng.integers(90, 300)), int(rng.integers(90, 300)), 3), dtype=np.uint8) for _ in range(7)]
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(tmp_path / f"{i}.png")
    pool = IOProcesses(1, 1, ring_mb=4, register=Cropper._pin_ring)
    try:
        got = [pool.read(str(tmp_path / f"{i}.png")) for i in range(7)]
        assert all(np.array_equal(a, im) for (a, _), im in zip(got, imgs))
        flags = pool.pinned_flags([t for _, t in got])
        assert all(flags), "the ring was not registered as page-locked memory"
        want = build_batch(imgs, (256, 192), "constant", device)[0]
        ring_imgs = [a for a, _ in got]
        assert torch.equal(build_batch(ring_imgs, (256, 192), "constant", device, flags)[0], want)
        mixed = [a if i % 2 else a.copy() for i, a in enumerate(ring_imgs)]
        assert torch.equal(build_batch(mixed, (256, 192), "constant", device, [bool(i % 2) for i in range(7)])[0], want)
        torch.cuda.synchronize()
        pool.release([t for _, t in got])
    finally:
        pool.close()


def test_area_integral_ratios_match_scikit_image_block_means(device):
    """The batch-builder kernel on the scikit-image block-mean fixture (tests/test_batch_oracle.py): INTER_AREA at integral
    ratios is the rounded block mean; the kernel equals the oracle byte for byte there as everywhere."""
    import os
    from face_crop_plus_amd.batch import build_batch
    from oracle import batch_ref as B
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_area.npz"))
    for k in range(int(z["cases"])):
        img, f = z[f"img{k}"], int(z[f"factor{k}"])
        h, w = img.shape[0] // f, img.shape[1] // f
        got = build_batch([img], (w, h), "constant", device)[0].cpu().numpy()[0]      # exact fit: no padding
        assert np.array_equal(got, B.resize_area_u8(img, w, h))
        assert np.abs(got.astype(np.float64) - z[f"mean_x64_{k}"].astype(np.float64) / 64).max() <= 0.5 + 1 / 64
