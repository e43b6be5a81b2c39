"""End-to-end Cropper.process_dir on generated image files (config 0: plumbing on the GPU path)."""
import os

import numpy as np
import pytest
import torch

from oracle import retinaface_ref as R, align_ref as A, batch_ref as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def image_dir(tmp_path_factory):
    from PIL import Image
    d = tmp_path_factory.mktemp("imgs")
    rng = np.random.default_rng(7)
    for i, (h, w) in enumerate([(160, 160), (160, 160), (120, 160), (200, 100), (160, 160)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(d / f"{i:06d}.png")
    (d / "broken.png").write_bytes(b"nope")
    return str(d)


def test_process_dir_matches_oracle(image_dir, tmp_path, device):
    from PIL import Image
    from face_crop_plus_amd import Cropper, weights, utils
    sd = weights.generate_state_dict("retinaface")
    out = tmp_path / "faces"
    c = Cropper(output_size=64, resize_size=160, strategy="largest", det_threshold=0.6, batch_size=3,
                output_format="png", device="cuda:0", weights={"retinaface": sd})
    with pytest.warns(UserWarning):
        c.process_dir(image_dir, str(out), desc=None)
    written = sorted(os.listdir(out))
    assert written and all(f.endswith(".png") for f in written)
    # oracle: same batch building, CPU detector + OpenCV-restated crop
    names = sorted(f for f in os.listdir(image_dir) if f != "broken.png")
    imgs, _ = utils.read_images(names, image_dir)
    batch, _, pads = B.as_batch(imgs, (160, 160))           # oracle cv2.resize + pad
    lm, idx = R.predict(torch.from_numpy(batch).permute(0, 3, 1, 2).float(), sd, "largest", 0.6)
    assert written == [names[i] for i in idx]
    # landmarks: the detector on the same batch against the oracle, north_star's 1e-3 px; crops: the written files are
    # byte-equal to the oracle's estimate + warp of the GPU's own landmarks (a landmark difference of 1e-4 px flips isolated
    # fixed-point roundings between the two landmark sets; this comparison needs no byte budget)
    lm_g, idx_g = c.det_model.predict(torch.from_numpy(batch).to(c.device))
    assert list(idx_g) == list(idx) and np.abs(lm_g - lm).max() < 1e-3
    un = lm_g - pads[idx][:, None, [2, 0]].astype(np.float32)
    crops = A.crop_align(batch, pads, idx, un, A.landmarks_target((64, 64), 0.65), (64, 64), "constant")
    for k, i in enumerate(idx):
        got = np.asarray(Image.open(out / names[i]).convert("RGB"))
        assert np.array_equal(got, crops[k]), names[i]


def test_process_dir_all_strategy_groups_and_masks(image_dir, tmp_path, device):
    from face_crop_plus_amd import Cropper
    out = tmp_path / "grouped"
    c = Cropper(output_size=64, resize_size=160, strategy="all", det_threshold=0.6, batch_size=4,
                attr_groups={"hair": [17], "no_hat": [-14]}, mask_groups={"hair": [17]},
                device="cuda:0", weights={k: "generated" for k in ("retinaface", "bisenet")})
    with pytest.warns(UserWarning):
        c.process_dir(image_dir, str(out), desc=None)
    tree = {os.path.relpath(os.path.join(dp, f), out) for dp, _, fs in os.walk(out) for f in fs}
    assert tree, "no outputs written"
    assert all(p.split(os.sep)[0] in ("hair", "no_hat") for p in tree)
    assert any(p.split(os.sep)[1] == "hair_mask" for p in tree)          # attr / mask cross product (cropper.py:731-746)
    assert all(os.path.splitext(p)[0].rsplit("_", 1)[1].isdigit() for p in tree)   # "_N" suffix for strategy "all"


def test_given_landmarks_and_enhancement(image_dir, tmp_path, device):
    """Pre-computed landmark path (no detector, cropper.py:796-813) WITH RRDB enhancement switched on (cropper.py:833-836):
    the gate of rrdb.py:124-140 measures each image's faces against the area of images[0]; the threshold sits between
    the two face factors, so exactly one of the two images is enhanced.  Checked against the oracle: the gate decision,
    the enhanced image (rounding-boundary flips only) and the written crops (byte-equal to the oracle's estimate + warp of
    the enhanced / untouched image)."""
    from PIL import Image
    from face_crop_plus_amd import Cropper, weights
    from oracle import rrdb_ref as RR
    tgt = A.landmarks_target((48, 48), 0.65)
    lms = np.stack([tgt * 1.5 + 10, tgt * 1.2 + 20]).astype(np.float32)
    names = np.array(["000000.png", "000003.png"])
    sd = weights.generate_state_dict("rrdb")
    # face factors as the reference computes them: (x4-x0)(y4-y0) / (H0*W0) with images[0] = 000000.png (160x160)
    ff = [float((l[4, 0] - l[0, 0]) * (l[4, 1] - l[0, 1]) / (160 * 160)) for l in lms]
    assert ff[1] < ff[0]
    thr = 0.5 * (ff[0] + ff[1])
    out = tmp_path / "given"
    c = Cropper(output_size=48, landmarks=(lms, names), det_threshold=None, enh_threshold=thr, padding="reflect",
                device="cuda:0", weights={"rrdb": sd})
    assert c.enh_model is not None
    c.process_dir(image_dir, str(out), desc=None)
    assert sorted(os.listdir(out)) == ["000000.png", "000003.png"]
    files = sorted(f for f in os.listdir(image_dir) if f != "broken.png")
    imgs = [np.asarray(Image.open(os.path.join(image_dir, f)).convert("RGB")) for f in files]
    indices = [0, 3]                                             # batch order: the two files that have a landmark row
    assert RR.gate(lms, indices, len(imgs), 160, 160, thr) == [False, False, False, True, False]
    assert c.enh_model.gate(len(imgs), 160, 160, lms, indices) == [3]
    # image 0: below the gate -> the plain warp of the decoded file
    exp0 = A.warp_affine(imgs[0], A.estimate_transform(lms[0], tgt), (48, 48), A.BORDER["reflect"])
    assert np.array_equal(np.asarray(Image.open(out / "000000.png").convert("RGB")), exp0)
    # image 3 (200x100): enhanced.  GPU enhancer vs the oracle's predict on the same pixels ...
    x = torch.from_numpy(imgs[3])[None]
    ref = RR.predict(x.permute(0, 3, 1, 2).float(), sd, None, None).permute(0, 2, 3, 1).numpy()[0]
    enh = c.enh_model.predict(x.to(c.device), None, None)[0].cpu().numpy()
    diff = np.abs(enh.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3 and not np.array_equal(enh, imgs[3])
    # ... and the written crop is the oracle's warp of that enhanced image, byte for byte
    exp3 = A.warp_affine(enh, A.estimate_transform(lms[1], tgt), (48, 48), A.BORDER["reflect"])
    got3 = np.asarray(Image.open(out / "000003.png").convert("RGB"))
    assert np.array_equal(got3, exp3)
    plain3 = A.warp_affine(imgs[3], A.estimate_transform(lms[1], tgt), (48, 48), A.BORDER["reflect"])
    assert not np.array_equal(got3, plain3)                        # the enhancement really reached the file


def test_crop_align_numpy_signature(device):
    from face_crop_plus_amd import Cropper
    c = Cropper(output_size=32, det_threshold=None, device="cuda:0")
    rng = np.random.default_rng(2)
    imgs = rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    lm = np.stack([c.landmarks_target * 1.5 + 4, np.ones((5, 2), np.float32), c.landmarks_target + 9]).astype(np.float32)
    out = c.crop_align(imgs, np.array([[0, 0, 0, 0], [2, 2, 0, 0]]), [0, 1, 1], lm)
    assert out.shape == (2, 32, 32, 3) and out.dtype == np.uint8          # degenerate face dropped
    ragged = c.crop_align([imgs[0], imgs[1][:50]], None, [0, 1], lm[[0, 2]])
    assert ragged.shape == (2, 32, 32, 3)
    assert np.array_equal(ragged[0], out[0])


def test_nonsquare_resize_all_strategy_with_parse(tmp_path, device):
    """configs[4]-style shapes in miniature: wide images letter-boxed into a non-square batch
    (top/bottom padding, utils.py:322-326), strategy "all", parsing on; crops equal a per-face warp."""
    from PIL import Image
    from face_crop_plus_amd import Cropper, weights, utils
    from oracle import retinaface_ref as R, align_ref as A, batch_ref as B
    src = tmp_path / "wide"
    src.mkdir()
    rng = np.random.default_rng(11)
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (135, 240, 3), dtype=np.uint8)).save(src / f"w{i}.png")
    sd = weights.generate_state_dict("retinaface")
    out = tmp_path / "out"
    c = Cropper(output_size=(48, 64), resize_size=(192, 128), strategy="all", det_threshold=0.55, batch_size=3,
                mask_groups={"hair": [17]}, device="cuda:0", weights={"retinaface": sd, "bisenet": "generated"})
    c.process_dir(str(src), str(out), desc=None)
    names = sorted(os.listdir(src))
    imgs, _ = utils.read_images(names, str(src))
    batch, _, pads = B.as_batch(imgs, (192, 128))
    assert batch.shape == (3, 128, 192, 3) and pads[0].tolist() == [10, 10, 0, 0]
    lm, idx = R.predict(torch.from_numpy(batch).permute(0, 3, 1, 2).float(), sd, "all", 0.55)
    assert len(idx) > 3                                             # several faces per image
    lm_g, idx_g = c.det_model.predict(torch.from_numpy(batch).to(c.device))        # the GPU's landmarks on the same batch
    assert list(idx_g) == list(idx) and np.abs(lm_g - lm).max() < 1e-3            # north_star's tolerance vs the oracle
    lm = lm_g - pads[idx][:, None, [2, 0]].astype(np.float32)
    crops = A.crop_align(batch, pads, idx, lm, A.landmarks_target((48, 64), 0.65), (48, 64), "constant")
    assert crops.shape[1:] == (64, 48, 3)                           # (h, w) from output_size=(w, h)
    written = sorted(os.listdir(out / "hair"))
    per_file = {n: sum(1 for i in idx if names[i] == n) for n in names}
    exp_names = sorted(f"{os.path.splitext(n)[0]}_{k}.png" for n in names for k in range(per_file[n]))
    assert set(written) <= set(exp_names) and len(written) > 0
    k_of = {}
    for k, i in enumerate(idx):
        k_of.setdefault(i, []).append(k)
    for fn in written:
        stem, j = fn[:-4].rsplit("_", 1)
        face = k_of[names.index(stem + ".png")][int(j)]
        got = np.asarray(Image.open(out / "hair" / fn).convert("RGB"))
        assert got.shape == (64, 48, 3) and np.array_equal(got, crops[face])      # the oracle's warp of the GPU's landmarks, byte for byte
    assert sorted(os.listdir(out / "hair_mask")) == written


def test_process_dir_pipeline_is_deterministic(tmp_path, device):
    """Overlapped decode / device / encode pipeline: the output set does not depend on the number of GPU
    workers or I/O threads, and equals what the synchronous process_batch writes."""
    from PIL import Image
    from face_crop_plus_amd import Cropper
    src = tmp_path / "many"
    src.mkdir()
    rng = np.random.default_rng(5)
    for i in range(13):
        h, w = int(rng.integers(90, 200)), int(rng.integers(90, 200))
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(src / f"im{i:02d}.png")
    (src / "zz_broken.png").write_bytes(b"nope")
    kw = dict(output_size=64, resize_size=160, strategy="all", det_threshold=0.55, batch_size=3, device="cuda:0",
              weights={"retinaface": "generated"})
    outs = []
    for k, (np_, io, gw) in enumerate(((1, 2, 1), (3, 5, None), (1, 3, None))):     # one worker; three; the default (two for num_processes=1)
        c = Cropper(num_processes=np_, **kw)
        c.io_threads = io
        if gw is not None:
            c.gpu_workers = gw
        with pytest.warns(UserWarning, match="Could not read"):
            c.process_dir(str(src), str(tmp_path / f"o{k}"), desc=None)
        outs.append({f: (tmp_path / f"o{k}" / f).read_bytes() for f in sorted(os.listdir(tmp_path / f"o{k}"))})
    c = Cropper(**kw)
    files = sorted(os.listdir(src))
    with pytest.warns(UserWarning):
        for i in range(0, len(files), 3):
            c.process_batch(files[i:i + 3], str(src), str(tmp_path / "sync"))
    outs.append({f: (tmp_path / "sync" / f).read_bytes() for f in sorted(os.listdir(tmp_path / "sync"))})
    assert len(outs[0]) > 5 and outs[0] == outs[1] == outs[2] == outs[3]


def test_crop_align_plumbing_like_reference(device):
    """The semantics recorded from the reference's crop_align with a stand-in cv2 (tests/golden/plumbing.json):
    output (F, h, w, 3) for output_size = (w, h); the padded band of a batch image is never sampled
    (image[t:h-b, l:w-r] is the warp source); reflect border; empty input -> empty array."""
    from face_crop_plus_amd import Cropper
    c = Cropper(output_size=(96, 112), padding="reflect", det_threshold=None, device="cuda:0")
    images = np.stack([np.full((40, 60, 3), v, np.uint8) for v in (10, 20, 30)])
    paddings = np.array([[0, 0, 0, 0], [3, 4, 0, 0], [0, 0, 5, 6]])
    images[1, :3], images[1, 36:], images[2, :, :5], images[2, :, 54:] = 255, 255, 255, 255      # paint the bands
    tgt = c.landmarks_target
    lms = np.stack([tgt * 0.3 + 5, tgt * 0.25 + 2, tgt * 0.2 + 1, tgt * 0.3 + 3, tgt * 0.3 + 20]).astype(np.float32)
    out = c.crop_align(images, paddings, [0, 1, 1, 2, 2], lms)
    assert out.shape == (5, 112, 96, 3) and out.dtype == np.uint8
    assert [int(np.unique(o)[0]) for o in out] == [10, 20, 20, 30, 30] and all(len(np.unique(o)) == 1 for o in out)
    assert c.crop_align(images, paddings, [], lms[:0]).shape == (0,)
    lms[1] = 7.0                                                    # all five points coincide: no transform -> face dropped
    assert c.crop_align(images, paddings, [0, 1, 1], lms[:3]).shape == (2, 112, 96, 3)


# (width, height) of the reference's eight demo/input_images
DEMO_GEOMETRIES = [(1024, 624), (409, 687), (423, 594), (500, 281), (1648, 2464), (610, 826), (410, 594), (334, 500)]


def test_c1_demo_geometries_default_config(tmp_path, device):
    """BASELINE configs[0] shape: Cropper(strategy='largest', det_threshold=0.6) with every other argument at its
    default (resize_size 1024, output_size 256, batch_size 8) on eight images with the demo set's geometries
    (up- and down-scaled, portrait and landscape, INTER_CUBIC and INTER_AREA legs of as_batch), against the oracle
    chain batch_ref -> retinaface_ref -> align_ref.  The real photos / pretrained weights are not available
    here, so the pixels are generated and the weights are the seeded generator's.

    Tight form of the end-to-end check: the batch is byte-exact, faces / order identical, landmarks within
    1e-3 px of the oracle's, and the written crops are BYTE-EXACT to the oracle's estimate + warp applied to
    the landmarks the GPU produced (the 1e-4 px landmark noise is the only thing that could flip a byte)."""
    from PIL import Image
    from face_crop_plus_amd import Cropper, weights, utils
    from face_crop_plus_amd.batch import build_batch
    src = tmp_path / "demo"
    src.mkdir()
    rng = np.random.default_rng(2024)
    for i, (w, h) in enumerate(DEMO_GEOMETRIES):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        im = np.stack([128 + 100 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (7 + 5 * c)) for c in range(3)], -1)
        im = np.clip(im + rng.normal(0, 25, im.shape), 0, 255).astype(np.uint8)
        Image.fromarray(im).save(src / f"demo{i}.png")
    sd = weights.generate_state_dict("retinaface")
    c = Cropper(strategy="largest", det_threshold=0.6, device="cuda:0", weights={"retinaface": sd})
    assert (c.output_size, c.resize_size, c.batch_size, c.face_factor) == ((256, 256), (1024, 1024), 8, 0.65)
    out = tmp_path / "out"
    c.process_dir(str(src), str(out), desc=None)
    names = sorted(os.listdir(src))
    imgs, _ = utils.read_images(names, str(src))
    batch, _, pads = B.as_batch(imgs, (1024, 1024))
    dev_batch, _, dpads = build_batch(imgs, c.resize_size, "constant", c.device)
    assert dpads.tolist() == pads.tolist() and np.array_equal(dev_batch.cpu().numpy(), batch)
    lm_ref, idx_ref = R.predict(torch.from_numpy(batch).permute(0, 3, 1, 2).float(), sd, "largest", 0.6)
    lm, idx = c.det_model.predict(dev_batch)
    assert list(idx) == list(idx_ref) and len(idx) >= 6
    err = float(np.abs(lm - lm_ref).max())
    print("C1 landmark error vs oracle:", err)
    assert err < 1e-3
    assert sorted(os.listdir(out)) == [names[i] for i in idx]
    un = lm - pads[idx][:, None, [2, 0]].astype(np.float32)
    crops = A.crop_align(batch, pads, list(idx), un, A.landmarks_target((256, 256), 0.65), (256, 256), "constant")
    for k, i in enumerate(idx):
        got = np.asarray(Image.open(out / names[i]).convert("RGB"))
        assert np.array_equal(got, crops[k]), f"{names[i]}: crop differs from oracle(estimate + warp) of the same landmarks"


def test_two_croppers_two_threads_one_device(tmp_path, device):
    """Native state is per device and thread safe: two Croppers built and run concurrently from two host threads
    on cuda:0 (own HIP stream each; shared autotune cache and per-kernel LDS opt-in) write exactly what one
    Cropper writes alone."""
    import threading
    from PIL import Image
    from face_crop_plus_amd import Cropper
    from face_crop_plus_amd import engine as E
    src = tmp_path / "src"
    src.mkdir()
    rng = np.random.default_rng(13)
    for i in range(8):
        Image.fromarray(rng.integers(0, 256, (150, 170, 3), dtype=np.uint8)).save(src / f"t{i}.png")
    kw = dict(output_size=64, resize_size=(192, 160), strategy="all", det_threshold=0.55, batch_size=4,
              device="cuda:0", weights={"retinaface": "generated"})
    E.Autotune.cache.clear()                       # both threads meet untuned shapes
    errors, outs = [], {}

    def run(tag):
        try:
            with torch.cuda.device(0), torch.cuda.stream(torch.cuda.Stream()):
                c = Cropper(**kw)
                c.process_dir(str(src), str(tmp_path / tag), desc=None)
                torch.cuda.current_stream().synchronize()
            outs[tag] = {f: (tmp_path / tag / f).read_bytes() for f in sorted(os.listdir(tmp_path / tag))}
        except Exception as e:                      # surfaced below: a thread's exception is otherwise lost
            errors.append((tag, repr(e)))

    threads = [threading.Thread(target=run, args=(t,)) for t in ("a", "b")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    run("solo")
    assert not errors, errors
    assert len(outs["solo"]) > 8 and outs["a"] == outs["b"] == outs["solo"]


def test_process_dir_without_detector_copies_borrowed_images(tmp_path, device):
    """det_threshold=None and no landmarks: the decoded images themselves are written.  With decode worker processes they are
    views of recycled shared-memory rings while the encoders run asynchronously: the output must still equal the thread
    pipeline's, file for file (a tiny ring forces recycling inside the run)."""
    from PIL import Image
    from face_crop_plus_amd import Cropper
    src = tmp_path / "src"
    src.mkdir()
    rng = np.random.default_rng(21)
    for i in range(24):
        Image.fromarray(rng.integers(0, 256, (120, 150, 3), dtype=np.uint8)).save(src / f"p{i:02d}.png")
    outs = {}
    for tag, procs in (("threads", (0, 0)), ("procs", (2, 1))):
        c = Cropper(det_threshold=None, batch_size=4, device="cuda:0", output_format="png")
        c.io_processes, c.io_ring_mb = procs, 1
        c.process_dir(str(src), str(tmp_path / tag), desc=None)
        outs[tag] = {f: (tmp_path / tag / f).read_bytes() for f in sorted(os.listdir(tmp_path / tag))}
    assert len(outs["threads"]) == 24 and outs["threads"] == outs["procs"]
    for f, b in outs["procs"].items():                          # and they ARE the inputs (PNG is lossless)
        assert np.array_equal(np.asarray(Image.open(tmp_path / "procs" / f)), np.asarray(Image.open(src / f)))


def test_process_dir_recovers_after_a_decode_worker_dies(image_dir, tmp_path, device):
    """A run whose I/O worker process dies fails loudly — and the NEXT run on the same Cropper starts fresh workers instead of
    getting the dead process (or ring regions the aborted run never gave back) again; its output is the synchronous run's."""
    import warnings
    from face_crop_plus_amd import Cropper, weights
    sd = weights.generate_state_dict("retinaface")
    mk = lambda: Cropper(output_size=64, resize_size=160, strategy="largest", det_threshold=0.6, batch_size=2, output_format="png",
                         device="cuda:0", weights={"retinaface": sd})
    c = mk()
    c.io_processes = (2, 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c.process_dir(image_dir, str(tmp_path / "first"), desc=None)
        pool = c._io_procs
        assert pool is not None and pool.healthy()
        for w in pool._readers:                                  # every decoder gone: the next run cannot succeed with this pool
            w.proc.kill()
            w.proc.wait(timeout=5)
        c.process_dir(image_dir, str(tmp_path / "second"), desc=None)      # unhealthy pool replaced up front
        assert c._io_procs is not pool and pool.closed and c._io_procs.healthy()
        # a death DURING a run: the run raises, the pool is dropped, the run after it works
        pool2 = c._io_procs
        real = pool2.read_many

        def dying(paths):                                        # dead AND reaped before the request: the send fails
            for w in pool2._readers:
                w.proc.kill()
                w.proc.wait(timeout=5)
            return real(paths)
        pool2.read_many = dying
        with pytest.raises(RuntimeError, match="I/O worker process"):
            c.process_dir(image_dir, str(tmp_path / "broken"), desc=None)
        assert c._io_procs is None and pool2.closed
        c.process_dir(image_dir, str(tmp_path / "third"), desc=None)
        # the opposite ordering: the worker dies with the request already queued (stopped first, killed after the send) —
        # the failure then comes from the reply, and must be the same error
        import signal
        import threading
        pool3 = c._io_procs
        real3, timers = pool3.read_many, []

        def dying_after_send(paths):
            for w in pool3._readers:
                if w.proc.poll() is None:
                    os.kill(w.proc.pid, signal.SIGSTOP)
                    timers.append(threading.Timer(0.3, w.proc.kill))
                    timers[-1].start()
            return real3(paths)
        pool3.read_many = dying_after_send
        with pytest.raises(RuntimeError, match="I/O worker process"):
            c.process_dir(image_dir, str(tmp_path / "broken2"), desc=None)
        for t in timers:
            t.join()
        assert c._io_procs is None and pool3.closed
        c.process_dir(image_dir, str(tmp_path / "fourth"), desc=None)
    ref = sorted(os.listdir(tmp_path / "first"))
    assert ref and all(sorted(os.listdir(tmp_path / d)) == ref for d in ("second", "third", "fourth"))
    for f in ref:
        assert (tmp_path / "first" / f).read_bytes() == (tmp_path / "third" / f).read_bytes() == (tmp_path / "fourth" / f).read_bytes()
