"""Host I/O and weight-loading policy (CPU only)."""
import io
import os
import warnings

import numpy as np
import pytest


def _smooth(h=96, w=128):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return np.stack([127 + 100 * np.sin(xx / 9 + c) * np.cos(yy / 7) for c in range(3)], -1).clip(0, 255).astype(np.uint8)


def test_jpeg_written_at_opencv_default_quality(tmp_path):
    """cv2.imwrite's default is quality 95 / 4:2:0 (the reference writer, cropper.py:605-609); Pillow's own default
    (75) would give visibly lossier, much smaller files."""
    from PIL import Image
    from face_crop_plus_amd.utils import write_image
    img = _smooth()
    assert write_image(str(tmp_path / "a.jpg"), img) is True
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG")                 # Pillow default, for comparison
    size95, size75 = (tmp_path / "a.jpg").stat().st_size, len(buf.getvalue())
    assert size95 > 1.3 * size75
    with Image.open(tmp_path / "a.jpg") as im:
        assert im.get_format_mimetype() == "image/jpeg"
        q = im.quantization[0]
        back = np.asarray(im.convert("RGB")).astype(np.float64)
    assert max(q) <= 12, "luma quantisation table is coarser than quality 95"      # q95 table: entries 1..10
    psnr = 10 * np.log10(255 ** 2 / np.mean((back - img) ** 2))
    psnr75 = 10 * np.log10(255 ** 2 / np.mean((np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")).astype(np.float64) - img) ** 2))
    assert psnr > 38 and psnr > psnr75 + 3, (psnr, psnr75)


def test_png_webp_lossless_and_unknown_extension(tmp_path):
    from PIL import Image
    from face_crop_plus_amd.utils import write_image
    img = _smooth()
    mask = (img[..., 0] > 127).astype(np.uint8) * 255
    for name, arr in (("a.png", img), ("m.png", mask), ("a.webp", img), ("a.bmp", img)):
        assert write_image(str(tmp_path / name), arr)
        back = np.asarray(Image.open(tmp_path / name))
        assert np.array_equal(back, arr), name
    with pytest.warns(UserWarning, match="Could not write"):
        assert write_image(str(tmp_path / "a.xyz"), img) is False
    assert not (tmp_path / "a.xyz").exists()


def test_extensions_outside_the_tuned_table_still_write(tmp_path):
    """cv2.imwrite (cropper.py:605-609) also encodes .ppm / .pgm / .pnm / .jp2: an input of such a type, with
    output_format=None, must produce an output file (Pillow picks the encoder from the extension)."""
    from PIL import Image, features
    from face_crop_plus_amd.utils import write_image
    img = _smooth()
    mask = (img[..., 0] > 127).astype(np.uint8) * 255
    cases = [("a.ppm", img), ("m.pgm", mask), ("a.pnm", img)]
    if features.check_codec("jpg_2000"):
        cases.append(("a.jp2", img))
    for name, arr in cases:
        assert write_image(str(tmp_path / name), arr) is True, name
        back = np.asarray(Image.open(tmp_path / name))
        assert back.shape == arr.shape, name
        if not name.endswith(".jp2"):
            assert np.array_equal(back, arr), name


def test_write_errors_propagate_and_leave_no_partial_file(tmp_path):
    """Only "no encoder for this extension" is warn-and-skip (what cv2.imwrite's False becomes, cropper.py:605-609);
    a real I/O error raises, no truncated file stays behind, and formats cv2.imwrite refuses (.gif) are not written."""
    from face_crop_plus_amd.utils import write_image
    img = _smooth()
    with pytest.raises(OSError):
        write_image(str(tmp_path / "no_such_dir" / "a.ppm"), img)
    assert os.listdir(tmp_path) == []
    with pytest.warns(UserWarning, match="no encoder"):
        assert write_image(str(tmp_path / "a.gif"), img) is False
    assert os.listdir(tmp_path) == []


def test_emit_after_process_dir_unwinds(tmp_path):
    """A write task still in flight when process_dir's attributes are reset must finish (it holds its own references),
    and a slot is given back when the executor refuses the task."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    import face_crop_plus_amd.cropper as CR
    c = CR.Cropper.__new__(CR.Cropper)
    ex, writes, slots = ThreadPoolExecutor(2), [], threading.BoundedSemaphore(2)
    c._io, c._write_lock = (ex, writes, slots), threading.Lock()
    c._emit(str(tmp_path / "a.png"), np.zeros((2, 2, 3), np.uint8))
    c._io = None                                                   # what process_dir's finally block does
    for w in writes:
        w.result()                                                 # no AttributeError inside the task
    assert (tmp_path / "a.png").exists()
    ex.shutdown(wait=True)
    c._io = (ex, writes, slots)
    with pytest.raises(RuntimeError):                              # submit after shutdown
        c._emit(str(tmp_path / "b.png"), np.zeros((2, 2, 3), np.uint8))
    assert slots.acquire(blocking=False) and slots.acquire(blocking=False)     # both slots are free again


def test_read_image_applies_exif_orientation(tmp_path):
    """cv2.imread (utils.py:262) honours the EXIF orientation tag; so must the Pillow reader."""
    from PIL import Image
    from face_crop_plus_amd.utils import read_image
    img = _smooth(60, 90)
    im = Image.fromarray(img)
    exif = im.getexif()
    exif[0x0112] = 6                                  # "rotate 90 CW to display"
    im.save(tmp_path / "rot.png", exif=exif)          # PNG keeps the pixels exact
    got = read_image(str(tmp_path / "rot.png"))
    assert got.shape == (90, 60, 3)
    assert np.array_equal(got, np.rot90(img, k=-1))
    im.save(tmp_path / "plain.png")
    assert np.array_equal(read_image(str(tmp_path / "plain.png")), img)
    (tmp_path / "bad.png").write_bytes(b"nope")
    with pytest.warns(UserWarning, match="Could not read"):
        assert read_image(str(tmp_path / "bad.png")) is None


def test_weights_never_fall_back_silently(tmp_path, monkeypatch):
    """No checkpoint + no download => an error naming the places searched; random weights only on explicit opt-in."""
    import torch
    from face_crop_plus_amd import weights as W
    monkeypatch.setenv("FCP_WEIGHTS_DIR", str(tmp_path / "w"))
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "th"))
    monkeypatch.setenv("FCP_OFFLINE", "1")
    monkeypatch.delenv("FCP_WEIGHTS", raising=False)
    assert W.checkpoint_dirs() == [str(tmp_path / "w"), str(tmp_path / "th" / "hub" / "checkpoints")]
    with pytest.raises(FileNotFoundError, match="bise_parser.pth"):
        W.load_state_dict("bisenet")
    assert W.load_state_dict("bisenet", "generated")["conv_out.conv_out.weight"].shape == (19, 256, 1, 1)
    monkeypatch.setenv("FCP_WEIGHTS", "generated")
    with pytest.warns(UserWarning, match="RANDOM weights"):
        W.load_state_dict("bisenet")
    monkeypatch.delenv("FCP_WEIGHTS")
    # a real file in the torch hub cache (where the reference's download lands) is picked up
    os.makedirs(tmp_path / "th" / "hub" / "checkpoints")
    sd = W.generate_state_dict("bisenet", seed=3)
    torch.save(sd, tmp_path / "th" / "hub" / "checkpoints" / "bise_parser.pth")
    got = W.load_state_dict("bisenet")
    assert torch.equal(got["ffm.conv1.weight"], sd["ffm.conv1.weight"])
    # download path: a failing fetch is reported, a working one is validated and used
    os.remove(tmp_path / "th" / "hub" / "checkpoints" / "bise_parser.pth")
    monkeypatch.delenv("FCP_OFFLINE")
    monkeypatch.setattr(W, "_fetch_checkpoint", lambda m: (_ for _ in ()).throw(OSError("no route")))
    with pytest.raises(FileNotFoundError, match="no route"):
        W.load_state_dict("bisenet")
    monkeypatch.setattr(W, "_fetch_checkpoint", lambda m: sd)
    assert torch.equal(W.load_state_dict("bisenet")["ffm.conv2.weight"], sd["ffm.conv2.weight"])


def test_bounded_async_writes(tmp_path, monkeypatch):
    """process_dir's writer: at most MAX_PENDING_WRITES tasks in flight, errors surface at a later emit."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    import face_crop_plus_amd.cropper as CR
    gate, inflight, peak = threading.Event(), [0], [0]
    lock = threading.Lock()

    def slow_write(path, img):
        with lock:
            inflight[0] += 1
            peak[0] = max(peak[0], inflight[0])
        gate.wait(5)
        with lock:
            inflight[0] -= 1
        if path.endswith("boom.png"):
            raise OSError("disk full")
    monkeypatch.setattr(CR, "write_image", slow_write)
    c = CR.Cropper.__new__(CR.Cropper)
    c.MAX_PENDING_WRITES = 3
    c._io, c._write_lock = (ThreadPoolExecutor(8), [], threading.BoundedSemaphore(3)), threading.Lock()
    c.strategy, c.output_format = "largest", None
    t = threading.Thread(target=lambda: c.save_group([np.zeros((2, 2, 3), np.uint8)] * 6,
                                                     np.array([f"f{i}.png" for i in range(6)]), str(tmp_path)))
    t.start()
    t.join(0.5)
    assert t.is_alive() and peak[0] == 3          # the producer is blocked on the 4th write
    gate.set()
    t.join(5)
    assert not t.is_alive() and peak[0] == 3
    c._emit(str(tmp_path / "boom.png"), np.zeros((2, 2, 3), np.uint8))
    for w in list(c._io[1]):
        try:
            w.result()
        except OSError:
            pass
    with pytest.raises(OSError, match="disk full"):
        c._emit(str(tmp_path / "next.png"), np.zeros((2, 2, 3), np.uint8))
    c._io[0].shutdown(wait=True)


def test_selfcheck_mode_policy(monkeypatch):
    """The range guard runs by itself for weights that come from a checkpoint, never for in-memory / generated ones,
    and FCP_SELFCHECK overrides both ways."""
    from face_crop_plus_amd import engine as E
    monkeypatch.delenv("FCP_SELFCHECK", raising=False)
    assert E.selfcheck_mode(None) and E.selfcheck_mode("/some/retinaface_detector.pth")
    assert not E.selfcheck_mode("generated") and not E.selfcheck_mode({"a": 1})
    monkeypatch.setenv("FCP_SELFCHECK", "0")
    assert not E.selfcheck_mode(None)
    monkeypatch.setenv("FCP_SELFCHECK", "1")
    assert E.selfcheck_mode("generated") and E.selfcheck_mode({"a": 1})


def test_io_workers_are_divided_between_the_ranks_of_a_node():
    """Pre-flight for the 8-GPU node: one process per GPU, the ranks of a node share the host's cores — every rank sizes its
    decode / encode worker pool for cores / LOCAL_WORLD_SIZE (DESIGN.md section 6 carries the table this pins)."""
    from face_crop_plus_amd.cropper import Cropper
    f = Cropper.default_io_processes
    assert f(128, 1) == (12, 3)          # the measured optimum on the 2 x 64-core box
    assert f(128, 8) == (5, 2)           # 8 ranks: 16 cores each -> 40 decoders + 16 encoders on the node
    assert f(256, 8) == (10, 3)
    assert f(16, 8) == (2, 1)            # never below a working pool
    assert f(4, 1) == (2, 1)
    total = lambda cores, r: r * sum(f(cores, r))
    assert total(128, 8) <= 128 and total(256, 8) <= 256      # the node is not oversubscribed by I/O processes
