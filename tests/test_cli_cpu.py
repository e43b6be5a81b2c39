"""CLI flag surface (reference __main__.py:117-226) — host logic only."""
import json
import os

import pytest


def test_defaults_and_threshold_sentinels(tmp_path):
    from face_crop_plus_amd.__main__ import parse_args
    a = parse_args(["-i", str(tmp_path)])
    assert a["output_size"] == [256, 256] and a["resize_size"] == [1024, 1024]
    assert a["strategy"] == "largest" and a["padding"] == "constant" and a["face_factor"] == 0.65
    assert a["det_threshold"] == 0.6 and a["enh_threshold"] is None      # -1 -> disabled
    assert a["batch_size"] == 8 and a["num_processes"] == 1 and a["device"].startswith("cuda:")
    b = parse_args(["-i", "x", "-dt", "-1", "-et", "0.001", "-st", "all", "-s", "200", "300", "-a",
                    "-ag", '{"glasses": [6]}', "-mg", '{"eyes": [4, 5]}'])
    assert b["det_threshold"] is None and b["enh_threshold"] == 0.001 and b["strategy"] == "all"
    assert b["output_size"] == [200, 300] and b["allow_skew"] is True
    assert b["attr_groups"] == {"glasses": [6]} and b["mask_groups"] == {"eyes": [4, 5]}


def test_config_file_supplies_defaults(tmp_path):
    from face_crop_plus_amd.__main__ import parse_args
    cfg = tmp_path / "demo.json"
    cfg.write_text(json.dumps({"input_dir": "imgs", "output-size": [128], "strategy": "best", "det-threshold": 0.7}))
    a = parse_args(["-c", str(cfg), "-st", "all"])
    assert a["input_dir"] == "imgs" and a["output_size"] == [128] and a["det_threshold"] == 0.7
    assert a["strategy"] == "all"            # the command line wins over the config


def test_input_dir_is_required():
    from face_crop_plus_amd.__main__ import parse_args
    with pytest.raises(ValueError, match="Input directory must be specified"):
        parse_args([])


def test_host_utils(tmp_path):
    import numpy as np
    from PIL import Image
    from face_crop_plus_amd import utils
    rng = np.random.default_rng(0)
    for name, (h, w) in {"a.png": (50, 80), "b.png": (90, 40)}.items():
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / name)
    (tmp_path / "broken.jpg").write_bytes(b"not an image")
    with pytest.warns(UserWarning):
        imgs, names = utils.read_images(["a.png", "broken.jpg", "b.png"], str(tmp_path))
    assert names.tolist() == ["a.png", "b.png"] and imgs[0].shape == (50, 80, 3)
    from face_crop_plus_amd.batch import batch_geometry
    from oracle import batch_ref
    for im in imgs + [np.zeros((2160, 3840, 3), np.uint8), np.zeros((64, 64, 3), np.uint8)]:
        h, w = im.shape[:2]
        for size in ((64, 64), (512, 256), (1024, 1024)):
            ww, hh, pad, unscale, interp = batch_geometry(h, w, size)
            assert (ww, hh, pad, unscale, ("cubic", "area")[interp]) == batch_ref.geometry(h, w, size)
    assert batch_geometry(50, 80, (64, 64))[2] == [12, 12, 0, 0]   # [t, b, l, r], utils.py:322-331
    assert batch_geometry(90, 40, (64, 64))[2] == [0, 0, 18, 18]
    with pytest.raises(RuntimeError, match="no CPU fallback"):     # the pixels come from the GPU builder only
        utils.as_batch(imgs, (64, 64), device="cpu")
    assert utils.get_ldm_slices(5, 68)[0] == slice(36, 42)
    with pytest.raises(ValueError):
        utils.get_ldm_slices(5, 7)
    lf = tmp_path / "lm.txt"
    lf.write_text("a.png 1 2 3 4 5 6 7 8 9 10\nb.png 2 3 4 5 6 7 8 9 10 11\n")
    lm, fn = utils.parse_landmarks_file(str(lf))
    assert lm.shape == (2, 5, 2) and fn.tolist() == ["a.png", "b.png"]


def test_clean_names(tmp_path):
    """utils.clean_names (reference utils.py:344-453): transliteration, excluded characters, truncation,
    case-insensitive de-duplication; copy mode and in-place mode."""
    from face_crop_plus_amd import utils
    src = tmp_path / "in"
    src.mkdir()
    names = ["České.jpg", "a?b#c.d.png", "dup.jpg", "DUP.jpg", "dup-1.jpg", "plain.png", "x" * 80 + ".jpg"]
    for i, n in enumerate(names):
        (src / n).write_bytes(bytes([i]))
    out = tmp_path / "out"
    utils.clean_names(str(src), str(out), desc=None)
    got = sorted(os.listdir(out))
    assert len(got) == len(names) and len({g.lower() for g in got}) == len(names)       # unique ignoring case
    assert "Ceske.jpg" in got and "abcd.png" in got and "plain.png" in got
    assert all(g.isascii() and not (set(os.path.splitext(g)[0]) & set("?#.")) for g in got)
    assert (out / "Ceske.jpg").read_bytes() == bytes([0]) and sorted(os.listdir(src)) == sorted(names)   # copies
    # truncation budget: max_chars counts the directory path too
    utils.clean_names(str(src), str(tmp_path / "short"), max_chars=len(str(src)) + 20, desc=None)
    assert max(len(n) for n in os.listdir(tmp_path / "short")) <= 22                    # 20 + a "-N" suffix at most
    with pytest.raises(RuntimeError, match="too long"):
        utils.clean_names(str(src), None, max_chars=len(str(src)) + 3, desc=None)
    utils.clean_names(str(src), None, desc=None)                                         # in place
    assert sorted(os.listdir(src)) == got
