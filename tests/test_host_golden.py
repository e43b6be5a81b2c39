"""CPU: host-side logic against golden vectors produced by the reference's OWN functions
(tests/golden/make_golden_host.py compiles them out of the reference's utils.py / cropper.py syntax trees):
the 5-point target (cropper.py:392-439), landmark slices (utils.py:90-168), landmark files (utils.py:21-88)
and clean_names (utils.py:344-453)."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "host_logic.npz"))


def test_landmark_target_matches_reference(gold):
    from face_crop_plus_amd import align
    from face_crop_plus_amd.cropper import landmarks_target
    from oracle import align_ref
    assert np.array_equal(align.STANDARD_LANDMARKS_5, gold["standard_landmarks_5"])
    assert np.array_equal(align_ref.STANDARD_LANDMARKS_5, gold["standard_landmarks_5"])
    i = 0
    while f"target_{i}" in gold.files:
        w, h, ff = gold[f"target_{i}_cfg"]
        for fn in (landmarks_target, align_ref.landmarks_target):
            got = fn((int(w), int(h)), float(ff))
            assert got.dtype == np.float32 and np.array_equal(got, gold[f"target_{i}"]), (i, fn.__module__)
        i += 1
    assert i >= 5


def test_landmark_slices_match_reference(gold):
    from face_crop_plus_amd import utils
    for k in (5, 12, 17, 21, 29, 49, 68, 98, 106):
        got = np.array([[s.start, s.stop] for s in utils.get_ldm_slices(5, k)])
        assert np.array_equal(got, gold[f"slices_{k}"]), k
    for a, b in ((5, 7), (6, 68)):
        with pytest.raises(ValueError) as e:
            utils.get_ldm_slices(a, b)
        assert str(e.value) == str(gold[f"slices_bad_{a}_{b}"])


def test_unsupported_standard_landmark_count_message(gold):
    from face_crop_plus_amd.cropper import Cropper
    c = Cropper.__new__(Cropper)                     # no GPU needed for this check
    c.num_std_landmarks, c.output_size, c.face_factor = 7, (256, 256), 0.65
    with pytest.raises(ValueError) as e:
        c._init_landmarks_target()
    assert str(e.value) == str(gold["target_bad_raises"])


def test_landmark_files_match_reference(gold, tmp_path):
    from face_crop_plus_amd import utils
    for fn in ("lm.csv", "lm.txt"):
        p = tmp_path / fn
        p.write_text(str(gold[f"lmfile_{fn}_text"]))
        lm, names = utils.parse_landmarks_file(str(p))
        assert lm.dtype == np.float32 and np.array_equal(lm, gold[f"lmfile_{fn}_landmarks"])
        assert names.tolist() == gold[f"lmfile_{fn}_names"].tolist()
    # The reference itself raises on these two (np.array(dict.values()) is a 0-d object array: TypeError;
    # a single data row collapses to 1-D: IndexError).  Here they parse to the evidently intended result.
    assert str(gold["lmfile_lm.json_error"]) == "TypeError" and str(gold["lmfile_one.txt_error"]) == "IndexError"
    p = tmp_path / "lm.json"
    p.write_text(str(gold["lmfile_lm.json_text"]))
    lm, names = utils.parse_landmarks_file(str(p))
    assert lm.shape == (3, 5, 2) and np.allclose(lm, gold["lmfile_lm.txt_landmarks"]) and names.tolist() == ["a.jpg", "b c.png", "d.jpeg"]
    p = tmp_path / "one.txt"
    p.write_text(str(gold["lmfile_one.txt_text"]))
    lm, names = utils.parse_landmarks_file(str(p))
    assert lm.shape == (1, 5, 2) and names.tolist() == ["only.jpg"]


def test_clean_names_matches_reference(tmp_path, monkeypatch):
    from face_crop_plus_amd import utils
    g = json.load(open(os.path.join(G, "host_clean_names.json")))
    src = tmp_path / "names"
    src.mkdir()
    for i, n in enumerate(g["raw"]):
        (src / n).write_bytes(bytes([i]))
    real_listdir = os.listdir
    # de-duplication suffixes depend on the iteration order: replay the order the reference saw
    monkeypatch.setattr(os, "listdir", lambda d: list(g["listing_order"]) if os.path.abspath(d) == str(src) else real_listdir(d))
    utils.clean_names(str(src), str(tmp_path / "clean"), desc=None)
    got = {str((tmp_path / "clean" / n).read_bytes()[0]): n for n in real_listdir(tmp_path / "clean")}
    assert got == g["copy"]
    utils.clean_names(str(src), str(tmp_path / "short"), max_chars=len(str(src)) + g["short_max_chars_extra"], desc=None)
    got = {str((tmp_path / "short" / n).read_bytes()[0]): n for n in real_listdir(tmp_path / "short")}
    assert got == g["short"]
