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


@pytest.fixture(scope="module")
def plumb():
    return json.load(open(os.path.join(G, "plumbing.json")))


def test_batch_geometry_matches_reference_as_batch_calls(plumb):
    """What the reference asks OpenCV for inside as_batch (utils.py:316-335), recorded with a stand-in cv2:
    resize target, interpolation constant, border widths, un-scale factors, batch shape."""
    from face_crop_plus_amd.batch import batch_geometry
    from face_crop_plus_amd.align import border_code
    from oracle import batch_ref
    for key, rec in plumb.items():
        if not key.startswith("as_batch_") or key == "as_batch_border_mode":
            continue
        size = eval(key[len("as_batch_"):])
        size = (size, size) if isinstance(size, int) else tuple(size)
        calls = rec["calls"]
        assert rec["batch_shape"] == [len(rec["shapes"]), size[1], size[0], 3]
        for i, (h, w) in enumerate(rec["shapes"]):
            rz, mb = calls[2 * i], calls[2 * i + 1]
            ww, hh, pad, unscale, interp = batch_geometry(h, w, size)
            assert rz[0] == "resize" and rz[1] == [h, w, 3] and rz[2] == [ww, hh], (key, h, w)
            assert rz[3] == (3 if interp == 1 else 2)                       # cv2.INTER_AREA = 3, cv2.INTER_CUBIC = 2
            assert mb[0] == "copyMakeBorder" and mb[1] == [hh, ww, 3] and mb[2] == pad == rec["paddings"][i] and mb[3] == 0
            assert unscale == rec["unscales"][i]
            assert batch_ref.geometry(h, w, size) == (ww, hh, pad, unscale, "area" if interp else "cubic")
    assert plumb["as_batch_border_mode"][1][3] == border_code("reflect_101") == 4


def test_save_group_paths_match_reference(plumb, tmp_path, monkeypatch):
    """File naming / directory layout of save_group and save_groups (cropper.py:554-746): same paths, same order."""
    import face_crop_plus_amd.cropper as CR
    written = []
    monkeypatch.setattr(CR, "write_image", lambda path, img: written.append(os.path.relpath(path, tmp_path)))
    c = CR.Cropper.__new__(CR.Cropper)
    c._io = None
    faces = [np.zeros((4, 4, 3), np.uint8)] * 5 + [np.zeros((4, 4), np.uint8)]
    names = np.array(["a.jpg", "a.jpg", "b.png", "a.jpg", "c.jpeg", "b.png"])
    for strategy, fmt in (("all", None), ("largest", None), ("all", "png"), ("best", "jpg")):
        written.clear()
        c.strategy, c.output_format = strategy, fmt
        c.save_group(faces, names, str(tmp_path / "o"))
        assert written == [p[1] for p in plumb[f"save_group_{strategy}_{fmt}"] if p[0] == "imwrite"], (strategy, fmt)
    c.strategy, c.output_format = "all", None
    attr = {"glasses": [0, 2, 4], "no_glasses": [1, 3]}
    masks = {"eyes": ([0, 1, 4], np.zeros((3, 4, 4), np.uint8)), "hair": ([2], np.zeros((1, 4, 4), np.uint8))}
    for key, out, a, m in (("save_groups_attr_mask", "g", attr, masks), ("save_groups_none", "n", None, None),
                           ("save_groups_mask_only", "m", None, masks)):
        written.clear()
        c.save_groups(faces[:5], names[:5], str(tmp_path / out), a, m)
        assert written == plumb[key], key


def test_crop_align_plumbing_constants(plumb):
    """crop_align's OpenCV arguments (cropper.py:511-547): border constant, dsize = output_size as (w, h),
    ransacReprojThreshold = inf, the un-padded source slice, estimator choice, skipped faces."""
    from face_crop_plus_amd.align import border_code
    calls = plumb["crop_align"]["calls"]
    warps = [c for c in calls if c[0] == "warpAffine"]
    assert all(c[5] == border_code("reflect") == 2 and c[4] == [96, 112] for c in warps)
    assert [c[1] for c in warps] == [[40, 60, 3], [33, 60, 3], [40, 49, 3], [40, 49, 3]]     # image[t:h-b, l:w-r]
    assert all(c[2] for c in calls if c[0].startswith("estimate"))                            # threshold is inf
    assert plumb["crop_align"]["result_shape"] == [4, 112, 96, 3] and plumb["crop_align"]["result_values"] == [10, 20, 30, 30]
    assert [c[0] for c in plumb["crop_align_skew_list_nopad"]["calls"]] == ["estimateAffine2D", "warpAffine"]
    assert plumb["crop_align_empty"]["result_shape"] == [0]


def test_cli_matches_reference_parser(tmp_path):
    """Every flag, default, config-file override and threshold sentinel of the reference's own parser
    (__main__.py:10-249, golden: tests/golden/cli.json).  One difference by design: "auto" resolves to this
    rank's GPU ("cuda:<LOCAL_RANK>") instead of "cuda"."""
    from face_crop_plus_amd.__main__ import parse_args
    g = json.load(open(os.path.join(G, "cli.json")))
    cfg = tmp_path / "cfg.json"
    cfg.write_text(json.dumps(g["_config_file"]))
    n = 0
    for name, case in g.items():
        if name.startswith("_"):
            continue
        argv = [str(cfg) if a == "<CFG>" else a for a in case["argv"]]
        got, exp = parse_args(argv), dict(case["kwargs"])
        if exp["device"] == "cuda":
            assert got["device"] == f"cuda:{os.environ.get('LOCAL_RANK', '0')}"
            exp["device"] = got["device"]
        assert got == exp, (name, {k: (got.get(k), exp.get(k)) for k in set(got) | set(exp) if got.get(k) != exp.get(k)})
        n += 1
    assert n >= 7 and g["_no_input"] == "ValueError"
    with pytest.raises(ValueError):
        parse_args([])
