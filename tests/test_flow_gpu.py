"""GPU: control flow of Cropper.__init__ / _init_models / process_batch against what the reference's own method
bodies do (tests/golden/flow.json, recorded by running them with stand-ins for models and helpers)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def flow():
    return json.load(open(os.path.join(G, "flow.json")))


KW = {
    "defaults": {},
    "sizes_int": {"output_size": 300, "resize_size": 640},
    "sizes_len1": {"output_size": [200], "resize_size": [512]},
    "no_detection": {"det_threshold": None},
    "enhance_and_parse": {"enh_threshold": 0.001, "attr_groups": {"g": [6]}},
    "masks_only_with_landmarks": {"mask_groups": {"eyes": [4, 5]}, "landmarks": (np.zeros((1, 5, 2), np.float32), np.array(["a.jpg"]))},
}


@pytest.mark.parametrize("name", sorted(KW))
def test_init_builds_the_same_models(name, flow, device):
    from face_crop_plus_amd import Cropper
    exp = flow[f"init_{name}"]
    c = Cropper(device="cuda:0", weights={k: "generated" for k in ("retinaface", "rrdb", "bisenet")}, **KW[name])
    assert list(c.output_size) == exp["output_size"] and list(c.resize_size) == exp["resize_size"]
    assert c.num_std_landmarks == exp["num_std_landmarks"]
    got = {k: (None if getattr(c, k) is None else type(getattr(c, k)).__name__) for k in ("det_model", "enh_model", "par_model")}
    assert got == exp["models"]
    # constructor arguments the reference passes on (cropper.py:378-390)
    for entry in exp["log"]:
        if entry[:2] == ["construct", "RetinaFace"]:
            assert (repr(c.det_model.strategy), repr(c.det_model.vis_threshold)) == tuple(entry[2:4])
        if entry[:2] == ["construct", "RRDBNet"]:
            assert repr(c.enh_model.min_face_factor) == entry[2]
        if entry[:2] == ["construct", "BiSeNet"]:
            assert (repr(c.par_model.attr_groups), repr(c.par_model.mask_groups), repr(c.par_model.batch_size)) == tuple(entry[2:5])


def _run(c, files, tmp_path, monkeypatch, det=None):
    """process_batch with recorders in place of the device stages."""
    import face_crop_plus_amd.cropper as CR
    from PIL import Image
    log = []
    for i, f in enumerate(files):
        if f.startswith("broken"):
            (tmp_path / f).write_bytes(b"x")
        else:
            Image.fromarray(np.full((8, 8, 3), i, np.uint8)).save(tmp_path / f)
    if det is not None:
        c.det_model = det

    def crop_dev(images_dev, paddings, indices, lms):
        log.append(["crop_align", None if paddings is None else np.asarray(paddings).tolist(), list(map(int, indices)),
                    lms.cpu().numpy().round(4).tolist()])
        n = len(indices)
        return torch.zeros((n, 4, 4, 3), dtype=torch.uint8, device=c.device), torch.ones(n, dtype=torch.int32, device=c.device)

    def crop_np(images, padding, indices, lms):
        log.append(["crop_align", None if padding is None else np.asarray(padding).tolist(), list(map(int, indices)),
                    np.asarray(lms).round(4).tolist()])
        return np.zeros((len(indices), 4, 4, 3), np.uint8)
    c._crop_align_device, c.crop_align = crop_dev, crop_np
    c.save_groups = lambda faces, names, out, a, m: log.append(["save_groups", len(faces), [str(x) for x in names], out, a is None, m is None])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c.process_batch(files, str(tmp_path), "out")
    return log


def test_given_landmarks_flow(flow, tmp_path, monkeypatch, device):
    from face_crop_plus_amd import Cropper
    lm5 = np.arange(40, dtype=np.float32).reshape(4, 5, 2)
    names = np.array(["b.png", "a.jpg", "b.png", "zzz.png"])
    c = Cropper(landmarks=(lm5, names), device="cuda:0")
    assert _run(c, ["a.jpg", "b.png", "c.png", "broken.png"], tmp_path, monkeypatch) == flow["flow_given_landmarks"]
    c = Cropper(landmarks=(lm5, names), device="cuda:0")
    assert _run(c, ["c.png"], tmp_path, monkeypatch) == flow["flow_given_landmarks_none_match"] == []
    lm68 = np.linspace(0, 1, 2 * 68 * 2, dtype=np.float32).reshape(2, 68, 2)
    c = Cropper(landmarks=(lm68, np.array(["a.jpg", "b.png"])), device="cuda:0")
    got, exp = _run(c, ["a.jpg", "b.png"], tmp_path, monkeypatch), flow["flow_given_landmarks_68"]
    assert got[0][:3] == exp[0][:3] and np.allclose(got[0][3], exp[0][3], atol=1e-4) and got[1] == exp[1]


def test_detect_flow(flow, tmp_path, monkeypatch, device):
    """Detected landmarks are un-padded with paddings[indices][:, None, [left, top]] (cropper.py:822) and the faces
    saved under the file names of their images; zero faces: silent return; no detector and no landmarks: the images
    themselves are saved."""
    from face_crop_plus_amd import Cropper

    class Det:
        strategy = "all"

        def predict(self, images):
            assert tuple(images.shape) == (3, 64, 64, 3)
            return np.array([[[10, 20]] * 5, [[30, 40]] * 5, [[50, 60]] * 5], np.float32), [0, 2, 2]
    from PIL import Image
    files, sizes = ["a.jpg", "b.png", "c.png"], [(8, 8), (16, 8), (8, 32)]     # (h, w): paddings differ per image
    for f, (h, w) in zip(files, sizes):
        Image.fromarray(np.zeros((h, w, 3), np.uint8)).save(tmp_path / f)
    exp = flow["flow_detect"]
    c3 = Cropper(resize_size=64, det_threshold=None, device="cuda:0")
    c3.det_model = Det()                                              # asserts the batch shape it is handed
    log = []

    def crop_dev(images_dev, pads, idx, lms):
        log.append([np.asarray(pads).tolist(), list(idx), lms.cpu().numpy().tolist()])
        return (torch.zeros((len(idx), 4, 4, 3), dtype=torch.uint8, device=c3.device),
                torch.ones(len(idx), dtype=torch.int32, device=c3.device))
    c3._crop_align_device = crop_dev
    c3.save_groups = lambda faces, names, out, a, m: log.append([len(faces), [str(x) for x in names]])
    c3.process_batch(files, str(tmp_path), "out")
    pads, idx, lms = log[0]
    assert idx == exp[1][2] == [0, 2, 2] and log[1] == [3, exp[2][2]]
    assert pads == [[0, 0, 0, 0], [0, 0, 16, 16], [24, 24, 0, 0]]   # as_batch geometry of 8x8, 16x8, 8x32 into 64x64
    raw = np.array([[[10, 20]] * 5, [[30, 40]] * 5, [[50, 60]] * 5], np.float32)
    unp = raw - np.array(pads)[idx][:, None, [2, 0]]
    assert np.array_equal(np.array(lms), unp)                        # reference: x -= left, y -= top

    class DetNone:
        strategy = "all"

        def predict(self, images):
            return np.zeros((0, 5, 2), np.float32), []
    c4 = Cropper(resize_size=64, det_threshold=None, device="cuda:0")
    assert _run(c4, ["a.jpg"], tmp_path, monkeypatch, det=DetNone()) == flow["flow_detect_no_faces"] == []
    c5 = Cropper(det_threshold=None, device="cuda:0")
    assert _run(c5, ["a.jpg", "b.png"], tmp_path, monkeypatch) == flow["flow_no_detection_no_landmarks"]
