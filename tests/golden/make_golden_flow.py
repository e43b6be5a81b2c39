"""Golden vectors for the control flow of ``Cropper.process_batch`` / ``__init__`` / ``_init_models`` (build
container only): the reference's own method bodies (compiled out of cropper.py's syntax tree) run with recording
stand-ins for everything they call — models, read_images, as_batch, crop_align, save_groups.  Recorded: which
models a configuration builds, attribute normalisation, and for each scenario the (indices, landmarks, paddings)
handed to crop_align and the file names handed to save_groups.

    python tests/golden/make_golden_flow.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_host import REF, extract  # noqa: E402

LOG = []


class Model:
    def __init__(self, kind, *a, **k):
        self.kind, self.args = kind, [repr(x) for x in a] + [f"{n}={v!r}" for n, v in k.items()]
        LOG.append(["construct", kind] + self.args)

    def load(self, device=None):
        LOG.append(["load", self.kind, str(device)])


def main():
    ns = {"np": np, "torch": torch, "RetinaFace": lambda *a, **k: Model("RetinaFace", *a, **k),
          "RRDBNet": lambda *a, **k: Model("RRDBNet", *a, **k), "BiSeNet": lambda *a, **k: Model("BiSeNet", *a, **k)}
    extract(os.path.join(REF, "utils.py"), {"get_landmark_slices_5", "get_ldm_slices"}, ns)
    extract(os.path.join(REF, "cropper.py"), {"Cropper.__init__", "Cropper._init_models", "Cropper.process_batch"}, ns)
    out = {}

    class Self:
        pass

    # ---- __init__ normalisation + which models get built
    for name, kw in {
        "defaults": {},
        "sizes_int": {"output_size": 300, "resize_size": 640},
        "sizes_len1": {"output_size": [200], "resize_size": [512]},
        "no_detection": {"det_threshold": None},
        "enhance_and_parse": {"enh_threshold": 0.001, "attr_groups": {"g": [6]}},
        "masks_only_with_landmarks": {"mask_groups": {"eyes": [4, 5]}, "landmarks": (np.zeros((1, 5, 2), np.float32), np.array(["a.jpg"]))},
    }.items():
        LOG.clear()
        s = Self()
        s._init_models = lambda s=s: ns["_init_models"](s)
        s._init_landmarks_target = lambda: LOG.append(["_init_landmarks_target"])
        ns["__init__"](s, **kw)
        out[f"init_{name}"] = {
            "log": [list(x) for x in LOG], "output_size": list(s.output_size), "resize_size": list(s.resize_size),
            "device": str(s.device), "num_std_landmarks": s.num_std_landmarks,
            "models": {k: (None if getattr(s, k) is None else getattr(s, k).kind) for k in ("det_model", "enh_model", "par_model")}}

    # ---- process_batch
    def scenario(name, files, landmarks=None, det=None, images_hw=None, num_std=5):
        LOG.clear()
        s = Self()
        s.landmarks, s.det_model, s.enh_model, s.par_model = landmarks, det, None, None
        s.resize_size, s.device, s.num_std_landmarks = (64, 64), "cpu", num_std
        imgs = [np.full((hw[0], hw[1], 3), i, np.uint8) for i, hw in enumerate(images_hw or [(8, 8)] * len(files))]
        readable = [f for f in files if not f.startswith("broken")]
        ns["read_images"] = lambda fn, d: ([imgs[files.index(f)] for f in fn if f in readable], np.array([f for f in fn if f in readable]))
        ns["as_batch"] = lambda im, size: (np.zeros((len(im), size[1], size[0], 3), np.uint8), np.ones(len(im)),
                                           np.array([[i, i, 0, 0] for i in range(len(im))]))
        ns["as_tensor"] = lambda x, dev: x
        ns["as_numpy"] = lambda x: x

        def crop_align(images, padding, indices, lms):
            LOG.append(["crop_align", None if padding is None else np.asarray(padding).tolist(), list(map(int, indices)),
                        np.asarray(lms).round(4).tolist()])
            return np.zeros((len(indices), 4, 4, 3), np.uint8)
        s.crop_align = crop_align
        s.save_groups = lambda faces, names, out_dir, a, m: LOG.append(
            ["save_groups", len(faces), [str(n) for n in names], out_dir, a is None, m is None])
        ns["process_batch"](s, files, "in", "out")
        out[f"flow_{name}"] = [list(x) for x in LOG]

    lm5 = np.arange(40, dtype=np.float32).reshape(4, 5, 2)
    scenario("given_landmarks", ["a.jpg", "b.png", "c.png", "broken.png"], landmarks=(lm5, np.array(["b.png", "a.jpg", "b.png", "zzz.png"])))
    scenario("given_landmarks_none_match", ["c.png"], landmarks=(lm5, np.array(["b.png", "a.jpg", "b.png", "zzz.png"])))
    lm68 = np.linspace(0, 1, 2 * 68 * 2, dtype=np.float32).reshape(2, 68, 2)
    scenario("given_landmarks_68", ["a.jpg", "b.png"], landmarks=(lm68, np.array(["a.jpg", "b.png"])))

    class Det:
        def predict(self, images):
            LOG.append(["det.predict", list(images.shape)])
            return np.array([[[10, 20]] * 5, [[30, 40]] * 5, [[50, 60]] * 5], np.float32), [0, 2, 2]
    scenario("detect", ["a.jpg", "b.png", "c.png"], det=Det())

    class DetNone:
        def predict(self, images):
            return np.zeros((0, 5, 2), np.float32), []
    scenario("detect_no_faces", ["a.jpg"], det=DetNone())
    scenario("no_detection_no_landmarks", ["a.jpg", "b.png"])
    json.dump(out, open(os.path.join(HERE, "flow.json"), "w"), indent=0, sort_keys=True)
    print("wrote flow.json:", sorted(out))


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
