"""Where does process_dir's host time go?  Wraps the stages with wall-clock timers (num_processes=1) — helper."""
import os, sys, tempfile, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import face_crop_plus_amd.cropper as CR
from face_crop_plus_amd import Cropper, utils, batch as B
T = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); T[name] += time.perf_counter() - t0; return r
    return w
n, size = 512, 640
with tempfile.TemporaryDirectory() as d:
    src, dst = os.path.join(d, "in"), os.path.join(d, "out")
    os.makedirs(src)
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (size // 8, size // 8, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((size, size), Image.BICUBIC))
    for i in range(n):
        Image.fromarray(np.roll(img, i, 1)).save(os.path.join(src, f"{i:05d}.jpg"), quality=90)
    c = Cropper(resize_size=size, batch_size=64, num_processes=1, device="cuda:0", weights={"retinaface": "generated"})
    c.process_dir(src, dst + "_warm", desc=None)
    CR.build_batch = timed("build_batch", CR.build_batch)
    c.det_model.predict = timed("det.predict (incl. sync)", c.det_model.predict)
    c._crop_align_device = timed("crop_align_device", c._crop_align_device)
    c.save_groups = timed("save_groups (submit)", c.save_groups)
    c._process_images = timed("_process_images total", c._process_images)
    t0 = time.perf_counter()
    c.process_dir(src, dst, desc=None)
    tot = time.perf_counter() - t0
    print(f"total {tot*1e3:.0f} ms for {n} images = {n/tot:.0f} img/s")
    for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v*1e3:8.1f} ms")
