"""Is the detection step power-limited?  Time single steps after idle gaps vs back-to-back — profiling helper."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
from face_crop_plus_amd.retinaface import RetinaFace
dev = torch.device("cuda:0")
E.Autotune.enabled = True
det = RetinaFace("largest", 0.6).load(dev, "generated")
imgs = torch.randint(0, 256, (64, 640, 640, 3), dtype=torch.uint8, device=dev)
for _ in range(3):
    det.predict(imgs)
torch.cuda.synchronize()
def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        det.detect(imgs) if hasattr(det, "detect") else det.predict(imgs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("back-to-back x20:", round(timed(20), 2), "ms/step")
for gap in (0.0, 0.05, 0.2, 1.0):
    ts = []
    for _ in range(6):
        time.sleep(gap)
        ts.append(timed(1))
    print(f"single step after {gap:4.2f}s idle:", [round(t, 2) for t in ts])
print("back-to-back x50:", round(timed(50), 2), "ms/step")
