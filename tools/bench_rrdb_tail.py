"""RRDB x4 tail (upconv2 -> HRconv -> conv_last) timing per band height at a 1024^2 input — profiling helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E, weights
from face_crop_plus_amd.rrdb import RRDBNet
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
m = RRDBNet(0.02).load(dev, weights.generate_state_dict("rrdb"), precision=prec)
f = 1 if m.precision == 1 else 0
fea2 = E.Act(torch.randn(1, 2 * H, 2 * H, 64, device=dev))
if f:
    fea2 = E.f32_to_split32(fea2)
E.Autotune.enabled = True
for band in [int(b) for b in (sys.argv[3].split(",") if len(sys.argv) > 3 else "64,128,256,512".split(","))]:
    RRDBNet.TAIL_BAND = band
    for _ in range(2):
        m._tail(fea2, f)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        m._tail(fea2, f)
    e1.record(); torch.cuda.synchronize()
    print(f"{H}^2 {prec} band {band:4d}: {e0.elapsed_time(e1) / 3:8.2f} ms", flush=True)
img = torch.randint(0, 256, (1, H, H, 3), dtype=torch.uint8, device=dev)
RRDBNet.TAIL_BAND = 128
for _ in range(2):
    m.enhance_u8(img, [0])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    m.enhance_u8(img, [0])
e1.record(); torch.cuda.synchronize()
print(f"enhance_u8 {H}^2 {prec}: {e0.elapsed_time(e1) / 3:8.2f} ms / image")
