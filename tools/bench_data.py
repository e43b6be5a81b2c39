"""Data dependence of conv time (zeros vs random operands: power / clock effects) — profiling helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
CASES = {"l3c1": (64, 40, 1024, 256, 1), "l4c1": (64, 20, 2048, 512, 1), "l3c2": (64, 40, 256, 256, 3)}
for nm, (b, h, cin, cout, k) in CASES.items():
    for wz in (0, 1):
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5 * wz
        pc = E.pack_conv(w, torch.zeros(cout), None, 1, k // 2, dev, precision="f16x3")
        for xz in (0, 1, 2):
            xf = torch.randn(b, h, h, cin, device=dev) * (1 if xz else 0)
            if xz == 2:
                xf = xf.relu()
            x = E.f32_to_split32(E.Act(xf))
            out = E.Act.empty(b, h, h, cout, dev, 1)
            for _ in range(3):
                E.conv(pc, x, out, act_slope=0.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                E.conv(pc, x, out, act_slope=0.0)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            fl = pc.flops_per_pixel * b * h * h
            print(f"{nm} w={'rand' if wz else 'zero'} x={('zero','randn','relu')[xz]:5s} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
