"""Fixed-tile timings of the 256-row kernel on the detector's layer-3/4/FPN shapes — A/B helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
CASES = {"l3c1": (64, 40, 1024, 256, 1), "l3c2": (64, 40, 256, 256, 3), "l4c1": (64, 20, 2048, 512, 1), "l4c2": (64, 20, 512, 512, 3),
         "out1": (64, 80, 512, 256, 1), "merge1": (64, 80, 256, 256, 3), "l3b0c1": (64, 80, 512, 256, 1)}
tn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tot = 0.0
for nm, (b, h, cin, cout, k) in CASES.items():
    pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, 1, k // 2, dev, precision="f16x3")
    x = E.f32_to_split32(E.Act(torch.randn(b, h, h, cin, device=dev).relu()))
    out = E.Act.empty(b, h, h, cout, dev, 1)
    for _ in range(2):
        E.conv(pc, x, out, act_slope=0.0, tile_m=256, tile_n=tn)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        E.conv(pc, x, out, act_slope=0.0, tile_m=256, tile_n=tn)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    tot += us
    print(f"{nm:7s} {us:8.1f} us  {pc.flops_per_pixel * b * h * h / us / 1e6:6.1f} TFLOP/s", flush=True)
print(f"sum     {tot:8.1f} us")
