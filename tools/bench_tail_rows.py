"""One tile per CU, tile height R = 32..256: how does the 256-row kernel's time scale with the rows a tile really has?
(experiment for the balanced M-tile schedule).  python tools/bench_tail_rows.py [k] [cin] [cout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 256
tn = 256 if cout >= 256 else 128
pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, 1, k // 2, dev, precision="f16x3")
gn = -(-cout // tn)
for R in (32, 64, 96, 128, 160, 192, 224, 256):
    slots = 256 // gn
    h, w = 16, slots * R // 16
    x = E.f32_to_split32(E.Act(torch.randn(1, h, w, cin, device=dev).relu()))
    out = E.Act.empty(1, h, w, cout, dev, 1)
    for _ in range(3):
        E.conv(pc, x, out, act_slope=0.0, tile_m=256, tile_n=tn, balance_tail=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        E.conv(pc, x, out, act_slope=0.0, tile_m=256, tile_n=tn, balance_tail=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"R={R:3d}: {us:7.1f} us  {pc.flops_per_pixel * h * w / us / 1e6:6.1f} TF/s  ({slots * gn} tiles)", flush=True)
