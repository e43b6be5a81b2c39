"""RRDB dense-block layer timings at 1024^2 (split32, fp16x3) — profiling helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else H
ONLY = sys.argv[3].split(",") if len(sys.argv) > 3 else None          # e.g. "1" = halo kernel only
buf = E.f32_to_split32(E.Act(torch.randn(1, H, W, 192, device=dev)))
nxt = E.Act.empty(1, H, W, 192, dev, 1)
tot = 0
for c, (cin, cout) in enumerate([(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]):
    pc = E.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.zeros(cout), None, 1, 1, dev, precision="f16x3")
    out = buf.slice(64 + 32 * c, 32) if c < 4 else nxt.slice(0, 64)
    for tn in ((32, 64, 1) if cout == 32 else (64, 1)):
        if ONLY and str(tn) not in ONLY:
            continue
        tm = 1 if tn == 1 else 128
        for _ in range(2):
            E.conv(pc, buf.slice(0, cin), out, act_slope=0.2, tile_n=32 if tn == 1 else tn, tile_m=tm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            E.conv(pc, buf.slice(0, cin), out, act_slope=0.2, tile_n=32 if tn == 1 else tn, tile_m=tm)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = pc.flops_per_pixel * H * W
        print(f"{H}x{W} conv{c+1} {cin:3d}->{cout:2d} tile_n={tn:3d} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TFLOP/s alg  "
              f"{ms*1e3/(H*W/2**20):7.1f} us/Mpx", flush=True)
