"""Per-layer tile comparison (128x64 / 128x128 / 256x128 / 256x256), split32 in/out, fp16x3 — profiling helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
# name: (batch, h, cin, cout, k, stride, residual)
CASES = {"l1c1": (64, 160, 256, 64, 1, 1, 0), "l1c2": (64, 160, 64, 64, 3, 1, 0), "l1c3": (64, 160, 64, 256, 1, 1, 1),
         "l2c1": (64, 80, 512, 128, 1, 1, 0), "l2c2": (64, 80, 128, 128, 3, 1, 0), "l2c3": (64, 80, 128, 512, 1, 1, 1),
         "l3c1": (64, 40, 1024, 256, 1, 1, 0), "l3c2": (64, 40, 256, 256, 3, 1, 0), "l3c3": (64, 40, 256, 1024, 1, 1, 1),
         "l4c1": (64, 20, 2048, 512, 1, 1, 0), "l4c2": (64, 20, 512, 512, 3, 1, 0), "l4c3": (64, 20, 512, 2048, 1, 1, 1),
         "l3ds": (64, 80, 512, 1024, 1, 2, 0), "fpn3": (64, 20, 2048, 256, 1, 1, 0), "merge": (64, 80, 256, 256, 3, 1, 0),
         "ssh3": (64, 80, 256, 192, 3, 1, 0)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(CASES)
for nm in names:
    b, h, cin, cout, k, st, hasres = CASES[nm]
    pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, st, k // 2, dev, precision="f16x3")
    x = E.f32_to_split32(E.Act(torch.randn(b, h, h, cin, device=dev).relu()))
    ho = (h + 2 * (k // 2) - k) // st + 1
    out = E.Act.empty(b, ho, ho, cout, dev, 1)
    res = E.f32_to_split32(E.Act(torch.randn(b, ho, ho, cout, device=dev))) if hasres else None
    line = f"{nm:6s}"
    for tm, tn in ((128, 64), (128, 128), (256, 128), (256, 256)):
        if tn > cout and not (tn == 128 and cout > 64):
            line += f"  {tm}x{tn}:    -   "
            continue
        for _ in range(2):
            E.conv(pc, x, out, act_slope=0.0, res1=res, tile_m=tm, tile_n=tn)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            E.conv(pc, x, out, act_slope=0.0, res1=res, tile_m=tm, tile_n=tn)
        e1.record(); torch.cuda.synchronize()
        line += f"  {tm}x{tn}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f}"
    fl = pc.flops_per_pixel * b * ho * ho
    print(line + f"   us   ({fl / 1e9:.1f} GFLOP)", flush=True)
