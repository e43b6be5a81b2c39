"""3x3 convs with 33..128 filters: the wide halo-tile kernel (tile_m 1, tile_n 64 / 128) against the other tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
SHAPES = {"layer2.conv2 128->128 @80": (64, 80, 128, 128), "ssh 64->128 @80": (64, 80, 64, 128), "bise l2 128->128 @64": (32, 64, 128, 128),
          "ssh 64->128 @40": (64, 40, 64, 128), "256->128 @80": (64, 80, 256, 128), "l1 64->64 @160": (64, 160, 64, 64),
          "rrdb conv5 192->64 @1024": (1, 1024, 192, 64), "ssh 64->64 @80": (64, 80, 64, 64)}
for nm, (b, h, cin, cout) in SHAPES.items():
    x = E.f32_to_split32(E.Act(torch.randn(b, h, h, cin, device=dev)))
    pc = E.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.zeros(cout), None, 1, 1, dev, precision="f16x3")
    outs = {}
    tiles = [(128, 128, False), (128, 64, False)] + ([(256, 128, False), (256, 128, True), (1, 128, False)] if cout > 64 else [(1, 32, False), (1, 64, False)])
    for tm, tn, bal in tiles:
        out = E.conv(pc, x, act_slope=0.0, tile_m=tm, tile_n=tn, out_fmt=1, balance_tail=bal)
        for _ in range(2):
            E.conv(pc, x, out, act_slope=0.0, tile_m=tm, tile_n=tn, balance_tail=bal)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            E.conv(pc, x, out, act_slope=0.0, tile_m=tm, tile_n=tn, balance_tail=bal)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        outs[(tm, tn, bal)] = out.buf.clone()
        print(f"{nm:28s} tile {tm:3d}x{tn:<3d}{' bal' if bal else '    '} {us:8.1f} us {pc.flops_per_pixel * b * h * h / us / 1e6:7.1f} TFLOP/s", flush=True)
    ref = outs[(128, 128, False)]
    print("   same bits:", {k: bool(torch.equal(ref, o)) for k, o in outs.items()}, flush=True)
