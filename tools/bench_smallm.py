"""Small-M, large-K 3x3 convs (BiSeNet layer 4 / heads at 32 faces; RetinaFace's 20x20 level): tiles and LDS stages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
SHAPES = [(32, 16, 512, 512, 3), (32, 16, 512, 128, 3), (32, 32, 256, 256, 3), (32, 32, 256, 128, 3), (64, 20, 512, 512, 3), (32, 16, 256, 512, 3)]
for b, h, cin, cout, k in SHAPES:
    x = E.f32_to_split32(E.Act(torch.randn(b, h, h, cin, device=dev)))
    pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, 1, k // 2, dev, precision="f16x3")
    line = f"{k}x{k} {cin:4d}->{cout:4d} @{h:2d} b{b} stages={os.environ.get('FCP_CONV_DMA', '2')}:"
    for tm, tn in ((128, 128), (128, 64), (256, 128), (256, 256)):
        if tm == 256 and (cout < tn or b * h * h < 256 * 64):
            continue
        out = E.conv(pc, x, act_slope=0.0, tile_m=tm, tile_n=tn, out_fmt=1)
        for _ in range(2):
            E.conv(pc, x, out, act_slope=0.0, tile_m=tm, tile_n=tn)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            E.conv(pc, x, out, act_slope=0.0, tile_m=tm, tile_n=tn)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        line += f"  {tm}x{tn}: {us:6.1f} us ({pc.flops_per_pixel * b * h * h / us / 1e6:5.0f} TF)"
    print(line, flush=True)
