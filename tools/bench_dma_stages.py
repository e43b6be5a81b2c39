"""1x1 convs of the ResNet body on the LDS-DMA kernel, two vs three LDS stages (FCP_CONV_DMA=3 with a FCP_CONV_PROFILING build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
SHAPES = [(64, 40, 256, 1024, True), (64, 20, 512, 2048, True), (64, 80, 512, 128, False), (64, 40, 1024, 256, False), (64, 20, 2048, 512, False),
          (64, 80, 512, 256, True)]
for b, h, cin, cout, res in SHAPES:
    x = E.f32_to_split32(E.Act(torch.randn(b, h, h, cin, device=dev)))
    r = E.f32_to_split32(E.Act(torch.randn(b, h, h, cout, device=dev))) if res else None
    pc = E.pack_conv(torch.randn(cout, cin, 1, 1) / cin ** 0.5, torch.zeros(cout), None, 1, 0, dev, precision="f16x3")
    line = f"1x1 {cin:4d}->{cout:4d} @{h:2d}{' +res' if res else '     '} stages={os.environ.get('FCP_CONV_DMA', '2')}:"
    for tm, tn, bal in ((128, 128, False), (128, 64, False), (256, 128, False), (256, 256, False), (256, 256, True)):
        if tm == 256 and cout < tn: continue
        out = E.conv(pc, x, act_slope=0.0, tile_m=tm, tile_n=tn, out_fmt=1, res1=r, balance_tail=bal)
        for _ in range(2):
            E.conv(pc, x, out, act_slope=0.0, tile_m=tm, tile_n=tn, res1=r, balance_tail=bal)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            E.conv(pc, x, out, act_slope=0.0, tile_m=tm, tile_n=tn, res1=r, balance_tail=bal)
        e1.record(); torch.cuda.synchronize()
        line += f"  {tm}x{tn}{'b' if bal else ''}: {e0.elapsed_time(e1) / 10 * 1e3:6.1f}"
    print(line, flush=True)
