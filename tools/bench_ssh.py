"""SSH 'ab' conv (3x3 256 -> 192: conv3X3 | conv5X5_1 of a level, _layers.py:98-125) per tile choice — A/B helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for h in (80, 40, 20):
    pc = E.pack_conv(torch.randn(192, 256, 3, 3) / (256 * 9) ** 0.5, torch.zeros(192), None, 1, 1, dev, precision="f16x3")
    x = E.f32_to_split32(E.Act(torch.randn(B, h, h, 256, device=dev).relu()))
    s = E.Act.empty(B, h, h, 384, dev, 1)
    ref = None
    for tm, tn, bal in ((128, 64, False), (128, 128, False), (256, 128, False), (256, 192, False), (256, 192, True), (256, 256, False)):
        out = s.slice(0, 192)
        f = lambda: E.conv(pc, x, out, act_slope=0.0, tile_m=tm, tile_n=tn, balance_tail=bal)
        f(); f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        cur = s.buf[..., :192].clone()
        same = True if ref is None else torch.equal(ref, cur)
        ref = cur if ref is None else ref
        print(f"{h}x{h} tile ({tm},{tn}) balanced={bal}: {us:8.1f} us {pc.flops_per_pixel * B * h * h / us / 1e6:6.1f} TF/s same_bits={same}", flush=True)
