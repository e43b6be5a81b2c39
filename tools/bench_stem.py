"""Fused stem + pool kernel vs the three separate launches — profiling helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
img = torch.randint(0, 256, (64, 640, 640, 3), dtype=torch.uint8, device=dev)
wt = torch.randn(64, 3, 7, 7) / 12
bn = {"weight": torch.ones(64), "bias": torch.zeros(64), "running_mean": torch.zeros(64), "running_var": torch.ones(64)}
ps = E.pack_stem_fused(wt, bn, dev)
pc = E.pack_conv(wt, None, bn, 2, 3, dev, precision="f16x3")
cat = E.Act.empty(64, 160, 160, 128, dev, 1)
def t(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("fused   %.1f us" % t(lambda: E.stem_relu_pool_u8(ps, img, cat.slice(64, 64))))
print("separate %.1f us" % t(lambda: E.maxpool3x3s2(E.conv(pc, E.u8_to_nhwc4(img, sub=(123.0, 117.0, 104.0)), act_slope=0.0, out_fmt=1), cat.slice(64, 64))))
