"""Does producing/consuming layer-1 tensors in sub-batches (so they stay in the 256 MB Infinity Cache) pay?
Runs the bottleneck chain c1 -> c2 -> c3(+res) of layer 1 at several batch sizes and prints ms per image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with E.default_precision("f16x3"):
    pc1 = E.pack_conv(torch.randn(64, 256, 1, 1) / 16, torch.zeros(64), None, 1, 0, dev)
    pc2 = E.pack_conv(torch.randn(64, 64, 3, 3) / 24, torch.zeros(64), None, 1, 1, dev)
    pc3 = E.pack_conv(torch.randn(256, 64, 1, 1) / 8, torch.zeros(256), None, 1, 0, dev)
    for b in (2, 4, 8, 16, 64):
        x = E.f32_to_split32(E.Act(torch.randn(b, 160, 160, 256, device=dev)))
        def chain(x=x):
            y = x
            for _ in range(3):                       # three bottleneck blocks
                o = E.conv(pc1, y, act_slope=0.0, out_fmt=1)
                o = E.conv(pc2, o, act_slope=0.0, out_fmt=1)
                y = E.conv(pc3, o, act_slope=0.0, res1=y, res1_pre=True, out_fmt=1)
            return y
        ms = timeit(chain)
        print(f"batch {b:3d}: {ms:.3f} ms  {ms/b:.4f} ms/img", flush=True)
