"""Micro-benchmark of the conv engine on RetinaFace / BiSeNet / RRDB layer shapes.
Usage: python tools/bench_conv.py [batch] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 640
dev = torch.device("cuda:0")
# (name, cin, cout, k, stride, h_in divisor)
SHAPES = [
    ("l1.conv1 64->64 1x1", 64, 64, 1, 1, 4), ("l1.conv2 64->64 3x3", 64, 64, 3, 1, 4),
    ("l1.conv3 64->256 1x1", 64, 256, 1, 1, 4), ("l1.c1 256->64 1x1", 256, 64, 1, 1, 4),
    ("l2.conv2 128->128 3x3", 128, 128, 3, 1, 8), ("l2.conv3 128->512 1x1", 128, 512, 1, 1, 8),
    ("l2.c1 512->128 1x1", 512, 128, 1, 1, 8),
    ("l3.conv2 256->256 3x3", 256, 256, 3, 1, 16), ("l3.conv3 256->1024", 256, 1024, 1, 1, 16),
    ("l3.c1 1024->256", 1024, 256, 1, 1, 16),
    ("l4.conv2 512->512 3x3", 512, 512, 3, 1, 32), ("l4.conv3 512->2048", 512, 2048, 1, 1, 32),
    ("fpn.merge1 256->256 3x3", 256, 256, 3, 1, 8), ("ssh 256->192 3x3", 256, 192, 3, 1, 8),
    ("head 256->32 1x1", 256, 32, 1, 1, 8),
    ("rrdb 192->64 3x3", 192, 64, 3, 1, 4), ("rrdb 64->32 3x3", 64, 32, 3, 1, 4),
]
print(f"batch={batch} size={size}")
for name, cin, cout, k, stride, div in SHAPES:
    h = size // div
    x = E.Act(torch.randn(batch, h, h, cin, device=dev))
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    pc = E.pack_conv(w, torch.zeros(cout), None, stride, k // 2, dev)
    out = E.conv(pc, x, act_slope=0.0)
    for tn in ([32] if cout <= 32 else [64, 128]):
        for _ in range(2):
            E.conv(pc, x, out, act_slope=0.0, tile_n=tn)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            E.conv(pc, x, out, act_slope=0.0, tile_n=tn)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = pc.flops_per_pixel * batch * out.h * out.w
        print(f"{name:28s} tile_n={tn:3d} M={batch*out.h*out.w:8d} {ms:8.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s")
