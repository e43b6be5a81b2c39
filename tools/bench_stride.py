"""Does a power-of-two pixel stride hurt the 1x1 convs with many input channels?  Time the same conv with
the input as a channel slice of buffers of different widths (in_ld) — profiling helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
CASES = {"l3c1": (64, 40, 1024, 256), "l4c1": (64, 20, 2048, 512), "l3c3": (64, 40, 256, 1024), "l2c1": (64, 80, 512, 128)}
for nm, (b, h, cin, cout) in CASES.items():
    pc = E.pack_conv(torch.randn(cout, cin, 1, 1) / cin ** 0.5, torch.zeros(cout), None, 1, 0, dev, precision="f16x3")
    for extra in (0, 32, 64, 96, 160):
        buf = E.Act.empty(b, h, h, cin + extra, dev, 1)
        buf.buf.zero_()
        x = buf.slice(0, cin)
        for oextra in ((0, 32) if nm == "l3c3" else (0,)):
            obuf = E.Act.empty(b, h, h, cout + oextra, dev, 1)
            out = obuf.slice(0, cout)
            for _ in range(3):
                E.conv(pc, x, out, act_slope=0.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                E.conv(pc, x, out, act_slope=0.0)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            fl = pc.flops_per_pixel * b * h * h
            print(f"{nm} in_ld={cin + extra:5d} out_ld={cout + oextra:5d} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
