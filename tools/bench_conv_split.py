"""Microbenchmark of the f16x3 conv with split32 activations (DMA vs register-staged kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
SHAPES = {"l3c2": (64, 40, 256, 256, 3, 64), "l3c2_128": (64, 40, 256, 256, 3, 128), "merge1": (64, 80, 256, 256, 3, 128),
          "l1c3": (64, 160, 64, 256, 1, 64), "l2c2": (64, 80, 128, 128, 3, 64), "l3c1": (64, 40, 1024, 256, 1, 64),
          "l2c3": (64, 80, 128, 512, 1, 128), "l4c2": (64, 20, 512, 512, 3, 128)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
with E.default_precision("f16x3"):
    for nm in names:
        b, h, cin, cout, k, tn = SHAPES[nm]
        x = E.f32_to_split32(E.Act(torch.randn(b, h, h, cin, device=dev)))
        pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, 1, k // 2, dev)
        out = E.conv(pc, x, act_slope=0.0, tile_n=tn, out_fmt=1)
        for _ in range(2):
            E.conv(pc, x, out, act_slope=0.0, tile_n=tn)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            E.conv(pc, x, out, act_slope=0.0, tile_n=tn)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = pc.flops_per_pixel * b * out.h * out.w
        print(f"dma={os.environ.get('FCP_CONV_DMA','2')} {nm:10s} {ms:8.3f} ms {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
