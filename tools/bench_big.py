"""Fixed-tile timings of the 256-row kernel on the detector's layer-3/4/FPN shapes — A/B helper.
    python tools/bench_big.py [tile_n] [batch] [cu_budget]
Prints uniform vs balanced-tail schedule per shape (and checks the two give the same bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
tn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
budget = int(sys.argv[3]) if len(sys.argv) > 3 else 0
CASES = {"l3c1": (B, 40, 1024, 256, 1), "l3c2": (B, 40, 256, 256, 3), "l4c1": (B, 20, 2048, 512, 1), "l4c2": (B, 20, 512, 512, 3),
         "out1": (B, 80, 512, 256, 1), "merge1": (B, 80, 256, 256, 3), "l3b0c1": (B, 80, 512, 256, 1),
         "l3c3ds": (B, 40, 768, 1024, 1), "l4c3ds": (B, 20, 1536, 2048, 1), "out3": (B, 20, 2048, 256, 1)}
tot = [0.0, 0.0]
with E.cu_budget(budget):
    for nm, (b, h, cin, cout, k) in CASES.items():
        pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, 1, k // 2, dev, precision="f16x3")
        x = E.f32_to_split32(E.Act(torch.randn(b, h, h, cin, device=dev).relu()))
        outs, us = [], []
        for bal in (False, True):
            out = E.Act.empty(b, h, h, cout, dev, 1)
            for _ in range(2):
                E.conv(pc, x, out, act_slope=0.0, tile_m=256, tile_n=tn, balance_tail=bal)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                E.conv(pc, x, out, act_slope=0.0, tile_m=256, tile_n=tn, balance_tail=bal)
            e1.record(); torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) / 5 * 1e3)
            outs.append(out.buf.clone())
        same = torch.equal(outs[0], outs[1])
        tot[0] += us[0]; tot[1] += us[1]
        gf = pc.flops_per_pixel * b * h * h
        print(f"{nm:7s} uniform {us[0]:8.1f} us {gf / us[0] / 1e6:6.1f} TF/s | balanced {us[1]:8.1f} us {gf / us[1] / 1e6:6.1f} TF/s  same_bits={same}", flush=True)
print(f"sum     uniform {tot[0]:8.1f} us | balanced {tot[1]:8.1f} us   (tile_n {tn}, batch {B}, cu_budget {budget})")
