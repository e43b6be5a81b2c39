"""Run a few named conv layers (split32 in/out, fp16x3) a few times each — target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
CASES = {"l3c1": (64, 40, 1024, 256, 1), "l4c1": (64, 20, 2048, 512, 1), "l3c2": (64, 40, 256, 256, 3),
         "l3c3": (64, 40, 256, 1024, 1), "l1c3": (64, 160, 64, 256, 1), "l2c1": (64, 80, 512, 128, 1),
         "merge": (64, 80, 256, 256, 3)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(CASES)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for nm in names:
    b, h, cin, cout, k = CASES[nm]
    pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, 1, k // 2, dev, precision="f16x3")
    x = E.f32_to_split32(E.Act(torch.randn(b, h, h, cin, device=dev).relu()))
    out = E.Act.empty(b, h, h, cout, dev, 1)
    res = E.f32_to_split32(E.Act(torch.randn(b, h, h, cout, device=dev))) if nm.endswith("c3") else None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    E.conv(pc, x, out, act_slope=0.0, res1=res)
    e0.record()
    for _ in range(reps):
        E.conv(pc, x, out, act_slope=0.0, res1=res)
    e1.record(); torch.cuda.synchronize()
    print(f"{nm} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)
