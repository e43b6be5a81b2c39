"""BiSeNet parse step (32 aligned faces of 256 x 256 -> label maps + histograms) with and without the fused fp32 stem, in one
process: ms per step and whether the label maps / histograms are the same.   python tools/bench_parse.py [faces]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E, weights
from face_crop_plus_amd.bise import BiSeNet
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = BiSeNet({"glasses": [6]}, {"eyes": [4, 5]}, F).load(dev, weights.generate_state_dict("bisenet"), "f16x3")
g = torch.Generator(device="cpu").manual_seed(11)
faces = torch.randint(0, 256, (F, 256, 256, 3), generator=g, dtype=torch.uint8).to(dev)
ref = None
for rep in range(2):
    for fused in (True, False):
        m.fused_stem = fused
        E.Autotune.enabled = True
        out = m.parse(faces)
        torch.cuda.synchronize()
        E.Autotune.enabled = False
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = m.parse(faces)
        e1.record(); torch.cuda.synchronize()
        labels = out[0] if isinstance(out, (tuple, list)) else out
        if ref is None:
            ref = labels.clone()
        same = float((labels != ref).float().mean())
        print(f"fused stem {fused!s:5s}: {e0.elapsed_time(e1) / 10:7.3f} ms per parse step of {F} faces; label mismatches vs the first configuration: {same:.2e}", flush=True)
