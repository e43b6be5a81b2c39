"""Randomised cross-check of the fused bottleneck forms against the separate convolutions (same bits required).
python tools/fuzz_chain.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
mk = lambda co, ci, k: E.pack_conv(torch.randn(co, ci, k, k, generator=g) * (2 / (ci * k * k)) ** 0.5, torch.randn(co, generator=g) * 0.1,
                                   None, 1, k // 2, dev, precision="f16x3")
FORMS = [(64, 256, 64, True, True), (64, 256, 128, True, True), (128, 512, 128, False, True), (128, 512, 256, False, True),
         (256, 1024, 256, False, True), (128, 256, 64, False, False)]
bad = 0
for case in range(cases):
    c, nout, cn, has_c2, residual = FORMS[case % len(FORMS)]
    n, h, w = ri(1, 4), ri(3, 50), ri(3, 50)
    pc2 = mk(c, c, 3) if has_c2 else None
    pc3, pc1 = mk(nout, c, 1), mk(cn, nout, 1)
    t = E.f32_to_split32(E.Act(torch.randn(n, h, w, c, generator=g).relu().to(dev)))
    xr = E.f32_to_split32(E.Act(torch.randn(n, h, w, nout, generator=g).relu().to(dev))) if residual else None
    o2 = E.conv(pc2, t, act_slope=0.0, out_fmt=1) if has_c2 else t
    o3 = E.conv(pc3, o2, act_slope=0.0, res1=xr, res1_pre=True, out_fmt=1)
    o1 = E.conv(pc1, o3, act_slope=0.0, out_fmt=1)
    out, t1n = E.bottleneck_chain(pc2, pc3, pc1, t, xr)                      # 128-pixel tiles, 4 waves
    out4, t1n4 = E.bottleneck_chain(pc2, pc3, pc1, t, xr, tile_m=256)        # and the 8-wave form (where supported)
    outl, t1nl = E.bottleneck_chain(pc2, pc3, pc1, t, xr, tile_m=128)        # linear 128-pixel tiles (the default of the conv2 forms is the patch form)
    if has_c2:
        out4, t1n4 = E.bottleneck_chain(pc2, pc3, pc1, t, xr, tile_m=32) if case % 2 else (out4, t1n4)   # 16 x 16 patches on 8 waves
    torch.cuda.synchronize()
    if not (torch.equal(out.buf, o3.buf) and torch.equal(t1n.buf, o1.buf) and torch.equal(out4.buf, o3.buf) and torch.equal(t1n4.buf, o1.buf)
            and torch.equal(outl.buf, o3.buf) and torch.equal(t1nl.buf, o1.buf)):
        bad += 1
        print(f"MISMATCH case {case}: form {(c, nout, cn, has_c2, residual)} n={n} h={h} w={w}", flush=True)
# the two-source pair (layer2.0: conv3 + stride-s downsample over [t (128 ch) | x(::s, ::s) (256 ch)], next conv1 512 -> 128)
for case in range(max(1, cases // 6)):
    n, h, w, st = ri(1, 3), ri(1, 40), ri(1, 40), ri(1, 2)
    hx, wx = (h - 1) * st + ri(1, st), (w - 1) * st + ri(1, st)
    pc3, pc1 = mk(512, 384, 1), mk(128, 512, 1)
    t = E.f32_to_split32(E.Act(torch.randn(n, h, w, 128, generator=g).relu().to(dev)))
    xb = E.f32_to_split32(E.Act(torch.randn(n, hx, wx, 256, generator=g).relu().to(dev)))
    o3 = E.conv(pc3, t, act_slope=0.0, out_fmt=1, x2=xb, x2_stride=st)
    o1 = E.conv(pc1, o3, act_slope=0.0, out_fmt=1)
    out, t1n = E.bottleneck_chain(None, pc3, pc1, t, None, t1b=xb, t1b_stride=st)
    torch.cuda.synchronize()
    if not (torch.equal(out.buf, o3.buf) and torch.equal(t1n.buf, o1.buf)):
        bad += 1
        print(f"MISMATCH two-source case {case}: n={n} h={h} w={w} stride={st} x {hx}x{wx}", flush=True)
print(f"{cases} cases, {bad} bad")
