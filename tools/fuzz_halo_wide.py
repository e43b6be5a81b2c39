"""Race hunt for the wide halo-tile kernel: random 3x3 shapes with 33..128 filters, every launch repeated several times with
other kernels in between; all results must equal the implicit-GEMM tile's bits.  python tools/fuzz_halo_wide.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for case in range(cases):
    cin = 64 * ri(1, 6)
    cout = 8 * ri(5, 16)
    n, h, w = ri(1, 4), ri(3, 150), ri(3, 200)
    up2 = ri(0, 4) == 0
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    pc = E.pack_conv(wt, torch.randn(cout, generator=g) * 0.1, None, 1, 1, dev, precision="f16x3")
    xa = E.f32_to_split32(E.Act(torch.randn(n, h, w, cin, generator=g).to(dev)))
    fmt = 1 if cout % 32 == 0 and ri(0, 1) else 0
    tn = 64 if cout <= 64 else 128
    base = E.conv(pc, xa, act_slope=0.2, out_fmt=fmt, tile_m=128, tile_n=64, in_up2=up2).buf.clone()
    ok = True
    for rep in range(4):
        o = E.conv(pc, xa, act_slope=0.2, out_fmt=fmt, tile_m=1, tile_n=tn, in_up2=up2)
        if rep & 1:
            E.conv(pc, xa, act_slope=0.2, out_fmt=fmt, tile_m=128, tile_n=128, in_up2=up2)     # something else on the device in between
        ok &= bool(torch.equal(o.buf, base))
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: n={n} h={h} w={w} cin={cin} cout={cout} up2={up2} fmt={fmt}", flush=True)
print(f"{cases} cases, {bad} bad")
