"""Randomised cross-check of the fp16x3 kernel geometries: every eligible tile choice must return the same bits
(and agree with torch fp32 within the split's accuracy).  python tools/fuzz_conv.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for case in range(cases):
    k = [1, 3][ri(0, 1)]
    stride = 1 if k == 1 or ri(0, 3) else 2
    cin = 32 * ri(1, 8) if ri(0, 1) else [64, 128, 256, 512][ri(0, 3)]
    cout = [8, 24, 32, 40, 64, 96, 128, 192, 256, 320, 512][ri(0, 10)]
    n, h, w = ri(1, 3), ri(5, 70), ri(5, 70)
    res = ri(0, 2) == 0 and stride == 1
    act = [0.0, 0.2, 1.0][ri(0, 2)]
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(n, cin, h, w, generator=g)
    pc = E.pack_conv(wt, bias, None, stride, k // 2, dev, precision="f16x3")
    xa = E.f32_to_split32(E.Act(x.permute(0, 2, 3, 1).contiguous().to(dev)))
    oh, ow = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    r = torch.randn(n, cout, oh, ow, generator=g) if res else None
    ra = E.f32_to_split32(E.Act(r.permute(0, 2, 3, 1).contiguous().to(dev))) if (res and cout % 32 == 0) else (
        E.Act(r.permute(0, 2, 3, 1).contiguous().to(dev)) if res else None)
    out_fmt = 1 if cout % 32 == 0 and ri(0, 1) else 0
    m = n * oh * ow
    tiles = [(128, 64), (128, 128)] + ([(128, 32)] if cout <= 32 else [])
    if cout % 8 == 0 and cout >= 128:
        tiles += [(256, 128)] + ([(256, 256)] if cout >= 256 else []) + ([(256, 192)] if cout in (192, 320) else [])
    if k == 3 and stride == 1 and cout <= 64 and cin >= 64 and cout % 8 == 0:
        tiles += [(1, 32)]
    if k == 3 and stride == 1 and cin % 64 == 0 and cout % 8 == 0 and (32 < cout <= 64 or (64 < cout <= 128 and not res)):
        tiles += [(1, 64 if cout <= 64 else 128)]          # wide halo-tile kernel
    if cout % 8 != 0:
        continue
    # the 256-row kernel also with its balanced M-tile schedule, laid out for a small CU budget so that the few M-tiles of
    # these shapes make full rounds + a tail of shorter tiles (every tail height 32..224 turns up over the cases)
    tiles = [t + (False, 0) for t in tiles] + [t + (True, [8, 8, 16, 24, 32, 64, 0][ri(0, 6)]) for t in tiles if t[0] == 256]
    outs = []
    for tm, tn, bal, budget in tiles:
        with E.cu_budget(budget):
            o = E.conv(pc, xa, act_slope=act, res1=ra, res1_pre=True, out_fmt=out_fmt, tile_m=tm, tile_n=tn, balance_tail=bal)
        outs.append((tm, tn, o.buf.clone(), bal, budget))
    torch.cuda.synchronize()
    ref = F.conv2d(x, wt, bias, stride, k // 2)
    if res:
        ref = ref + r
    ref = torch.where(ref >= 0, ref, ref * act)
    got = E.Act(outs[0][2], fmt=out_fmt).nchw().cpu()
    err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
    same = all(torch.equal(outs[0][2], o[2]) for o in outs[1:])
    if not same or err > 2e-5:
        bad += 1
        print(f"MISMATCH case {case}: n={n} h={h} w={w} cin={cin} cout={cout} k={k} s={stride} res={res} act={act} fmt={out_fmt} "
              f"tiles={[(o[0], o[1], o[3], o[4]) for o in outs]} same={same} err={err:.2e}", flush=True)
print(f"{cases} cases, {bad} bad")
