import sys, torch
sys.path.insert(0, "/root/repo")
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
n, h, w, cin, cout, k = 1, 16, 16, 64, 256, 1
wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
x = torch.randn(n, cin, h, w, generator=g)
pc = E.pack_conv(wt, torch.zeros(cout), None, 1, 0, dev, precision="f16x3")
xa = E.f32_to_split32(E.Act(x.permute(0, 2, 3, 1).contiguous().to(dev)))
base = E.conv(pc, xa, act_slope=1.0, out_fmt=0, tile_m=128, tile_n=128).buf[0].reshape(-1, cout).cpu()
o = E.conv(pc, xa, act_slope=1.0, out_fmt=0, tile_m=256, tile_n=256).buf[0].reshape(-1, cout).cpu()
eq = (base == o)
print("channels fully equal:", [c for c in range(64) if bool(eq[:, c].all())])
print("pixels fully equal:", [p for p in range(64) if bool(eq[p].all())][:20])
# where does o[p, c] come from in base?
for (p_, c_) in ((0, 1), (0, 4), (0, 8), (1, 0), (33, 1), (5, 9)):
    val = o[p_, c_]
    hit = (base == val).nonzero()
    print((p_, c_), "found at", hit[:4].tolist())
for p_ in (0, 5, 40):
    mp = []
    for c_ in range(32):
        hit = (base[p_] == o[p_, c_]).nonzero().flatten().tolist()
        mp.append(hit[0] if hit else -1)
    print("pixel", p_, "o channel c holds base channel:", mp)
# and across pixels for channel 1
print("o[:8,1] found at (pixel, channel):", [ (base == o[q, 1]).nonzero()[:1].tolist() for q in range(8)])
ws = pc.wscale.cpu()[:cout]
bu, ou = base / ws, o / ws
for (p_, c_) in ((0, 1), (0, 2), (0, 3), (0, 5), (0, 8), (0, 9), (0, 12), (0, 16), (7, 1), (7, 8)):
    hit = (bu == ou[p_, c_]).nonzero()[:3].tolist()
    print("o_unscaled", (p_, c_), "== base_unscaled at", hit)
