"""Per-layer sensitivity of the detector to dropping ONE term of the fp16x3 product.

CPU only (build container).  Every conv of the RetinaFace oracle is emulated as the engine computes it
(operands split into binary16 hi + lo, filters pre-scaled by a power of two, fp32 accumulation of
ah*bh + ah*bl + al*bh); then, one layer at a time, that layer alone is run with a 2-term product:

  "act-hi"  a_hi*(w_hi + w_lo)      activation rounded to ONE binary16 (round to nearest), filter exact
  "w-hi"    (a_hi + a_lo)*w_hi      filter rounded to one binary16 (nearest, after the power-of-two row scale)

and the detector's outputs are compared with the all-3-term run: max landmark displacement (px), whether the
kept faces / their order change, max change of the face scores.  The adoption bar of the verdict: landmarks
<= 2e-4 px and every index unchanged.  Usage:  python tools/precision_sweep.py [size] > profiles/r02_precision_sweep.md
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from face_crop_plus_amd import weights
from oracle import retinaface_ref as R

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.set_num_threads(os.cpu_count() or 8)
sd = weights.generate_state_dict("retinaface")
torch.manual_seed(0)
img = torch.randint(0, 256, (1, 3, size, size)).float()
orig = F.conv2d


def split_rne(x):
    h = x.half().float()
    return h, (x - h).half().float()


def split_rtz(x):                     # the kernel's activation split (cvt_pkrtz): truncate to 11 significant bits
    h = (x.contiguous().view(torch.int32) & ~0x1FFF).view(torch.float32)
    r = x - h
    return h, (r.contiguous().view(torch.int32) & ~0x1FFF).view(torch.float32)


class Emu:
    def __init__(self):
        self.idx, self.special, self.mode, self.log = 0, None, None, []

    def __call__(self, x, w, b=None, stride=1, padding=0, *a, **k):
        i = self.idx
        self.idx += 1
        s = torch.exp2(torch.floor(torch.log2(w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30))))
        wh, wl = split_rne(w / s)
        if len(self.log) <= i:
            self.log.append((tuple(w.shape), stride, tuple(x.shape[2:])))
        if i == self.special and self.mode == "act-hi":
            xh = x.half().float()                                   # one binary16, round to nearest
            out = orig(xh, wh, None, stride, padding) + orig(xh, wl, None, stride, padding)
        elif i == self.special and self.mode == "w-hi":
            xh, xl = split_rtz(x)
            out = orig(xh, wh, None, stride, padding) + orig(xl, wh, None, stride, padding)
        else:
            xh, xl = split_rtz(x)
            out = orig(xh, wh, None, stride, padding) + orig(xh, wl, None, stride, padding) + orig(xl, wh, None, stride, padding)
        out = out * s.view(1, -1, 1, 1)
        return out if b is None else out + b.view(1, -1, 1, 1)


emu = Emu()


PRI = torch.from_numpy(R.prior_box(size, size))


def run(special=None, mode=None):
    """-> (kept landmarks, kept image indices, dense: (face probability (P,), decoded landmarks of ALL priors (P,10) px))"""
    emu.idx, emu.special, emu.mode = 0, special, mode
    F.conv2d = emu
    try:
        with torch.no_grad():
            lm, idx, _ = R.predict(img, sd, "all", 0.6, return_all=True)
            emu.idx = 0
            prob, box, ldm = R.forward(R.preprocess(img), sd)
    finally:
        F.conv2d = orig
    # dense landmark decode (retinaface.py:204-210): prior centre + delta * variance[0] * prior size, in pixels
    p = PRI
    d = ldm[0].view(-1, 5, 2)
    dense = (p[:, None, :2] + d * 0.1 * p[:, None, 2:]) * size
    return lm, idx, (prob[0, :, 1].numpy(), dense.reshape(-1, 10).numpy())


t0 = time.time()
lm0, idx0, ex0 = run()
n_layers = emu.idx
assert n_layers == 82, n_layers
with torch.no_grad():
    lm32, idx32, _ = R.predict(img, sd, "all", 0.6, return_all=True)
print(f"# Per-layer 2-term sensitivity of RetinaFace (oracle emulation, one {size}x{size} noise image, generated weights)\n")
print(f"All layers 3-term vs torch fp32: same faces {idx0 == idx32}, {len(idx0)} faces after NMS, {int((ex0[0] > 0.6).sum())} candidate priors, "
      f"landmark difference {np.abs(lm0 - lm32).max():.2e} px.  One forward = {time.time() - t0:.1f} s on {torch.get_num_threads()} threads.\n")
print("Bar for running a layer 2-term (VERDICT r1 item 4): landmarks move <= 2e-4 px AND kept faces / order unchanged.\n")
print("`d landmarks` = max displacement of the decoded landmarks over ALL priors whose face probability exceeds 0.6 in the 3-term run "
      "(every pyramid level), `d prob` = max change of their probability; `faces same` = the kept faces and their order after NMS.\n")
print("| # | conv (cout,cin,kh,kw) / stride @ input | act-hi: d landmarks px | d prob | faces same | w-hi: d landmarks px | d prob | faces same | qualifies |")
print("|---|---|---|---|---|---|---|---|---|")
names = []
ok_layers = []
for i in range(n_layers):
    row = []
    q = False
    for mode in ("act-hi", "w-hi"):
        lm, idx, (pr, dn) = run(i, mode)
        same = idx == idx0
        cand = ex0[0] > 0.6
        d = float(np.abs(dn[cand] - ex0[1][cand]).max())
        dp = float(np.abs(pr[cand] - ex0[0][cand]).max())
        row += [f"{d:.2e}", f"{dp:.1e}", "yes" if same else "NO"]
        q = q or (same and d <= 2e-4)
    shp, st, hw = emu.log[i]
    if q:
        ok_layers.append(i)
    print(f"| {i} | {shp} / {st} @ {hw[0]}x{hw[1]} | " + " | ".join(row) + f" | {'yes' if q else 'no'} |", flush=True)
print(f"\nLayers meeting the bar: {ok_layers if ok_layers else 'none'}  ({time.time() - t0:.0f} s total)")
