"""Numerical gate for a cheaper split product (build container, CPU only; not collected by pytest): the two CORRECTION terms of
the fp16x3 product, ah*bl + al*bh, are ~2^-11 of the result — could they run on the fp8 matrix instructions (twice the f16 rate
on gfx950: 3 MFMA units of work per product would become 2)?  An fp8 MFMA takes fp8 on BOTH sides, so the hi operand of a
correction term is rounded to fp8 as well.  Emulation inside the RetinaFace oracle: main term ah*bh on binary16 operands, the
corrections on operands rounded to float8 (e4m3 / e5m2) with a per-tensor power-of-two scale, fp32 accumulation; head and
landmark errors against an fp64 evaluation, like tests/precision_sim.py.   python tests/precision_sim_fp8.py"""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from face_crop_plus_amd import weights
from oracle import retinaface_ref as R
sd = weights.generate_state_dict('retinaface')
torch.manual_seed(0)
img = torch.randint(0, 256, (1, 3, 640, 640)).float()
orig_conv = F.conv2d


def split16(x):
    h = x.half().float(); l = (x - h).half().float(); return h, l


def q8(x, dt):
    """round to float8 with a per-tensor power-of-two scale that puts max |x| near the top of the format's range"""
    top = 128.0 if dt == torch.float8_e4m3fn else 16384.0   # max |x| / s in [top, 2 top): below 448 (e4m3) / 57344 (e5m2)
    s = torch.exp2(torch.floor(torch.log2(x.abs().max().clamp_min(1e-30) / top)))
    return (x / s).to(dt).float() * s


def make(dt8, which):
    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        s = torch.exp2(torch.floor(torch.log2(w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30))))
        xh, xl = split16(x); wh, wl = split16(w / s)
        out = orig_conv(xh, wh, None, stride, padding)
        if which == "f16":                       # the product path: corrections on binary16 operands
            out = out + orig_conv(xh, wl, None, stride, padding) + orig_conv(xl, wh, None, stride, padding)
        elif which == "fp8":                     # corrections on float8 operands (both sides)
            out = out + orig_conv(q8(xh, dt8), q8(wl, dt8), None, stride, padding) + orig_conv(q8(xl, dt8), q8(wh, dt8), None, stride, padding)
        elif which == "none":
            pass
        out = out * s.view(1, -1, 1, 1)
        if b is not None: out = out + b.view(1, -1, 1, 1)
        return out
    return conv


with torch.no_grad():
    ref_lm, ref_idx, _ = R.predict(img, sd, 'all', 0.6, return_all=True)
    x = R.preprocess(img)
    sd64 = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
    raw64 = R.forward_raw(x.double(), sd64)
    for name, dt8, which in (("3 terms on binary16 (product)", None, "f16"), ("corrections on e4m3", torch.float8_e4m3fn, "fp8"),
                             ("corrections on e5m2", torch.float8_e5m2, "fp8"), ("no corrections (1 term)", None, "none")):
        F.conv2d = make(dt8, which)
        try:
            raw = R.forward_raw(x, sd)
            lm, idx, _ = R.predict(img, sd, 'all', 0.6, return_all=True)
        finally:
            F.conv2d = orig_conv
        herr = [float((a.double() - b).abs().max()) for a, b in zip(raw, raw64)]
        same = idx == ref_idx
        lerr = float(np.abs(lm - ref_lm).max()) if same and len(lm) else None
        print(f"{name:34s} head err vs fp64 {['%.2e' % e for e in herr]}  same faces {same}  landmarks vs torch fp32 {lerr} px  ({len(idx)} faces)", flush=True)
