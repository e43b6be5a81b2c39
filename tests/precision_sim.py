"""Numerical study behind the fp16x3 conv path (build container, CPU only; not collected by pytest):
emulates split-precision convolutions inside the RetinaFace oracle and reports head / landmark errors
against an fp64 evaluation.  Usage: python tests/precision_sim.py"""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from face_crop_plus_amd import weights
from oracle import retinaface_ref as R
sd = weights.generate_state_dict('retinaface')
torch.manual_seed(0)
img = torch.randint(0, 256, (1, 3, 640, 640)).float()
orig_conv = F.conv2d
def split(x, dt):
    h = x.to(dt).float(); l = (x - h).to(dt).float(); return h, l
def make(dt, terms, wscale=True):
    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        # per-output-channel power-of-two scaling of weights so fp16 lo parts do not underflow
        if wscale:
            s = torch.exp2(torch.floor(torch.log2(w.abs().amax(dim=(1,2,3), keepdim=True).clamp_min(1e-30))))
        else:
            s = torch.ones(w.shape[0],1,1,1)
        ws = w / s
        xs_scale = torch.exp2(torch.floor(torch.log2(x.abs().max().clamp_min(1e-30))))
        xs = x / xs_scale
        xh, xl = split(xs, dt); wh, wl = split(ws, dt)
        out = orig_conv(xh, wh, None, stride, padding)
        if terms >= 3:
            out = out + orig_conv(xh, wl, None, stride, padding) + orig_conv(xl, wh, None, stride, padding)
        if terms >= 4:
            out = out + orig_conv(xl, wl, None, stride, padding)
        out = out * (s.view(1,-1,1,1) * xs_scale)
        if b is not None: out = out + b.view(1,-1,1,1)
        return out
    return conv
with torch.no_grad():
    ref_lm, ref_idx, ex = R.predict(img, sd, 'all', 0.6, return_all=True)
    x = R.preprocess(img)
    ref_raw = [t.double() for t in R.forward_raw(x, sd)]
    # float64 truth
    sd64 = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
    raw64 = R.forward_raw(x.double(), sd64)
    print('fp32 torch vs fp64: head err', [ (a-b).abs().max().item() for a,b in zip(ref_raw, raw64)])
    for dt, terms in ((torch.float16, 1), (torch.float16, 3), (torch.float16, 4), (torch.bfloat16, 3), (torch.bfloat16, 4)):
        F.conv2d = make(dt, terms)
        try:
            raw = R.forward_raw(x, sd)
            lm, idx, ex2 = R.predict(img, sd, 'all', 0.6, return_all=True)
        finally:
            F.conv2d = orig_conv
        herr = [(a.double()-b).abs().max().item() for a,b in zip(raw, raw64)]
        same = idx == ref_idx
        lerr = np.abs(lm - ref_lm).max() if same and len(lm) else None
        print(dt, terms, 'head err vs fp64', herr, 'same faces', same, 'landmark err px', lerr, 'nfaces', len(idx))

print("---- scaling variants (fp16, 3 terms) ----")
def make2(xscale, wscale):
    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        s = torch.exp2(torch.floor(torch.log2(w.abs().amax(dim=(1,2,3), keepdim=True).clamp_min(1e-30)))) if wscale else torch.ones(w.shape[0],1,1,1)
        ws = w / s
        xs = x * xscale
        xh, xl = split(xs, torch.float16); wh, wl = split(ws, torch.float16)
        out = orig_conv(xh, wh, None, stride, padding) + orig_conv(xh, wl, None, stride, padding) + orig_conv(xl, wh, None, stride, padding)
        out = out * (s.view(1,-1,1,1) / xscale)
        if b is not None: out = out + b.view(1,-1,1,1)
        return out
    return conv
with torch.no_grad():
    for xscale, wscale in ((1.0, True), (1.0, False), (16.0, True), (1/16.0, True)):
        F.conv2d = make2(xscale, wscale)
        try:
            raw = R.forward_raw(x, sd)
            lm, idx, _ = R.predict(img, sd, 'all', 0.6, return_all=True)
        finally:
            F.conv2d = orig_conv
        herr = [(a.double()-b).abs().max().item() for a,b in zip(raw, raw64)]
        print('xscale', xscale, 'wscale', wscale, 'head err', herr, 'lm err', np.abs(lm-ref_lm).max() if idx==ref_idx else 'faces differ')

print("---- rounding-mode variants (fp16 3 terms, weight pow2 scaling) ----")
def trunc13(x):
    return (x.view(torch.int32) & ~0x1FFF).view(torch.float32)
def split_mode(x, hi_rtz, lo_rtz):
    h = trunc13(x.contiguous()) if hi_rtz else x.half().float()
    r = x - h
    l = trunc13(r.contiguous()) if lo_rtz else r.half().float()
    # fp16 subnormal flush approximation: values below 2^-24 vanish
    l = torch.where(l.abs() < 2.0**-24, torch.zeros_like(l), l)
    return h, l
def make3(hi_rtz, lo_rtz):
    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        s = torch.exp2(torch.floor(torch.log2(w.abs().amax(dim=(1,2,3), keepdim=True).clamp_min(1e-30))))
        wh, wl = split_mode(w / s, False, False)     # weights split offline with RNE
        xh, xl = split_mode(x, hi_rtz, lo_rtz)
        out = orig_conv(xh, wh, None, stride, padding) + orig_conv(xh, wl, None, stride, padding) + orig_conv(xl, wh, None, stride, padding)
        out = out * s.view(1,-1,1,1)
        if b is not None: out = out + b.view(1,-1,1,1)
        return out
    return conv
with torch.no_grad():
    for hr, lr in ((False, False), (True, False), (True, True)):
        F.conv2d = make3(hr, lr)
        try:
            raw = R.forward_raw(x, sd)
            lm, idx, _ = R.predict(img, sd, 'all', 0.6, return_all=True)
        finally:
            F.conv2d = orig_conv
        herr = [(a.double()-b).abs().max().item() for a,b in zip(raw, raw64)]
        print('hi_rtz', hr, 'lo_rtz', lr, 'head err', herr, 'lm err', np.abs(lm-ref_lm).max() if idx==ref_idx else 'faces differ')
