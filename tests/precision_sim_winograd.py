"""Numerical gate of a Winograd F(2x2, 3x3) conv in the split-binary16 format (build container, CPU
only; not collected by pytest).  Every 3x3 / stride-1 conv with cin >= 64 of the RetinaFace oracle is replaced by an
emulation of the kernel that would run it:

    U = G g G^T       offline, float64 -> per-filter power-of-two scale -> hi + lo binary16 (round to nearest)
    V = B^T d B       float32 adds on the decoded activations -> hi + lo binary16 (truncation, like cvt_pkrtz)
    M[xi] = sum_c (Uh Vh + Uh Vl + Ul Vh)    float32 accumulation (the f16 MFMA)
    Y = A^T M A       float32

every other conv runs the 3-term split of the product path.  Reported: head error against a float64 evaluation,
landmark displacement against torch fp32, identity of the kept faces.  Gate: landmarks <= 5e-4 px, same faces.
Usage: python tests/precision_sim_winograd.py [n_images]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_crop_plus_amd import weights
from oracle import retinaface_ref as R

torch.set_num_threads(8)
sd = weights.generate_state_dict("retinaface")
orig_conv = F.conv2d

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def trunc13(x):
    return (x.contiguous().view(torch.int32) & ~0x1FFF).view(torch.float32)


def split_act(x):                       # cvt_pkrtz hi, exact remainder, cvt_pkrtz lo; binary16 range ignored (|x| << 65504)
    h = trunc13(x)
    l = trunc13(x - h)
    l = torch.where(l.abs() < 2.0 ** -24, torch.zeros_like(l), l)
    return h, l


def split_w(w):
    h = w.half().float()
    return h, (w - h).half().float()


def conv3(x, w, b, stride, padding):
    s = torch.exp2(torch.floor(torch.log2(w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30))))
    wh, wl = split_w(w / s)
    xh, xl = split_act(x)
    out = orig_conv(xh, wh, None, stride, padding) + orig_conv(xh, wl, None, stride, padding) + orig_conv(xl, wh, None, stride, padding)
    out = out * s.view(1, -1, 1, 1)
    return out if b is None else out + b.view(1, -1, 1, 1)


STATS = {"wino": 0, "direct": 0}


def conv_wino(x, w, b):
    n, c, h, wd = x.shape
    k = w.shape[0]
    assert h % 2 == 0 and wd % 2 == 0
    u = torch.einsum("ij,kcjl,ml->kcim", G, w.double(), G)                      # (K, C, 4, 4) float64
    s = torch.exp2(torch.floor(torch.log2(u.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30))))
    uh, ul = split_w((u / s).float())
    xp = F.pad(x, (1, 1, 1, 1))
    th, tw = h // 2, wd // 2
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                       # (n, c, th, tw, 4, 4)
    # B^T d B as two passes of float32 adds (rows then columns), the order a kernel would use
    r0 = d[..., 0, :] - d[..., 2, :]; r1 = d[..., 1, :] + d[..., 2, :]; r2 = d[..., 2, :] - d[..., 1, :]; r3 = d[..., 1, :] - d[..., 3, :]
    rows = torch.stack([r0, r1, r2, r3], -2)
    c0 = rows[..., 0] - rows[..., 2]; c1 = rows[..., 1] + rows[..., 2]; c2 = rows[..., 2] - rows[..., 1]; c3_ = rows[..., 1] - rows[..., 3]
    v = torch.stack([c0, c1, c2, c3_], -1)                                       # (n, c, th, tw, 4, 4)
    vh, vl = split_act(v)
    m = (torch.einsum("kcij,ncyxij->nkyxij", uh, vh) + torch.einsum("kcij,ncyxij->nkyxij", uh, vl)
         + torch.einsum("kcij,ncyxij->nkyxij", ul, vh))
    m = m * s.view(1, k, 1, 1, 1, 1).float()
    y = torch.einsum("ai,nkyxij,bj->nkyxab", AT, m, AT)                          # (n, k, th, tw, 2, 2)
    out = y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, h, wd)
    return out if b is None else out + b.view(1, -1, 1, 1)


def make(wino):
    def conv(x, w, b=None, stride=1, padding=0, *a, **kw):
        if wino and w.shape[2:] == (3, 3) and stride == 1 and padding == 1 and w.shape[1] >= 64:
            STATS["wino"] += 1
            return conv_wino(x, w, b)
        STATS["direct"] += 1
        return conv3(x, w, b, stride, padding)
    return conv


nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
worst = {}
with torch.no_grad():
    for i in range(nimg):
        img = torch.randint(0, 256, (1, 3, 640, 640)).float()
        ref_lm, ref_idx, ex = R.predict(img, sd, "all", 0.6, return_all=True)
        x = R.preprocess(img)
        sd64 = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
        raw64 = R.forward_raw(x.double(), sd64)
        raw32 = R.forward_raw(x, sd)
        print(f"image {i}: {len(ref_idx)} faces; fp32 torch vs fp64 head err", [f"{(a.double() - b).abs().max().item():.2e}" for a, b in zip(raw32, raw64)], flush=True)
        for name, wino in (("3-term direct (product path)", False), ("Winograd F(2x2,3x3) on the 3x3/s1 convs", True)):
            F.conv2d = make(wino)
            try:
                raw = R.forward_raw(x, sd)
                lm, idx, ex2 = R.predict(img, sd, "all", 0.6, return_all=True)
            finally:
                F.conv2d = orig_conv
            herr = [(a.double() - b).abs().max().item() for a, b in zip(raw, raw64)]
            same = idx == ref_idx and ex2["sel"] == ex["sel"]
            lerr = float(np.abs(lm - ref_lm).max()) if same and len(lm) else float("nan")
            # dense landmark displacement over every prior above threshold (independent of NMS)
            mask = ex["scores"] > 0.6
            dense = float(np.abs(ex2["landms"][mask] - ex["landms"][mask]).max())
            dprob = float(np.abs(ex2["scores"][mask] - ex["scores"][mask]).max())
            print(f"  {name}: head err vs fp64 {[f'{e:.2e}' for e in herr]}  same faces {same}  landmark err {lerr:.2e} px  "
                  f"dense landmark err {dense:.2e} px  d prob {dprob:.1e}  convs {STATS}", flush=True)
            worst[name] = max(worst.get(name, 0.0), dense if same else float("inf"))
            STATS.update(wino=0, direct=0)
print("worst dense landmark displacement:", {k: f"{v:.2e}" for k, v in worst.items()}, " gate 5e-4 px")
