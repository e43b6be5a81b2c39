"""Generate golden vectors from the REFERENCE itself (build container only).

    python tests/golden/make_golden.py

Imports the reference's model modules by file path from /root/reference (see
_ref_loader.py), loads the build's deterministic generated weights into them and
records inputs + outputs of the reference's own functions as small ``.npz``
fixtures next to this script.  The fixtures are data only; neither this script
nor the fixtures contain reference source.  On the GPU box /root/reference does
not exist: tests read only the committed ``.npz`` files.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from _ref_loader import load_reference_models  # noqa: E402
from face_crop_plus_amd import weights  # noqa: E402


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].numpy()).tobytes())
    return h.hexdigest()


def arr_digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden_priors(ref):
    out = {}
    for (h, w) in [(64, 96), (100, 75)]:
        out[f"priors_{h}x{w}"] = ref.PriorBox((h, w)).forward().numpy()
    for (h, w) in [(640, 640), (1024, 1024), (576, 1024)]:
        p = ref.PriorBox((h, w)).forward().numpy()
        out[f"digest_{h}x{w}"] = np.array(arr_digest(p))
        out[f"count_{h}x{w}"] = np.array(p.shape[0])
    np.savez_compressed(os.path.join(HERE, "retina_priors.npz"), **out)


def golden_postprocess(ref):
    """decode_* + filter_preds + take_by_strategy on synthetic head outputs, incl.
    engineered score ties, duplicate boxes and an IoU pinned at the 0.4 boundary."""
    g = torch.Generator().manual_seed(11)
    h, w = 64, 96
    priors = ref.PriorBox((h, w)).forward()
    P = priors.shape[0]
    n = 3
    loc = torch.randn(n, P, 4, generator=g) * 1.5
    ldm = torch.randn(n, P, 10, generator=g) * 2.0
    logits = torch.randn(n, P, 2, generator=g) * 2.0
    # ties: copy logits of some priors onto others (identical scores)
    logits[0, 10:20] = logits[0, 30:40]
    logits[1, 5] = logits[1, 100]
    logits[2, :, 1] -= 6.0     # image 2: (almost) no faces
    logits[2, 7, 1] += 12.0
    # duplicate boxes with different scores
    loc[0, 50:55] = loc[0, 50]
    scores = torch.softmax(logits, -1)
    m = ref.RetinaFace("all", 0.6)
    boxes = m.decode_bboxes(loc, priors) * torch.tensor([w, h] * 2)
    landms = m.decode_landms(ldm, priors) * torch.tensor([w, h] * 5)
    out = dict(h=np.array(h), w=np.array(w), loc=loc.numpy(), ldm=ldm.numpy(), logits=logits.numpy(),
               scores=scores[..., 1].numpy(), boxes=boxes.numpy(), landms=landms.numpy())
    fl, fb, sidx = m.filter_preds(scores[..., 1], boxes, landms)
    out.update(filt_landms=fl.numpy(), filt_boxes=fb.numpy(), filt_idx=np.array(sidx, np.int64))
    for strat in ("all", "best", "largest"):
        m.strategy = strat
        lm, idx = m.take_by_strategy(fl, fb, sidx)
        out[f"{strat}_landms"] = lm.numpy()
        out[f"{strat}_idx"] = np.array(idx, np.int64)
    np.savez_compressed(os.path.join(HERE, "retina_postprocess.npz"), **out)

    # direct NMS stress: many heavily-overlapping boxes, explicit ties, IoU exactly at threshold
    g = torch.Generator().manual_seed(12)
    n, P = 2, 1500
    cxy = torch.rand(n, P, 2, generator=g) * 200
    wh = torch.rand(n, P, 2, generator=g) * 60 + 4
    bx = torch.cat([cxy - wh / 2, cxy + wh / 2], -1)
    sc = torch.rand(n, P, generator=g) * 0.5 + 0.5
    sc[0, 100:140] = sc[0, 100]                       # 40-way tie
    bx[0, 200:204] = torch.tensor([10.0, 10.0, 19.0, 19.0])          # identical boxes
    # IoU == 0.4 exactly: boxes (0,0,9,9) area 100 and (0,0,9,3)? inter=40, union=100 -> 0.4
    bx[1, 0] = torch.tensor([300.0, 300.0, 309.0, 309.0]); sc[1, 0] = 0.999
    bx[1, 1] = torch.tensor([300.0, 300.0, 309.0, 303.0]); sc[1, 1] = 0.998
    lmk = torch.rand(n, P, 10, generator=g) * 200
    m = ref.RetinaFace("all", 0.6)
    fl, fb, sidx = m.filter_preds(sc, bx, lmk)
    out = dict(scores=sc.numpy(), boxes=bx.numpy(), landms=lmk.numpy(), filt_landms=fl.numpy(),
               filt_boxes=fb.numpy(), filt_idx=np.array(sidx, np.int64))
    for strat in ("all", "best", "largest"):
        m.strategy = strat
        lm, idx = m.take_by_strategy(fl, fb, sidx)
        out[f"{strat}_landms"] = lm.numpy()
        out[f"{strat}_idx"] = np.array(idx, np.int64)
    np.savez_compressed(os.path.join(HERE, "retina_nms.npz"), **out)


def golden_retina_full(ref):
    sd = weights.generate_state_dict("retinaface")
    m = ref.RetinaFace("all", 0.6)
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(21)
    img = torch.randint(0, 256, (2, 128, 160, 3), generator=g, dtype=torch.uint8)
    x = img.permute(0, 3, 1, 2).float()
    out = dict(image=img.numpy(), sd_digest=np.array(sd_digest(sd)))
    with torch.no_grad():
        xin = x[:, [2, 1, 0]] - torch.tensor([104, 117, 123]).view(3, 1, 1)
        s, b, l = m(xin)
        out.update(prob=s.numpy(), loc=b.numpy(), ldm=l.numpy())
        for strat, thr in (("all", 0.6), ("best", 0.6), ("largest", 0.6), ("all", 0.5)):
            m.strategy, m.vis_threshold = strat, thr
            lm, idx = m.predict(x)
            out[f"pred_{strat}_{thr}_landmarks"] = lm
            out[f"pred_{strat}_{thr}_indices"] = np.array(idx, np.int64)
    np.savez_compressed(os.path.join(HERE, "retina_full.npz"), **out)


def golden_bisenet(ref):
    sd = weights.generate_state_dict("bisenet")
    attr = {"hair_and_hat": [17, 14], "no_cloth": [-16], "hair_only": [17, -14], "neck": [12]}
    mask = {"hair": [17], "neck_or_hat": [12, 14], "eyes": [4, 5]}
    m = ref.BiSeNet(attr, mask, 2)
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(31)
    faces = torch.randint(0, 256, (3, 64, 64, 3), generator=g, dtype=torch.uint8)
    faces[1, :, :32] //= 4            # darker half: a little variety between faces
    x = faces.permute(0, 3, 1, 2).float()
    out = dict(faces=faces.numpy(), sd_digest=np.array(sd_digest(sd)))
    with torch.no_grad():
        import torch.nn.functional as F
        mean = torch.tensor(m.mean).view(1, 3, 1, 1)
        std = torch.tensor(m.std).view(1, 3, 1, 1)
        xin = (F.interpolate(x.div(255), (512, 512), mode="bilinear") - mean) / std
        o = F.interpolate(m(xin), (64, 64), mode="nearest")
        t2 = torch.topk(o, 2, dim=1).values
        out["top2_gap"] = (t2[:, 0] - t2[:, 1]).numpy()      # margin of every label decision
        out["labels"] = o.argmax(1).numpy().astype(np.uint8)
        ag, mg = m.predict(x)
    out["attr_keys"] = np.array(sorted(ag.keys()))
    for k, v in ag.items():
        out[f"attr_{k}"] = np.array(v, np.int64)
    out["mask_keys"] = np.array(sorted(mg.keys()))
    for k, (idx, masks) in mg.items():
        out[f"mask_{k}_idx"] = np.array(idx, np.int64)
        out[f"mask_{k}"] = masks
    np.savez_compressed(os.path.join(HERE, "bisenet.npz"), **out)
    print("bisenet groups:", {k: v for k, v in ag.items()}, {k: v[0] for k, v in mg.items()},
          np.bincount(out["labels"].ravel(), minlength=19))


def golden_rrdb(ref):
    sd = weights.generate_state_dict("rrdb")
    m = ref.RRDBNet(0.02)
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(41)
    img = torch.randint(0, 256, (3, 20, 24, 3), generator=g, dtype=torch.uint8)
    x = img.permute(0, 3, 1, 2).float()
    # image 0: small face (enhanced), image 1: large face (kept), image 2: no landmarks (kept)
    lm = np.zeros((2, 5, 2), np.float32)
    lm[0, 0] = [5, 5]; lm[0, 4] = [7, 8]
    lm[1, 0] = [2, 2]; lm[1, 4] = [20, 18]
    idx = [0, 1]
    out = dict(image=img.numpy(), landmarks=lm, indices=np.array(idx), sd_digest=np.array(sd_digest(sd)))
    with torch.no_grad():
        out["x4_image0"] = m(x[:1].div(255)).numpy()
        res = m.predict(x.clone(), lm, idx)
        out["pred"] = res.numpy()
        res_all = m.predict(x.clone(), None, None)
        out["pred_all"] = res_all.numpy()
    np.savez_compressed(os.path.join(HERE, "rrdb.npz"), **out)
    print("rrdb changed images:", [bool((res[i] != x[i]).any()) for i in range(3)])


def main():
    ref = load_reference_models()
    torch.manual_seed(0)
    golden_priors(ref)
    golden_postprocess(ref)
    golden_retina_full(ref)
    golden_bisenet(ref)
    golden_rrdb(ref)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
