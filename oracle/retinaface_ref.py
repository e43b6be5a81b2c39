"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the reference's RetinaFace path, written against a plain
state_dict (torch CPU fp32 for the float network, numpy for the integer /
index work).  Pinned against the reference modules themselves (loaded by file
path in the build container) by ``tests/golden/make_golden.py``; the resulting
vectors live in ``tests/golden/*.npz``.

Follows (all paths relative to /root/reference/src/face_crop_plus):
  forward            models/retinaface.py:137-144, torchvision ResNet-50 v1.5 body
  FPN / SSH / Head   models/_layers.py:64-162
  PriorBox           models/_layers.py:41-62
  decode_*           models/retinaface.py:169-178, :204-210, :455-461
  filter_preds       models/retinaface.py:263-304
  take_by_strategy   models/retinaface.py:363-408
  predict            models/retinaface.py:449-470
"""
from __future__ import annotations

from math import ceil

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding)


def _bottleneck(x, sd, p, stride):
    """torchvision Bottleneck, v1.5 (stride on the 3x3)."""
    out = F.relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1"))
    out = F.relu(_bn(_conv(out, sd, p + ".conv2", stride, 1), sd, p + ".bn2"))
    out = _bn(_conv(out, sd, p + ".conv3"), sd, p + ".bn3")
    if (p + ".downsample.0.weight") in sd:
        x = _bn(_conv(x, sd, p + ".downsample.0", stride), sd, p + ".downsample.1")
    return F.relu(out + x)


def body(x, sd):
    x = F.relu(_bn(_conv(x, sd, "body.conv1", 2, 3), sd, "body.bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, blocks in enumerate((3, 4, 6, 3), 1):
        for b in range(blocks):
            x = _bottleneck(x, sd, f"body.layer{li}.{b}", 2 if (b == 0 and li > 1) else 1)
        if li >= 2:
            feats.append(x)
    return feats


def _cb(x, sd, p, k, relu):
    # conv(k) + BN + LeakyReLU(slope 0 for 256-channel FPN/SSH == ReLU) — _layers.py:68,:103
    x = _bn(_conv(x, sd, p + ".0", 1, k // 2), sd, p + ".1")
    return F.leaky_relu(x, 0.0) if relu else x


def fpn(feats, sd):
    o1 = _cb(feats[0], sd, "fpn.output1", 1, True)
    o2 = _cb(feats[1], sd, "fpn.output2", 1, True)
    o3 = _cb(feats[2], sd, "fpn.output3", 1, True)
    up3 = F.interpolate(o3, size=o2.shape[2:], mode="nearest")
    o2 = _cb(o2 + up3, sd, "fpn.merge2", 3, True)
    up2 = F.interpolate(o2, size=o1.shape[2:], mode="nearest")
    o1 = _cb(o1 + up2, sd, "fpn.merge1", 3, True)
    return [o1, o2, o3]


def ssh(x, sd, p):
    c3 = _cb(x, sd, p + ".conv3X3", 3, False)
    c5_1 = _cb(x, sd, p + ".conv5X5_1", 3, True)
    c5 = _cb(c5_1, sd, p + ".conv5X5_2", 3, False)
    c7_2 = _cb(c5_1, sd, p + ".conv7X7_2", 3, True)
    c7 = _cb(c7_2, sd, p + ".conv7x7_3", 3, False)
    return F.relu(torch.cat([c3, c5, c7], 1))


def _head(x, sd, p, nout):
    o = _conv(x, sd, p + ".conv1x1").permute(0, 2, 3, 1).contiguous()
    return o.view(o.size(0), -1, nout)


def forward_raw(x, sd):
    """Pre-softmax head outputs: (cls (N,P,2), bbox (N,P,4), landm (N,P,10))."""
    f = fpn(body(x, sd), sd)
    fts = [ssh(f[i], sd, f"ssh{i + 1}") for i in range(3)]
    outs = []
    for head, nout in (("ClassHead", 2), ("BboxHead", 4), ("LandmarkHead", 10)):
        outs.append(torch.cat([_head(ft, sd, f"{head}.{i}", nout) for i, ft in enumerate(fts)], 1))
    return outs


def forward(x, sd):
    cls, box, ldm = forward_raw(x, sd)
    return F.softmax(cls, dim=-1), box, ldm


def preprocess(images):
    """RGB (N,3,H,W) float 0..255 -> BGR minus mean (retinaface.py:450-451)."""
    x = images[:, [2, 1, 0]]
    return x - torch.tensor([104, 117, 123]).view(3, 1, 1)


def prior_box(h, w):
    """(P,4) float32 (cx,cy,w,h), python-double arithmetic then one rounding."""
    steps = [8, 16, 32]
    min_sizes = [[16, 32], [64, 128], [256, 512]]
    out = []
    for k, s in enumerate(steps):
        fh, fw = ceil(h / s), ceil(w / s)
        i = np.arange(fh, dtype=np.float64)[:, None, None]
        j = np.arange(fw, dtype=np.float64)[None, :, None]
        ms = np.array(min_sizes[k], dtype=np.float64)[None, None, :]
        cx = np.broadcast_to((j + 0.5) * s / w, (fh, fw, 2))
        cy = np.broadcast_to((i + 0.5) * s / h, (fh, fw, 2))
        aw = np.broadcast_to(ms / w, (fh, fw, 2))
        ah = np.broadcast_to(ms / h, (fh, fw, 2))
        out.append(np.stack([cx, cy, aw, ah], -1).reshape(-1, 4))
    return np.concatenate(out, 0).astype(np.float32)


def decode(cls_prob1, box, ldm, priors, h, w, variance=(0.1, 0.2)):
    """numpy float32 restatement of decode_bboxes / decode_landms + scaling.

    Every operation is a separately rounded float32 op in the reference's order
    (scalar python floats are applied as float32, like torch does)."""
    f = np.float32
    box = box.astype(f)
    ldm = ldm.astype(f)
    pxy, pwh = priors[:, :2], priors[:, 2:]
    v0, v1 = f(variance[0]), f(variance[1])
    cxy = pxy + (box[..., :2] * v0) * pwh
    wh = pwh * np.exp(box[..., 2:] * v1)
    x1y1 = cxy - wh / f(2)
    x2y2 = wh + x1y1
    scale_b = np.array([w, h, w, h], dtype=f)
    boxes = np.concatenate([x1y1, x2y2], -1) * scale_b
    pts = [pxy + (ldm[..., 2 * i:2 * i + 2] * v0) * pwh for i in range(5)]
    scale_l = np.array([w, h] * 5, dtype=f)
    landms = np.concatenate(pts, -1) * scale_l
    return boxes.astype(f), landms.astype(f)


def nms_single(boxes, scores, nms_threshold=0.4):
    """Greedy NMS of one image, float32 ops in the reference's order.
    Order: score descending, ties by ascending candidate position (torch CPU
    ``argsort(descending=True)`` is stable in practice; pinned by goldens)."""
    f = np.float32
    boxes = boxes.astype(f)
    area = (boxes[:, 2] - boxes[:, 0] + f(1)) * (boxes[:, 3] - boxes[:, 1] + f(1))
    order = np.argsort(-scores.astype(f), kind="stable")
    thr = f(nms_threshold)
    keep = []
    while order.size > 0:
        j = order[0]
        keep.append(int(j))
        rest = order[1:]
        xy1 = np.maximum(boxes[j, :2], boxes[rest, :2])
        xy2 = np.minimum(boxes[j, 2:], boxes[rest, 2:])
        w = np.maximum(f(0), xy2[:, 0] - xy1[:, 0] + f(1))
        h = np.maximum(f(0), xy2[:, 1] - xy1[:, 1] + f(1))
        a = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = a / (area[j] + area[rest] - a)
        order = rest[ovr <= thr]
    return keep


def filter_preds(scores, boxes, landms, vis_threshold=0.6, nms_threshold=0.4):
    """-> (landms (K,10), boxes (K,4), sample_idx list, cand_pos list-of-lists).

    ``cand_pos[i]`` are the kept candidates of image i as *prior indices*."""
    f = np.float32
    masks = scores > f(vis_threshold)
    out_l, out_b, sample_idx, kept_priors = [], [], [], []
    for i in range(scores.shape[0]):
        idx = np.nonzero(masks[i])[0]
        keep = nms_single(boxes[i, idx], scores[i, idx], nms_threshold)
        pri = idx[keep]
        out_l.append(landms[i, pri])
        out_b.append(boxes[i, pri])
        sample_idx.extend([i] * len(keep))
        kept_priors.append(pri.tolist())
    nl = landms.shape[-1]
    return (np.concatenate(out_l, 0) if out_l else np.zeros((0, nl), f),
            np.concatenate(out_b, 0) if out_b else np.zeros((0, 4), f),
            sample_idx, kept_priors)


def take_by_strategy(landms, boxes, idx, strategy):
    if strategy not in ("all", "best", "largest"):
        raise ValueError(f"Unsupported startegy: {strategy}")
    if len(idx) == 0:
        return np.zeros((0, landms.shape[-1] if landms.ndim == 2 else 10), np.float32), [], []
    f = np.float32
    idx = np.asarray(idx)
    sel = []
    starts = np.flatnonzero(np.r_[True, idx[1:] != idx[:-1]])
    ends = np.r_[starts[1:], len(idx)]
    for s, e in zip(starts, ends):
        if strategy == "all":
            sel.extend(range(s, e))
        elif strategy == "best":
            sel.append(s)
        else:
            b = boxes[s:e].astype(f)
            areas = (b[:, 2] - b[:, 0] + f(1)) * (b[:, 3] - b[:, 1] + f(1))
            sel.append(s + int(np.argmax(areas)))
    return landms[sel], idx[sel].tolist(), sel


@torch.no_grad()
def predict(images, sd, strategy="all", vis=0.6, nms_threshold=0.4, return_all=False):
    """images: torch (N,3,H,W) float32 RGB 0..255 -> ((F,5,2) float32, list[int])."""
    x = preprocess(images)
    prob, box, ldm = forward(x, sd)
    h, w = x.shape[2], x.shape[3]
    priors = prior_box(h, w)
    scores = prob[..., 1].numpy()
    boxes, landms = decode(scores, box.numpy(), ldm.numpy(), priors, h, w)
    fl, fb, sidx, kept = filter_preds(scores, boxes, landms, vis, nms_threshold)
    lm, indices, sel = take_by_strategy(fl, fb, sidx, strategy)
    lm = lm.reshape(-1, 5, 2)
    if return_all:
        return lm, indices, dict(scores=scores, boxes=boxes, landms=landms, kept=kept,
                                 fl=fl, fb=fb, sidx=sidx, sel=sel)
    return lm, indices


@torch.no_grad()
def landmarks_fp64(images, sd, ex):
    """float64 evaluation of the landmarks of the faces ``predict(..., return_all=True)`` selected (``ex`` = its third return
    value): the network (forward_raw) and decode_landms (retinaface.py:204-210, :455-461) in double precision, on the float32
    priors the reference uses, for exactly the priors that survived the float32 path's filter / NMS / strategy.  Test
    infrastructure: the yardstick that says how far the float32 oracle ITSELF is from the exact result, so that a GPU tolerance
    looser than north_star's 1e-3 px is never needed — |gpu - fp64| <= |oracle - fp64| + 1e-3 is the assertion."""
    x = preprocess(images).double()
    sd64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
    _, _, ldm = forward_raw(x, sd64)
    h, w = x.shape[2], x.shape[3]
    pri = prior_box(h, w).astype(np.float64)
    flat = [(i, p) for i, kept in enumerate(ex["kept"]) for p in kept]      # filter_preds order: image-major, keep order
    out = np.zeros((len(ex["sel"]), 5, 2), np.float64)
    scale = np.array([w, h], np.float64)
    for r, s in enumerate(ex["sel"]):
        i, p = flat[s]
        d = ldm[i, p].numpy().reshape(5, 2)
        out[r] = (pri[p, :2] + (d * 0.1) * pri[p, 2:]) * scale
    return out
