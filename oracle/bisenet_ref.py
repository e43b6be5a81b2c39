"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement (torch CPU fp32 + numpy) of the reference's BiSeNet face parser,
written against a plain state_dict.  Pinned against the reference module itself
by tests/golden/make_golden.py.

Follows (paths relative to /root/reference/src/face_crop_plus):
  forward                 models/bise.py:195-212, models/_layers.py:206-368
  predict                 models/bise.py:372-418
  group_by_attributes     models/bise.py:249-267
  group_by_masks          models/bise.py:310-325
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
MEAN = [0.485, 0.456, 0.406]
STD = [0.229, 0.224, 0.225]


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, BN_EPS)


def _cbr(x, sd, p, k=3, stride=1, pad=1):
    return F.relu(_bn(F.conv2d(x, sd[p + ".conv.weight"], None, stride, pad), sd, p + ".bn"))


def _basic(x, sd, p, stride):
    r = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
    r = _bn(F.conv2d(r, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    s = x
    if (p + ".downsample.0.weight") in sd:
        s = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
    return F.relu(s + r)


def resnet18(x, sd, p="cp.resnet"):
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, 2, 3), sd, p + ".bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li in (1, 2, 3, 4):
        for b in (0, 1):
            x = _basic(x, sd, f"{p}.layer{li}.{b}", 2 if (b == 0 and li > 1) else 1)
        if li >= 2:
            feats.append(x)
    return feats


def _arm(x, sd, p):
    feat = _cbr(x, sd, p + ".conv")
    att = F.avg_pool2d(feat, feat.shape[2:])
    att = torch.sigmoid(_bn(F.conv2d(att, sd[p + ".conv_atten.weight"]), sd, p + ".bn_atten"))
    return feat * att


def forward_logits8(x, sd):
    """Logits at 1/8 resolution (before the final bilinear upsample)."""
    feat8, feat16, feat32 = resnet18(x, sd)
    avg = F.avg_pool2d(feat32, feat32.shape[2:])
    avg = _cbr(avg, sd, "cp.conv_avg", 1, 1, 0)
    avg_up = F.interpolate(avg, feat32.shape[2:])
    f32s = _arm(feat32, sd, "cp.arm32") + avg_up
    f32u = _cbr(F.interpolate(f32s, feat16.shape[2:]), sd, "cp.conv_head32")
    f16s = _arm(feat16, sd, "cp.arm16") + f32u
    f16u = _cbr(F.interpolate(f16s, feat8.shape[2:]), sd, "cp.conv_head16")
    fcat = torch.cat([feat8, f16u], 1)
    feat = _cbr(fcat, sd, "ffm.convblk", 1, 1, 0)
    att = F.avg_pool2d(feat, feat.shape[2:])
    att = F.relu(F.conv2d(att, sd["ffm.conv1.weight"]))
    att = torch.sigmoid(F.conv2d(att, sd["ffm.conv2.weight"]))
    feat = feat * att + feat
    out = _cbr(feat, sd, "conv_out.conv")
    return F.conv2d(out, sd["conv_out.conv_out.weight"])


def forward(x, sd):
    return F.interpolate(forward_logits8(x, sd), x.shape[2:], None, "bilinear", True)


def preprocess(images):
    """(N,3,H,W) float 0..255 -> normalised 512x512 (bise.py:387-393)."""
    mean = torch.tensor(MEAN).view(1, 3, 1, 1)
    std = torch.tensor(STD).view(1, 3, 1, 1)
    x = F.interpolate(images.div(255), (512, 512), mode="bilinear")
    return (x - mean) / std


@torch.no_grad()
def parse_labels(images, sd, batch_size=8):
    """-> (N,H,W) int64 label maps."""
    x = preprocess(images)
    outs = []
    for sub in torch.split(x, batch_size):
        o = forward(sub, sd)
        outs.append(F.interpolate(o, images.shape[2:], mode="nearest").argmax(1))
    return torch.cat(outs).numpy()


def group_by_attributes(labels, attr_groups, attr_threshold=5, join_and=True):
    out = {}
    for k, v in attr_groups.items():
        counts = np.stack([(labels == abs(a)).sum((1, 2)) for a in v], 1)
        tests = np.stack([counts[:, i] > attr_threshold if a > 0 else counts[:, i] <= attr_threshold
                          for i, a in enumerate(v)], 1)
        ok = tests.all(1) if join_and else tests.any(1)
        out[k] = [i for i in range(len(labels)) if ok[i]]
    return {k: v for k, v in out.items() if len(v) > 0}


def group_by_masks(labels, mask_groups, mask_threshold=10):
    out = {}
    for k, v in mask_groups.items():
        mask = np.isin(labels, np.asarray(v))
        inds = [i for i in range(len(labels)) if mask[i].sum() > mask_threshold]
        if len(inds) > 0:
            out[k] = (inds, (mask[inds] * 255).astype(np.uint8))
    return out


@torch.no_grad()
def predict(images, sd, attr_groups=None, mask_groups=None, batch_size=8):
    labels = parse_labels(images, sd, batch_size)
    ag = None if attr_groups is None else group_by_attributes(labels, attr_groups)
    mg = None if mask_groups is None else group_by_masks(labels, mask_groups)
    return ag, mg
