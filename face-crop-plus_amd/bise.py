"""BiSeNet face parser on MI355X (mirror of the reference's ``models/bise.py``
interface: ``BiSeNet(attr_groups, mask_groups, max_batch_size).load(device)``,
``.predict(images) -> (attr_groups | None, mask_groups | None)``; tunables
``attr_join_by_and`` / ``attr_threshold`` / ``mask_threshold`` / ``mean`` /
``std`` stay plain attributes, bise.py:184-188).

Data path per sub-batch: uint8 crops -> fused /255 + bilinear-512 + normalise ->
ResNet-18 context path / ARM / FFM / output head as fused convs (concat of
feat8 | feat16_up is a shared NHWC buffer) -> fused bilinear x8 + nearest +
argmax + 19-bin histogram.  Only label maps / masks of *selected* faces leave
the device.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _native as N
from . import engine as E
from .weights import load_state_dict

NUM_CLASSES = 19


class BiSeNet:
    WEIGHTS_FILENAME = "bise_parser.pth"

    def __init__(self, attr_groups=None, mask_groups=None, max_batch_size: int = 8):
        self.attr_groups = attr_groups
        self.mask_groups = mask_groups
        self.batch_size = max_batch_size
        self.attr_join_by_and = True
        self.attr_threshold = 5
        self.mask_threshold = 10
        self.mean = [0.485, 0.456, 0.406]
        self.std = [0.229, 0.224, 0.225]
        self.device = None
        self._p = None
        self.precision = 0
        # fp16x3 path: conv1 + bn1 + relu + maxpool of the ResNet-18 stem in one launch (A/B switch: FCP_BISE_FUSED_STEM=0)
        self.fused_stem = os.environ.get("FCP_BISE_FUSED_STEM", "1") != "0"

    def load(self, device="cuda:0", weights=None, precision=None):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("face_crop_plus_amd runs on an AMD GPU only; there is no CPU fallback")
        N.lib()
        self.device = device
        sd = load_state_dict("bisenet", weights, device=device)
        self._repack(sd, precision)
        E.selfcheck_at_load(self, sd, weights, precision, lambda: self._repack(sd, "f32"))
        return self

    def _repack(self, sd, precision):
        with torch.cuda.device(self.device), E.default_precision(precision):
            self._p = self._pack(sd, self.device)
        self.precision = E.resolve_precision(precision)

    @torch.no_grad()
    def selfcheck(self, sd=None, faces_u8: torch.Tensor | None = None, rel_tol: float = 1e-4):
        """Range / accuracy guard of the fp16x3 path (see ``RetinaFace.selfcheck``): two calibration crops through the
        network with a max-|x| reduction behind every conv launch (``FloatingPointError`` at 2^15), and — with the state
        dict — the 1/8-resolution class logits against an exact-fp32 twin."""
        if self.precision != 1:
            self.selfcheck_report = {"skipped": "exact-fp32 path: nothing to guard"}
            return self.selfcheck_report
        dev = self.device
        with torch.cuda.device(dev):
            if faces_u8 is None:
                g = torch.Generator(device="cpu").manual_seed(20260928)
                ramp = (torch.arange(256)[:, None] // 2 + torch.arange(256)[None, :] // 2).to(torch.uint8)
                faces_u8 = torch.stack([torch.randint(0, 256, (256, 256, 3), generator=g, dtype=torch.uint8),
                                        ramp[..., None].expand(256, 256, 3).contiguous()])
            faces_u8 = faces_u8.to(dev).contiguous()
            f, h, w, _ = faces_u8.shape
            x4 = E.Act.empty(f, 512, 512, 4, dev)
            N.check(N.lib().fcp_bise_preprocess_u8(N.ptr(faces_u8), f, h, w, x4.ptr(), 512, 512, (C.c_float * 3)(*self.mean),
                                                   (C.c_float * 3)(*self.std), N.stream_ptr()), "fcp_bise_preprocess_u8")
            tuning, E.Autotune.enabled = E.Autotune.enabled, False
            try:
                with E.RangeMonitor() as mon:
                    lg = self.forward_logits8(x4)
                rep = {"launch_absmax": mon.check("BiSeNet"), "limit": E.RangeMonitor.LIMIT, "logit_rel_diff": None}
                if sd is not None:
                    twin = BiSeNet(self.attr_groups, self.mask_groups, self.batch_size).load(dev, sd, "f32")
                    ref = twin.forward_logits8(x4)
                    rep["logit_rel_diff"] = E.selfcheck_compare("BiSeNet logits", lg.buf[..., :NUM_CLASSES],
                                                                ref.buf[..., :NUM_CLASSES], rel_tol)
                    del twin, ref                                        # the exact-fp32 twin lives for this comparison only
                    torch.cuda.empty_cache()
            finally:
                E.Autotune.enabled = tuning
        self.selfcheck_report = rep
        return rep

    @staticmethod
    def _pack(sd, dev):
        pc, bn = E.pack_conv, E.bn_of
        p = {"stem": pc(sd["cp.resnet.conv1.weight"], None, bn(sd, "cp.resnet.bn1"), 2, 3, dev)}
        if E.DEFAULT_PRECISION == 1:
            # normalised fp32 face -> conv1 + bn1 + relu + maxpool in one launch (hi / lo planes staged in LDS)
            p["stem_fused"] = E.pack_stem_fused(sd["cp.resnet.conv1.weight"], bn(sd, "cp.resnet.bn1"), dev)
        blocks = []
        for li in (1, 2, 3, 4):
            for b in (0, 1):
                pre = f"cp.resnet.layer{li}.{b}"
                stride = 2 if (b == 0 and li > 1) else 1
                blk = {"c1": pc(sd[pre + ".conv1.weight"], None, bn(sd, pre + ".bn1"), stride, 1, dev),
                       "c2": pc(sd[pre + ".conv2.weight"], None, bn(sd, pre + ".bn2"), 1, 1, dev), "ds": None,
                       "feat": b == 1 and li >= 2, "li": li}
                if (pre + ".downsample.0.weight") in sd:
                    blk["ds"] = pc(sd[pre + ".downsample.0.weight"], None, bn(sd, pre + ".downsample.1"), stride, 0, dev)
                blocks.append(blk)
        p["blocks"] = blocks

        def fc(wkey, bnp):
            w = sd[wkey].reshape(sd[wkey].shape[0], -1).float().contiguous().to(dev)
            if bnp is None:
                return (w, None, None)
            b = {k: v.numpy() for k, v in bn(sd, bnp).items()}
            _, shift = E.fold_bn(np.zeros((len(b["weight"]), 1, 1, 1), np.float32), b, None)
            scale = (b["weight"] / np.sqrt(b["running_var"] + np.float32(E.BN_EPS))).astype(np.float32)
            return (w, torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev))

        for nm in ("arm16", "arm32"):
            p[nm + ".conv"] = pc(sd[f"cp.{nm}.conv.conv.weight"], None, bn(sd, f"cp.{nm}.conv.bn"), 1, 1, dev)
            p[nm + ".att"] = fc(f"cp.{nm}.conv_atten.weight", f"cp.{nm}.bn_atten")
        for nm in ("conv_head32", "conv_head16"):
            p[nm] = pc(sd[f"cp.{nm}.conv.weight"], None, bn(sd, f"cp.{nm}.bn"), 1, 1, dev)
        p["conv_avg"] = fc("cp.conv_avg.conv.weight", "cp.conv_avg.bn")
        p["ffm.convblk"] = pc(sd["ffm.convblk.conv.weight"], None, bn(sd, "ffm.convblk.bn"), 1, 0, dev)
        p["ffm.conv1"] = fc("ffm.conv1.weight", None)
        p["ffm.conv2"] = fc("ffm.conv2.weight", None)
        p["out.conv"] = pc(sd["conv_out.conv.conv.weight"], None, bn(sd, "conv_out.conv.bn"), 1, 1, dev)
        p["out.cls"] = pc(sd["conv_out.conv_out.weight"], None, None, 1, 0, dev)
        return p

    # ----------------------------------------------------------- primitives
    @staticmethod
    def _avgpool(x: E.Act):
        out = torch.empty((x.n, x.c), dtype=torch.float32, device=x.buf.device)
        N.check(N.lib().fcp_avgpool_nhwc_f32(x.ptr(), x.n, x.h * x.w, x.c, x.ld, N.ptr(out), N.stream_ptr()),
                "fcp_avgpool_nhwc_f32")
        return out

    @staticmethod
    def _fc(vec, wsb, act):
        w, scale, shift = wsb
        n, cin = vec.shape
        out = torch.empty((n, w.shape[0]), dtype=torch.float32, device=vec.device)
        N.check(N.lib().fcp_fc_f32(N.ptr(vec), N.ptr(w), N.ptr(scale), N.ptr(shift), n, cin, w.shape[0], act,
                                   N.ptr(out), N.stream_ptr()), "fcp_fc_f32")
        return out

    @staticmethod
    def _scale_add(x: E.Act, s, add_nc=None, add_t: E.Act | None = None, out: E.Act | None = None):
        if out is None:
            out = E.Act.empty(x.n, x.h, x.w, x.c, x.buf.device)
        N.check(N.lib().fcp_scale_add_nhwc_f32(x.ptr(), x.ld, N.ptr(s), N.ptr(add_nc),
                                               add_t.ptr() if add_t is not None else None,
                                               add_t.ld if add_t is not None else 0, x.n, x.h * x.w, x.c,
                                               out.ptr(), out.ld, N.stream_ptr()), "fcp_scale_add_nhwc_f32")
        return out

    # --------------------------------------------------------------- forward
    def forward_logits8(self, x4: E.Act) -> E.Act:
        """Normalised NHWC4 input (n,512,512,4) -> class logits at 1/8 resolution (n,64,64,19)."""
        p = self._p
        # fp16x3 path: conv-to-conv activations live in split32 (operands are then LDS-DMA copies); the tensors
        # read by the small attention kernels (avg-pool / scale-add) stay fp32
        f = 1 if self.precision == 1 else 0
        sp = (lambda a: E.f32_to_split32(a)) if f else (lambda a: a)
        if f and self.fused_stem and "stem_fused" in p:
            x = E.stem_relu_pool_f32(p["stem_fused"], x4, out_fmt=1)        # the 256 x 256 x 64 stem map never reaches HBM
        else:
            x = E.maxpool3x3s2(E.conv(p["stem"], x4, act_slope=0.0, out_fmt=f))
        feats = {}
        fcat = None
        for blk in p["blocks"]:
            o = E.conv(blk["c1"], x, act_slope=0.0, out_fmt=f)
            idt = x if blk["ds"] is None else E.conv(blk["ds"], x, out_fmt=f)
            out = None
            if blk["feat"] and blk["li"] == 2:
                # feat8 is the first half of FFM's concat buffer (torch.cat([fsp, fcp]), _layers.py:358)
                fcat = E.Act.empty(o.n, o.h, o.w, 256, o.buf.device, f)
                out = fcat.slice(0, 128)
            x = E.conv(blk["c2"], o, out, act_slope=0.0, res1=idt, res1_pre=True, out_fmt=f)
            if blk["feat"]:
                feats[blk["li"]] = x
        feat8, feat16, feat32 = feats[2], feats[3], feats[4]
        # ContextPath (_layers.py:326-346)
        feat32_f = E.split32_to_f32(feat32) if f else feat32
        avg = self._fc(self._avgpool(feat32_f), p["conv_avg"], 1)                     # (n,128)
        f32 = E.conv(p["arm32.conv"], feat32, act_slope=0.0)
        att = self._fc(self._avgpool(f32), p["arm32.att"], 2)
        f32s = self._scale_add(f32, att, add_nc=avg)                                  # feat*atten + avg_up
        f32u = E.conv(p["conv_head32"], f32s, act_slope=0.0, in_up2=True)             # nearest x2 fused
        f16 = E.conv(p["arm16.conv"], feat16, act_slope=0.0)
        att = self._fc(self._avgpool(f16), p["arm16.att"], 2)
        f16s = self._scale_add(f16, att, add_t=f32u)
        E.conv(p["conv_head16"], f16s, fcat.slice(128, 128), act_slope=0.0, in_up2=True)
        # FeatureFusionModule (_layers.py:357-368)
        feat = E.conv(p["ffm.convblk"], fcat, act_slope=0.0)
        att = self._fc(self._fc(self._avgpool(feat), p["ffm.conv1"], 1), p["ffm.conv2"], 2)
        feat = self._scale_add(feat, att, add_t=feat)                                 # feat*atten + feat
        out = E.conv(p["out.conv"], sp(feat), act_slope=0.0)                          # fp32 out: out.cls has 19 filters
        return E.conv(p["out.cls"], out)

    def parse(self, faces_u8: torch.Tensor):
        """(F,h,w,3) uint8 device crops -> (labels (F,h,w) uint8, counts (F,19) int32), device."""
        assert faces_u8.dtype == torch.uint8 and faces_u8.dim() == 4 and faces_u8.shape[3] == 3
        faces_u8 = faces_u8.contiguous()
        f, h, w, _ = faces_u8.shape
        dev = faces_u8.device
        labels = torch.empty((f, h, w), dtype=torch.uint8, device=dev)
        counts = torch.empty((f, NUM_CLASSES), dtype=torch.int32, device=dev)
        mean = (C.c_float * 3)(*self.mean)
        std = (C.c_float * 3)(*self.std)
        lib, st = N.lib(), N.stream_ptr()
        for s in range(0, f, self.batch_size):
            e = min(f, s + self.batch_size)
            x4 = E.Act.empty(e - s, 512, 512, 4, dev)
            N.check(lib.fcp_bise_preprocess_u8(N.ptr(faces_u8, s * h * w * 3), e - s, h, w, x4.ptr(), 512, 512,
                                               mean, std, st), "fcp_bise_preprocess_u8")
            lg = self.forward_logits8(x4)
            N.check(lib.fcp_parse_tail(lg.ptr(), e - s, lg.h, lg.w, lg.ld, NUM_CLASSES, 512, 512, h, w,
                                       N.ptr(labels, s * h * w), N.ptr(counts, s * NUM_CLASSES * 4), st),
                    "fcp_parse_tail")
        return labels, counts

    # ------------------------------------------------------------- grouping
    def group_by_attributes(self, counts: np.ndarray):
        """bise.py:249-267 on the per-face class histogram."""
        out = {}
        for k, v in self.attr_groups.items():
            tests = np.stack([counts[:, abs(a)] > self.attr_threshold if a > 0
                              else counts[:, abs(a)] <= self.attr_threshold for a in v], 1)
            ok = tests.all(1) if self.attr_join_by_and else tests.any(1)
            out[k] = [int(i) for i in np.nonzero(ok)[0]]
        return out

    def group_by_masks(self, labels: torch.Tensor, counts: np.ndarray):
        """bise.py:310-325: mask = any-of classes, kept iff its pixel count > mask_threshold."""
        out = {}
        f, h, w = labels.shape
        for k, v in self.mask_groups.items():
            cls = sorted({int(a) for a in v if 0 <= int(a) < NUM_CLASSES})
            sums = counts[:, cls].sum(1) if cls else np.zeros(f, np.int64)
            inds = [int(i) for i in np.nonzero(sums > self.mask_threshold)[0]]
            masks = np.zeros((0, h, w), np.uint8)
            if inds:
                bits = 0
                for a in cls:
                    bits |= 1 << a
                sel = labels[torch.as_tensor(inds, device=labels.device)].contiguous()
                m = torch.empty_like(sel)
                N.check(N.lib().fcp_label_mask_u8(N.ptr(sel), sel.numel(), bits, N.ptr(m), N.stream_ptr()),
                        "fcp_label_mask_u8")
                masks = m.cpu().numpy()
            out[k] = (inds, masks)
        return out

    @torch.no_grad()
    def predict(self, images):
        """Reference signature (bise.py:328-418).  ``images``: (N,3,H,W) float 0..255 tensor, a
        list of such (3,H,W) tensors, or — the fast path — (N,H,W,3) uint8 crops."""
        with torch.cuda.device(self.device):
            if isinstance(images, list):
                images = torch.stack(images)
            if images.dtype != torch.uint8:
                images = images.permute(0, 2, 3, 1).to(torch.uint8)     # crops are integral 0..255 values
            images = images.to(self.device).contiguous()
            labels, counts = self.parse(images)
            counts = counts.cpu().numpy().astype(np.int64)
            attr_groups, mask_groups = None, None
            if self.attr_groups is not None:
                attr_groups = {k: v for k, v in self.group_by_attributes(counts).items() if len(v) > 0}
            if self.mask_groups is not None:
                mask_groups = {k: v for k, v in self.group_by_masks(labels, counts).items() if len(v[0]) > 0}
        return attr_groups, mask_groups
