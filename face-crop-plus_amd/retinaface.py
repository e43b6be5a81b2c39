"""RetinaFace 5-point landmark detector on MI355X (mirror of the reference's
``models/retinaface.py`` interface: ``RetinaFace(strategy, vis).load(device)``,
``.predict(images) -> (ndarray (F,5,2) f32, list[int])``; tunables
``nms_threshold`` / ``variance`` / ``vis_threshold`` / ``strategy`` stay plain
attributes, retinaface.py:87-90).

Every layer is one ``fcp_conv2d_nhwc_f32`` launch (BatchNorm folded, ReLU /
residual / FPN top-down add fused); SSH's three-way concat and the three heads
are realised as channel slices of shared NHWC buffers; decode / threshold /
compaction / sort / NMS / strategy run in three more kernels, nothing of the
data path touches the host before the final landmark copy.
"""
from __future__ import annotations


import os
import threading

import numpy as np
import torch

from . import _native as N
from . import engine as E
from . import torch_ops as T
from .weights import load_state_dict

STRATEGIES = {"all": 0, "best": 1, "largest": 2}


class RetinaFace:
    WEIGHTS_FILENAME = "retinaface_detector.pth"

    def __init__(self, strategy: str = "all", vis: float = 0.6):
        self.strategy = strategy
        self.vis_threshold = vis
        self.nms_threshold = 0.4
        self.variance = [0.1, 0.2]
        self.device = None
        self._p = None
        self.precision = 0
        self.fused_stem = os.environ.get("FCP_FUSED_STEM", "1") != "0"   # fp16x3 path: uint8 -> stem + pool in one launch
        self.fused_stem_conv1 = os.environ.get("FCP_FUSED_STEM_CONV1", "1") != "0"   # ... + layer1.0.conv1 in the same launch
        # The network runs on the two halves of a batch concurrently, on two HIP streams: the tail wave of one
        # half's launch (layers 3-4 fill only ~78 % of their last round of workgroups) overlaps the head of the
        # other's.  Images are independent, so the result is bit-identical to the single-stream pass.
        # conv2 + conv3 of an identity bottleneck and conv1 of the next block in one launch (layer 1; bit-identical)
        self.fused_chain = os.environ.get("FCP_FUSED_CHAIN", "1") != "0"
        self.streams = int(os.environ.get("FCP_DET_STREAMS", "2"))
        self.min_images_per_stream = 6      # tools/probe_min_images_per_stream.py: batch 8 is best on one stream (1210 vs 1165 faces/s), batch 12 on two (1308 vs 1288)
        self.split_cu_budget = os.environ.get("FCP_SPLIT_CU_BUDGET", "1") != "0"
        self._tls = threading.local()

    # ------------------------------------------------------------------ load
    def load(self, device: str | torch.device = "cuda:0", weights=None, precision=None):
        """Pack the state dict for the HIP engine (reference ``LoadMixin.load``,
        _layers.py:16-25).  ``weights``: None (the real checkpoint: local file, else the
        reference's download, else an error), a path, a state dict, or "generated" (seeded random weights).  ``precision``:
        "f16x3" (split-fp16 MFMA, fp32-equivalent accuracy, default) or "f32" (exact fp32 MFMA)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("face_crop_plus_amd runs on an AMD GPU only (device must be cuda:N / hip); "
                               "there is no CPU fallback")
        N.lib()  # fail loudly now if the extension is missing
        self.device = device
        sd = load_state_dict("retinaface", weights, device=device)
        self._repack(sd, precision)
        E.device_props(device)          # cached here, on the loading thread: the worker threads only ever read the cache
        # Range / accuracy guard of the fp16x3 path (``selfcheck``): FCP_SELFCHECK=1 always, 0 never; by default ("auto")
        # whenever the weights come from a checkpoint — a file, the hub cache or a download — i.e. are not this package's
        # own generated ones or a state dict the caller built in memory.  Weights the split-binary16 path cannot carry
        # send a default-precision load to the exact-fp32 path (``E.selfcheck_at_load``).
        E.selfcheck_at_load(self, sd, weights, precision, lambda: self._repack(sd, "f32"))
        return self

    def _repack(self, sd, precision):
        with torch.cuda.device(self.device), E.default_precision(precision):
            self._p = self._pack(sd, self.device)
        self.precision = E.resolve_precision(precision)

    @torch.no_grad()
    def selfcheck(self, sd=None, images_u8: torch.Tensor | None = None, rel_tol: float = 1e-4):
        """Guard of the split-binary16 conv path against weights it cannot represent (the reference computes in fp32 and
        takes any checkpoint, _layers.py:16-35; here binary16 hi parts saturate at 65504).  Runs a small calibration batch
        (noise, white, black, a ramp — or ``images_u8``) through the network with a max-|x| reduction behind every
        launch and raises ``FloatingPointError`` naming the launch if any activation reaches 2^15; with ``sd`` (the
        state dict) it also packs an exact-fp32 twin and requires the fused head maps of the two paths to agree within
        ``rel_tol`` of the map's largest value.  Returns (and keeps in ``self.selfcheck_report``) the per-launch maxima
        and the head differences."""
        if self.precision != 1:
            self.selfcheck_report = {"skipped": "exact-fp32 path: nothing to guard"}
            return self.selfcheck_report
        dev = self.device
        with torch.cuda.device(dev):
            if images_u8 is None:
                g = torch.Generator(device="cpu").manual_seed(20260928)
                s = 256
                ramp = (torch.arange(s)[:, None] + torch.arange(s)[None, :]).clamp(max=255).to(torch.uint8)
                images_u8 = torch.stack([torch.randint(0, 256, (s, s, 3), generator=g, dtype=torch.uint8),
                                         torch.full((s, s, 3), 255, dtype=torch.uint8), torch.zeros((s, s, 3), dtype=torch.uint8),
                                         ramp[..., None].expand(s, s, 3).contiguous()]).to(dev)
            images_u8 = images_u8.to(dev).contiguous()
            tuning, E.Autotune.enabled = E.Autotune.enabled, False        # a check must not spend time tuning tiles
            try:
                with E.RangeMonitor() as mon:
                    heads = self.forward_heads(None, images_u8) if "stem_fused" in self._p and self.fused_stem else \
                        self.forward_heads(E.u8_to_nhwc4(images_u8, sub=(123.0, 117.0, 104.0)))
                rep = {"launch_absmax": mon.check("RetinaFace"), "limit": E.RangeMonitor.LIMIT, "head_rel_diff": None}
                if sd is not None:
                    twin = RetinaFace(self.strategy, self.vis_threshold)
                    twin.device, twin.precision = dev, 0
                    with E.default_precision("f32"):
                        twin._p = self._pack(sd, dev)
                    ref = twin.forward_heads(E.u8_to_nhwc4(images_u8, sub=(123.0, 117.0, 104.0)))
                    rep["head_rel_diff"] = [E.selfcheck_compare("RetinaFace head maps", a.buf, b.buf, rel_tol)
                                            for a, b in zip(heads, ref)]
                    del twin, ref                                        # the exact-fp32 twin (a second copy of the filters
                    torch.cuda.empty_cache()                             # on the device) lives for this comparison only
            finally:
                E.Autotune.enabled = tuning
        self.selfcheck_report = rep
        return rep

    @staticmethod
    def _pack(sd, dev):
        p = {}
        pc, bn = E.pack_conv, E.bn_of
        # stem: kernel keeps RGB channel order, so swap the filter's input channels (BGR) instead
        p["stem"] = pc(sd["body.conv1.weight"], None, bn(sd, "body.bn1"), 2, 3, dev, cin_perm=[2, 1, 0])
        if E.DEFAULT_PRECISION == 1:
            # uint8 batch -> conv1 + bn1 + relu + maxpool in one launch (x - mean is exact in binary16)
            p["stem_fused"] = E.pack_stem_fused(sd["body.conv1.weight"], bn(sd, "body.bn1"), dev, cin_perm=[2, 1, 0])
        blocks = []
        for li, nb in enumerate((3, 4, 6, 3), 1):
            for b in range(nb):
                pre = f"body.layer{li}.{b}"
                stride = 2 if (b == 0 and li > 1) else 1
                blk = {
                    "c1": pc(sd[pre + ".conv1.weight"], None, bn(sd, pre + ".bn1"), 1, 0, dev),
                    "c2": pc(sd[pre + ".conv2.weight"], None, bn(sd, pre + ".bn2"), stride, 1, dev),
                    "c3": pc(sd[pre + ".conv3.weight"], None, bn(sd, pre + ".bn3"), 1, 0, dev),
                    "ds": None, "feat": (b == nb - 1 and li >= 2),
                }
                if (pre + ".downsample.0.weight") in sd:
                    blk["ds"] = pc(sd[pre + ".downsample.0.weight"], None, bn(sd, pre + ".downsample.1"),
                                   stride, 0, dev)
                    if stride == 1 or E.DEFAULT_PRECISION == 1:
                        # bn3(conv3(o)) + bn_d(down(x)) is ONE 1x1 conv over the channel concat [o | x(::s, ::s)]:
                        # the downsampled identity tensor is never written nor re-read.  layer1.0 (stride 1, both
                        # sources at one resolution) reads a real concat buffer; the stride-2 blocks use the
                        # two-source form of the LDS-DMA kernels (fp16x3 path).
                        fold = lambda cw, cb: E.fold_bn(sd[pre + cw].numpy(), {q: v.numpy() for q, v in bn(sd, pre + cb).items()}, None)
                        (w3, b3), (wd, bd) = fold(".conv3.weight", ".bn3"), fold(".downsample.0.weight", ".downsample.1")
                        blk["c3ds"] = pc(np.concatenate([w3, wd], 1), b3 + bd, None, 1, 0, dev)
                blocks.append(blk)
        p["blocks"] = blocks
        for i in (1, 2, 3):
            p[f"fpn.output{i}"] = pc(sd[f"fpn.output{i}.0.weight"], None, bn(sd, f"fpn.output{i}.1"), 1, 0, dev)
        for i in (1, 2):
            p[f"fpn.merge{i}"] = pc(sd[f"fpn.merge{i}.0.weight"], None, bn(sd, f"fpn.merge{i}.1"), 1, 1, dev)
        # SSH buffer layout per level (384 ch): [B c5_1 (64) | A c3 (128) | E c7 (64) | C c5 (64) | D c7_2 (64)]
        head_perm = list(range(0, 128)) + list(range(192, 256)) + list(range(128, 192))  # reads [A | E | C]
        for k in (1, 2, 3):
            def folded(nm):
                return E.fold_bn(sd[f"ssh{k}.{nm}.0.weight"].numpy(), {q: v.numpy() for q, v in bn(sd, f"ssh{k}.{nm}.1").items()}, None)
            w_b, b_b = folded("conv5X5_1")
            w_a, b_a = folded("conv3X3")
            p[f"ssh{k}.ab"] = pc(np.concatenate([w_b, w_a]), np.concatenate([b_b, b_a]), None, 1, 1, dev)
            w_c, b_c = folded("conv5X5_2")
            w_d, b_d = folded("conv7X7_2")
            p[f"ssh{k}.cd"] = pc(np.concatenate([w_c, w_d]), np.concatenate([b_c, b_d]), None, 1, 1, dev)
            w_e, b_e = folded("conv7x7_3")
            p[f"ssh{k}.e"] = pc(w_e, b_e, None, 1, 1, dev)
            i = k - 1
            hw = np.concatenate([sd[f"{h}.{i}.conv1x1.weight"].numpy() for h in ("ClassHead", "BboxHead", "LandmarkHead")])
            hb = np.concatenate([sd[f"{h}.{i}.conv1x1.bias"].numpy() for h in ("ClassHead", "BboxHead", "LandmarkHead")])
            p[f"head{k}"] = pc(hw, hb, None, 1, 0, dev, cin_perm=head_perm)
        return p

    # --------------------------------------------------------------- forward
    def forward_heads(self, x4: E.Act | None, images_u8: torch.Tensor | None = None, heads_out=None):
        """NHWC4 (RGB - mean) — or, on the fp16x3 path, the uint8 batch itself — -> three fused head maps
        (n, h/8|16|32, w/.., 32).  ``heads_out``: optional three pre-allocated fp32 views to write them into."""
        p = self._p
        # fp16x3 path: activations between convs live in the "split32" format (hi/lo binary16 planes per 32
        # channels, same bytes as fp32) so every consumer conv copies its operand instead of converting it
        f = 1 if self.precision == 1 else 0
        stem_t1 = None
        if images_u8 is not None and "stem_fused" in p:
            n, h, w, _ = images_u8.shape
            hp, wp = ((h - 1) // 2) // 2 + 1, ((w - 1) // 2) // 2 + 1
            cat = E.Act.empty(n, hp, wp, 128, images_u8.device, f)                   # [conv2 out | pooled stem]
            c1 = p["blocks"][0]["c1"]
            if self.fused_stem_conv1 and f == 1 and E.stem_conv1_supported(c1):
                # ... and conv1 of layer1.0 in the same launch: the pooled map is written (downsample branch) but not re-read
                x, stem_t1 = E.stem_relu_pool_u8(p["stem_fused"], images_u8, cat.slice(64, 64), conv1=c1)
            else:
                x = E.stem_relu_pool_u8(p["stem_fused"], images_u8, cat.slice(64, 64))
        else:
            x = E.conv(p["stem"], x4, act_slope=0.0, out_fmt=f)
            cat = E.Act.empty(x.n, (x.h + 1) // 2, (x.w + 1) // 2, 2 * x.c, x.buf.device, f)
            x = E.maxpool3x3s2(x, cat.slice(x.c, x.c))
        feats = []
        blocks, pre = p["blocks"], stem_t1
        chain = bool(f) and self.fused_chain
        for bi, blk in enumerate(blocks):
            o = pre if pre is not None else E.conv(blk["c1"], x, act_slope=0.0, out_fmt=f)
            pre = None
            nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
            if (chain and blk["ds"] is None and not blk["feat"] and nxt is not None
                    and E.chain_supported(blk["c2"], blk["c3"], nxt["c1"])):
                # conv2 + conv3 (+ identity) of this block and conv1 of the next one in one launch (layer 1).  The last block
                # of layer 1 feeds nothing but layer2.0 — whose conv1 is `pre` and whose 1x1 / 2 downsample (folded into its
                # two-source conv3 below) samples x at even pixels only: three quarters of x need not be written
                sparse = "c3ds" in nxt and nxt["c2"].stride == 2 and not blk["feat"]
                x, pre = E.bottleneck_chain(blk["c2"], blk["c3"], nxt["c1"], o, x, out_even_only=sparse)   # pre: next block's conv1 output
                continue
            if "c3ds" in blk and blk["c2"].stride == 1:
                E.conv(blk["c2"], o, cat.slice(0, o.c), act_slope=0.0)
                if chain and nxt is not None and E.chain_supported(None, blk["c3ds"], nxt["c1"], residual=False):
                    x, pre = E.bottleneck_chain(None, blk["c3ds"], nxt["c1"], cat, None)   # conv3 + downsample, next conv1
                else:
                    x = E.conv(blk["c3ds"], cat, act_slope=0.0, out_fmt=f)
                del cat
                continue
            if "c3ds" in blk:
                o = E.conv(blk["c2"], o, act_slope=0.0, out_fmt=f)
                if (chain and not blk["feat"] and nxt is not None
                        and E.chain_supported(None, blk["c3ds"], nxt["c1"], residual=False, cb=x.c)):
                    # layer2.0: conv3 + the 1x1 / 2 downsample over the two sources [conv2 out | x(::2, ::2)] AND layer2.1's
                    # conv1 in one launch: the 512-channel block output is written once and not read back by a conv1 launch
                    x, pre = E.bottleneck_chain(None, blk["c3ds"], nxt["c1"], o, None, t1b=x, t1b_stride=blk["c2"].stride)
                    continue
                x = E.conv(blk["c3ds"], o, act_slope=0.0, out_fmt=f, x2=x, x2_stride=blk["c2"].stride)
                if blk["feat"]:
                    feats.append(x)
                continue
            o = E.conv(blk["c2"], o, act_slope=0.0, out_fmt=f)
            if chain and blk["ds"] is None and E.chain_supported(None, blk["c3"], None) and (
                    E.L3_FORM == "expand" or (E.L3_FORM == "pair" and (nxt is None or not E.chain_supported(None, blk["c3"], nxt["c1"])))):
                x, _ = E.bottleneck_chain(None, blk["c3"], None, o, x)       # conv3 + identity on the expand form; conv1 follows as a launch
                if blk["feat"]:
                    feats.append(x)
                continue
            if (chain and blk["ds"] is None and nxt is not None and E.chain_supported(None, blk["c3"], nxt["c1"])):
                x, pre = E.bottleneck_chain(None, blk["c3"], nxt["c1"], o, x)          # conv3 (+ identity) + next conv1 (layers 2-3)
                if blk["feat"]:
                    feats.append(x)
                continue
            idt = x if blk["ds"] is None else E.conv(blk["ds"], x, out_fmt=f)
            x = E.conv(blk["c3"], o, act_slope=0.0, res1=idt, res1_pre=True, out_fmt=f)
            if blk["feat"]:
                feats.append(x)
        if getattr(self, "_debug_feats", None) is not None:      # tests: the body's three feature maps (layer2..4)
            self._debug_feats.extend(feats)
        # FPN (_layers.py:127-145): LeakyReLU slope 0 == ReLU for 256 channels
        o3 = E.conv(p["fpn.output3"], feats[2], act_slope=0.0, out_fmt=f)
        o2 = E.conv(p["fpn.output2"], feats[1], act_slope=0.0, res1=o3, res1_pre=False, out_fmt=f)
        o2 = E.conv(p["fpn.merge2"], o2, act_slope=0.0, out_fmt=f)
        o1 = E.conv(p["fpn.output1"], feats[0], act_slope=0.0, res1=o2, res1_pre=False, out_fmt=f)
        o1 = E.conv(p["fpn.merge1"], o1, act_slope=0.0, out_fmt=f)
        heads = []
        for k, ft in zip((1, 2, 3), (o1, o2, o3)):
            s = E.Act.empty(ft.n, ft.h, ft.w, 384, ft.buf.device, f)
            E.conv(p[f"ssh{k}.ab"], ft, s.slice(0, 192), act_slope=0.0)
            E.conv(p[f"ssh{k}.cd"], s.slice(0, 64), s.slice(256, 128), act_slope=0.0)
            E.conv(p[f"ssh{k}.e"], s.slice(320, 64), s.slice(192, 64), act_slope=0.0)
            heads.append(E.conv(p[f"head{k}"], s.slice(64, 256),      # fp32 out: the decode kernel reads it
                                None if heads_out is None else heads_out[k - 1]))
        return heads

    def _side_streams(self, dev, k):
        """k HIP streams of the calling host thread (process_dir's GPU workers each get their own set), shared by every
        detector the thread runs and by the threads of later runs: ``engine.thread_side_streams``.  A thread that is already
        running on its own main stream (a GPU worker of process_dir) keeps one sub-batch there."""
        main = E.thread_streams(dev)["main"]
        return E.thread_side_streams(dev, k, with_main=main is not None and torch.cuda.current_stream(dev) == main)

    def _forward_heads_split(self, images_u8: torch.Tensor):
        """``forward_heads`` of a uint8 batch, its ``self.streams`` contiguous sub-batches enqueued on side streams
        that fork from and re-join the caller's stream; every sub-batch writes its rows of the shared head maps."""
        n, h, w, _ = images_u8.shape
        k = min(self.streams, n // max(1, self.min_images_per_stream))
        if k < 2 or torch.cuda.is_current_stream_capturing():
            return self.forward_heads(None, images_u8)
        dev = images_u8.device
        heads = [E.Act.empty(n, -(-h // s), -(-w // s), 32, dev) for s in (8, 16, 32)]
        cur = torch.cuda.current_stream(dev)
        bounds = [n * i // k for i in range(k + 1)]
        side = self._side_streams(dev, k)
        # each sub-batch lays out the last dispatch round of its 256-row conv launches for its share of the CUs
        cus = E.device_props(dev).multi_processor_count // k if self.split_cu_budget else 0
        for st, a, b in zip(side, bounds[:-1], bounds[1:]):
            tuned = len(E.Autotune.cache)
            if st != cur:
                st.wait_stream(cur)
            with torch.cuda.stream(st), E.cu_budget(cus):
                self.forward_heads(None, images_u8[a:b], [E.Act(hd.buf[a:b]) for hd in heads])
            if len(E.Autotune.cache) != tuned:
                # this sub-batch met untuned shapes (first call of a geometry): their tile candidates were timed with HIP
                # events, so let it finish alone before the next sub-batch, with the same shapes, starts
                st.synchronize()
        for st in side:
            if st != cur:
                cur.wait_stream(st)
        return heads

    # ---------------------------------------------------------------- detect
    def detect(self, images_u8: torch.Tensor | None = None, *, x4: E.Act | None = None,
               paddings: torch.Tensor | None = None, max_faces: int | None = None, want_dense: bool = False):
        """Device-resident detection.  ``images_u8``: (n,h,w,3) uint8 RGB on the GPU.

        Returns a dict of device tensors: ``landmarks`` (max_faces,5,2) f32 (padding
        already subtracted), ``img_idx`` (max_faces,) i32, ``face_offset`` (n+1,) i32
        (``face_offset[n]`` = number of faces), plus the candidate / keep arrays.
        """
        if self.strategy not in STRATEGIES:
            raise ValueError(f"Unsupported startegy: {self.strategy}")
        fused = x4 is None and "stem_fused" in self._p and self.fused_stem
        if x4 is None and not fused:
            # RGB order is kept; means are (R,G,B) = (123,117,104) (retinaface.py:450)
            x4 = E.u8_to_nhwc4(images_u8, sub=(123.0, 117.0, 104.0))
        if fused:
            images_u8 = images_u8.contiguous()
            (n, h, w), dev = images_u8.shape[:3], images_u8.device
            heads = self._forward_heads_split(images_u8)
        else:
            n, h, w = x4.n, x4.h, x4.w
            dev = x4.buf.device
            heads = self.forward_heads(x4)
        P = sum(2 * (-(-h // s)) * (-(-w // s)) for s in (8, 16, 32))
        f32, i32 = torch.float32, torch.int32
        cand_score = torch.empty((n, P), dtype=f32, device=dev)
        cand_box = torch.empty((n, P, 4), dtype=f32, device=dev)
        cand_ldm = torch.empty((n, P, 10), dtype=f32, device=dev)
        cand_prior = torch.empty((n, P), dtype=i32, device=dev)
        cand_count = torch.empty((n,), dtype=i32, device=dev)
        dense = (None, None, None)
        if want_dense:
            dense = (torch.empty((n, P), dtype=f32, device=dev), torch.empty((n, P, 4), dtype=f32, device=dev),
                     torch.empty((n, P, 10), dtype=f32, device=dev))
        if T.ENABLED and not want_dense:
            return self._postprocess_ops(heads, n, h, w, P, paddings, max_faces)
        st = N.stream_ptr()
        lib = N.lib()
        N.check(lib.fcp_retina_decode(heads[0].ptr(), heads[1].ptr(), heads[2].ptr(), n, h, w,
                                      float(self.vis_threshold), float(self.variance[0]), float(self.variance[1]),
                                      N.ptr(cand_score), N.ptr(cand_box), N.ptr(cand_ldm), N.ptr(cand_prior),
                                      N.ptr(cand_count), N.ptr(dense[0]), N.ptr(dense[1]), N.ptr(dense[2]), st),
                "fcp_retina_decode")
        out = nms_select(cand_score, cand_box, cand_count, self.nms_threshold, self.strategy)
        if max_faces is None:
            max_faces = n if self.strategy != "all" else int(out["sel_count"].sum().item())
        max_faces = max(int(max_faces), 1)
        landmarks = torch.empty((max_faces, 5, 2), dtype=f32, device=dev)     # the gather kernel zeroes the unused tail
        img_idx = torch.empty((max_faces,), dtype=i32, device=dev)
        face_offset = torch.empty((n + 1,), dtype=i32, device=dev)
        if paddings is not None:
            paddings = paddings.to(device=dev, dtype=i32).contiguous()
        N.check(lib.fcp_retina_gather_faces(N.ptr(cand_ldm), N.ptr(out["sel_pos"]), N.ptr(out["sel_count"]), n, P,
                                            N.ptr(paddings), max_faces, N.ptr(face_offset), N.ptr(landmarks),
                                            N.ptr(img_idx), st), "fcp_retina_gather_faces")
        out.update(landmarks=landmarks, img_idx=img_idx, face_offset=face_offset, cand_score=cand_score,
                   cand_box=cand_box, cand_ldm=cand_ldm, cand_prior=cand_prior, cand_count=cand_count,
                   heads=heads, dense=dense, max_faces=max_faces)
        return out

    def _postprocess_ops(self, heads, n, h, w, P, paddings, max_faces):
        """decode -> NMS / strategy -> gather through ``torch.ops.fcp`` (FCP_BOUNDARY=torch); same result dict."""
        ops = T.load()
        dev = heads[0].buf.device
        cand_score, cand_box, cand_ldm, cand_prior, cand_count = ops.retina_decode(
            heads[0].buf, heads[1].buf, heads[2].buf, h, w, float(self.vis_threshold), float(self.variance[0]),
            float(self.variance[1]))
        keep_pos, keep_count, sel_pos, sel_count = ops.nms_select(cand_score, cand_box, cand_count, float(self.nms_threshold),
                                                                  STRATEGIES[self.strategy])
        if max_faces is None:
            max_faces = n if self.strategy != "all" else int(sel_count.sum().item())
        max_faces = max(int(max_faces), 1)
        if paddings is not None:
            paddings = paddings.to(device=dev, dtype=torch.int32).contiguous()
        landmarks, img_idx, face_offset = ops.gather_faces(cand_ldm, sel_pos, sel_count, paddings, max_faces)
        return dict(keep_pos=keep_pos, keep_count=keep_count, sel_pos=sel_pos, sel_count=sel_count, landmarks=landmarks,
                    img_idx=img_idx, face_offset=face_offset, cand_score=cand_score, cand_box=cand_box, cand_ldm=cand_ldm,
                    cand_prior=cand_prior, cand_count=cand_count, heads=heads, dense=(None, None, None), max_faces=max_faces)

    # ----------------------------------------------------------------- graphs
    def graphed(self, n: int, h: int, w: int, paddings: torch.Tensor | None = None):
        """Capture one detection step for a fixed (n,h,w) uint8 batch into a HIP graph.

        Small batches are launch-bound (~75 kernel launches from Python per step); replaying a
        captured graph removes the host from the loop.  Returns ``(static_images, result, graph)``:
        write the batch into ``static_images``, call ``graph.replay()``, read ``result`` (the dict
        ``detect`` returns; all tensors are static).  Only for strategies with a bounded face count
        ("best" / "largest"), which need no host read-back inside the step."""
        if self.strategy == "all":
            raise ValueError('graph capture needs a bounded face count: strategy "best" or "largest"')
        key = (n, h, w, None if paddings is None else tuple(paddings.flatten().tolist()))
        cache = self.__dict__.setdefault("_graphs", {})
        if key in cache:
            return cache[key]
        with torch.cuda.device(self.device):
            static = torch.zeros((n, h, w, 3), dtype=torch.uint8, device=self.device)
            prev, E.Autotune.enabled = E.Autotune.enabled, True
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):              # warm-up: lazy module loads, tile autotuning, allocator
                    self.detect(static, paddings=paddings, max_faces=n)
            torch.cuda.current_stream().wait_stream(side)
            E.Autotune.enabled = False
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                res = self.detect(static, paddings=paddings, max_faces=n)
            E.Autotune.enabled = prev
        cache[key] = (static, res, graph)
        return cache[key]

    # --------------------------------------------------------------- predict
    @torch.no_grad()
    def predict(self, images: torch.Tensor):
        """Reference signature (retinaface.py:411-470): ``images`` (N,3,H,W) float
        RGB 0..255 (or (N,H,W,3) uint8) -> ((F,5,2) float32 ndarray, list[int])."""
        with torch.cuda.device(self.device):
            if images.dtype == torch.uint8:
                res = self.detect(images.to(self.device).contiguous())
            else:
                x4 = E.f32nchw_to_nhwc4(images.to(self.device, torch.float32), sub=(123.0, 117.0, 104.0))
                res = self.detect(x4=x4)
            nf = int(res["face_offset"][-1].item())
            nf = min(nf, res["max_faces"])
            landmarks = res["landmarks"][:nf].cpu().numpy()
            indices = res["img_idx"][:nf].cpu().numpy().astype(np.int64).tolist()
        return landmarks, indices


def nms_select(cand_score, cand_box, cand_count, nms_threshold=0.4, strategy="all"):
    """Sort + greedy NMS + take_by_strategy on device (retinaface.py:270-304, :363-408)."""
    if strategy not in STRATEGIES:
        raise ValueError(f"Unsupported startegy: {strategy}")
    n, cap = cand_score.shape
    dev = cand_score.device
    i32 = torch.int32
    lib = N.lib()
    ws = torch.empty((int(lib.fcp_retina_nms_workspace_bytes(n, cap)),), dtype=torch.uint8, device=dev)
    keep_pos = torch.empty((n, cap), dtype=i32, device=dev)
    keep_count = torch.empty((n,), dtype=i32, device=dev)
    sel_pos = torch.empty((n, cap), dtype=i32, device=dev)
    sel_count = torch.empty((n,), dtype=i32, device=dev)
    N.check(lib.fcp_retina_nms_select(N.ptr(cand_score), N.ptr(cand_box), N.ptr(cand_count), n, cap,
                                      float(nms_threshold), STRATEGIES[strategy], N.ptr(ws), N.ptr(keep_pos),
                                      N.ptr(keep_count), N.ptr(sel_pos), N.ptr(sel_count), N.stream_ptr()),
            "fcp_retina_nms_select")
    return dict(keep_pos=keep_pos, keep_count=keep_count, sel_pos=sel_pos, sel_count=sel_count)
