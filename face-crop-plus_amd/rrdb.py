"""RRDBNet (BSRGAN x4) enhancer on MI355X (mirror of the reference's
``models/rrdb.py`` interface: ``RRDBNet(min_face_factor).load(device)``,
``.predict(images, landmarks, indices)`` enhancing in place the images whose
mean face-area factor is <= ``min_face_factor``).

All 351 convolutions run on the matrix-core engine (fp16x3 split precision by
default, exact fp32 with ``precision="f32"``) with bias / LeakyReLU(0.2) /
``x5*0.2 + x`` / ``out*0.2 + x`` fused into the epilogues.  A dense block's
``torch.cat((x, x1, ..))`` is one 192-channel NHWC buffer: each conv reads the
leading channels and appends 32 more, so nothing is ever copied; the two nearest
x2 upsamples are folded into the following conv's operand fetch.  The x4-resolution
tail (upconv2 -> HRconv -> conv_last) is computed in bands of output rows, each
band recomputing the few rows of context its 3x3 convs need, so its 64-channel
intermediates are band-sized scratch buffers instead of two 4 GiB tensors.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _native as N
from . import engine as E
from .weights import load_state_dict


class RRDBNet:
    WEIGHTS_FILENAME = "bsrgan_x4_enhancer.pth"
    NUM_BLOCKS = 23

    def __init__(self, min_face_factor: float = 0.001):
        self.min_face_factor = min_face_factor
        self.device = None
        self._p = None
        self.precision = 0

    def load(self, device="cuda:0", weights=None, precision=None):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("face_crop_plus_amd runs on an AMD GPU only; there is no CPU fallback")
        N.lib()
        self.device = device
        sd = load_state_dict("rrdb", weights, device=device)
        self._repack(sd, precision)
        E.selfcheck_at_load(self, sd, weights, precision, lambda: self._repack(sd, "f32"))
        return self

    def _repack(self, sd, precision):
        device = self.device
        with torch.cuda.device(device), E.default_precision(precision):
            pc = lambda k, prec=None: E.pack_conv(sd[k + ".weight"], sd[k + ".bias"], None, 1, 1, device,
                                                  precision=prec)
            p = {k: pc(k) for k in ("conv_first", "trunk_conv", "upconv1", "upconv2", "HRconv")}
            # conv_last has 3 output channels; the split32 epilogue stores 8-channel groups, so its filter is
            # zero-padded to 8 outputs (channels 3..7 of the band scratch are zeros nobody reads)
            wl, bl = sd["conv_last.weight"], sd["conv_last.bias"]
            wl8 = np.zeros((8,) + tuple(wl.shape[1:]), np.float32); wl8[:3] = torch.as_tensor(wl).float().numpy()
            bl8 = np.zeros((8,), np.float32); bl8[:3] = torch.as_tensor(bl).float().numpy()
            p["conv_last"] = E.pack_conv(wl8, bl8, None, 1, 1, device)
            p["trunk"] = [[[pc(f"RRDB_trunk.{t}.RDB{r}.conv{c}") for c in range(1, 6)] for r in (1, 2, 3)]
                          for t in range(self.NUM_BLOCKS)]
            self._p = p
        self.precision = E.resolve_precision(precision)

    @torch.no_grad()
    def selfcheck(self, sd=None, image_u8: torch.Tensor | None = None, rel_tol: float = 1e-4):
        """Range / accuracy guard of the fp16x3 path (see ``RetinaFace.selfcheck``): one small calibration image through the
        351 convs with a max-|x| reduction behind every launch (``FloatingPointError`` at 2^15), and — with the state dict —
        the x4 output against an exact-fp32 twin."""
        if self.precision != 1:
            self.selfcheck_report = {"skipped": "exact-fp32 path: nothing to guard"}
            return self.selfcheck_report
        dev = self.device
        with torch.cuda.device(dev):
            if image_u8 is None:
                g = torch.Generator(device="cpu").manual_seed(20260928)
                image_u8 = torch.randint(0, 256, (1, 48, 64, 3), generator=g, dtype=torch.uint8)
                image_u8[0, :24] = (torch.arange(64)[None, :, None] * 4).clamp(max=255).to(torch.uint8)   # half ramp, half noise
            x4 = E.u8_to_nhwc4(image_u8.to(dev).contiguous(), sub=(0.0, 0.0, 0.0), div=255.0)
            tuning, E.Autotune.enabled = E.Autotune.enabled, False
            try:
                with E.RangeMonitor() as mon:
                    y = self.forward(x4)
                rep = {"launch_absmax": mon.check("RRDBNet"), "limit": E.RangeMonitor.LIMIT, "output_rel_diff": None}
                if sd is not None:
                    twin = RRDBNet(self.min_face_factor).load(dev, sd, "f32")
                    ref = twin.forward(x4)
                    rep["output_rel_diff"] = E.selfcheck_compare("RRDBNet x4 output", y.buf[..., :3], ref.buf[..., :3], rel_tol)
                    del twin, ref                                        # the exact-fp32 twin (a second copy of the 351 filters
                    torch.cuda.empty_cache()                             # on the device) lives for this comparison only
            finally:
                E.Autotune.enabled = tuning
        self.selfcheck_report = rep
        return rep

    def forward(self, x4: E.Act) -> E.Act:
        """NHWC4 image in [0,1] (n,h,w,4) -> (n,4h,4w,4) with the RGB output in channels 0..2."""
        p = self._p
        n, h, w, dev = x4.n, x4.h, x4.w, x4.buf.device
        f = 1 if self.precision == 1 else 0     # fp16x3: trunk activations in split32 (operands are DMA copies)
        bufs = [E.Act.empty(n, h, w, 192, dev, f) for _ in range(3)]  # one dense-block concat buffer per RDB
        fea0 = E.conv(p["conv_first"], x4, out_fmt=f)                  # kept for the trunk residual
        E.conv(p["conv_first"], x4, bufs[0].slice(0, 64))              # and as x of the first RDB
        band = self.TRUNK_BAND if 0 < self.TRUNK_BAND < h else 0
        for t, rrdb in enumerate(p["trunk"]):
            for r, convs in enumerate(rrdb):
                b, nxt = bufs[r], bufs[(r + 1) % 3]
                x_rrdb = bufs[0] if r == 2 else None                   # RDB3 also applies the RRDB's own residual
                if band:
                    self._dense_block_banded(convs, b, nxt, x_rrdb, band)
                    continue
                for c in range(4):                                     # x_{c+1} = lrelu(conv(cat(x..x_c)))
                    E.conv(convs[c], b.slice(0, 64 + 32 * c), b.slice(64 + 32 * c, 32), act_slope=0.2)
                if r < 2:                                              # x5*0.2 + x -> next RDB's x
                    E.conv(convs[4], b, nxt.slice(0, 64), alpha=0.2, res1=b.slice(0, 64), res1_pre=False)
                else:                                                  # (x5*0.2 + x)*0.2 + x_rrdb
                    E.conv(convs[4], b, nxt.slice(0, 64), alpha=0.2, res1=b.slice(0, 64), res1_pre=False,
                           res2=bufs[0].slice(0, 64), alpha2=0.2)
        fea = E.conv(p["trunk_conv"], bufs[0].slice(0, 64), res1=fea0, res1_pre=False, out_fmt=f)
        del bufs
        fea = E.conv(p["upconv1"], fea, act_slope=0.2, in_up2=True, out_fmt=f)
        return self._tail(fea, f)

    # Rows per band of the band-major dense-block schedule (0 = every conv over the whole image).
    TRUNK_BAND = int(os.environ.get("FCP_RRDB_BAND", "0"))

    @staticmethod
    def _dense_block_banded(convs, b: E.Act, nxt: E.Act, x_rrdb: E.Act | None, band: int):
        """One residual dense block (_layers.py:168-200 of the reference) band-major: conv1 .. conv5 on one band of rows
        before the next band, so that the block's growing concat (192 channels x 4 B x W x band: 108 MB for 128 rows of a
        1024-wide image) is re-read from the memory-side cache instead of from HBM (whole-image launches re-read 20 channel
        slices of 268 MB per block).  Rows [r0, r1) of conv5 need rows [r0 - m, r1 + m) of conv(5 - m): each conv computes
        exactly those rows of the whole-image convolution from an input view that holds one real row above and below
        (``band`` of ``engine.conv``; at the image's edges the view ends there and the zero padding is the right one).  The
        4 + 3 + 2 + 1 margin rows per side are computed twice (+3 % of the block's work at 128 rows) and give the same
        values both times: bit-identical to the whole-image schedule (tests/test_parse_enhance_gpu.py)."""
        n, h = b.n, b.h
        rows = lambda t, i, a, e: E.Act(t.buf[i:i + 1, a:e], t.c0, t.c, t.fmt)
        for i in range(n):
            for r0 in range(0, h, band):
                r1 = min(h, r0 + band)
                for c in range(5):
                    m = 4 - c
                    oa, ob = max(0, r0 - m), min(h, r1 + m)             # output rows of this conv
                    ia, ib = max(0, oa - 1), min(h, ob + 1)             # input rows it reads
                    bd = (oa - ia, ib - ob)
                    x = rows(b.slice(0, 64 + 32 * c), i, ia, ib)
                    if c < 4:
                        E.conv(convs[c], x, rows(b.slice(64 + 32 * c, 32), i, oa, ob), act_slope=0.2, band=bd)
                    elif x_rrdb is None:
                        E.conv(convs[4], x, rows(nxt.slice(0, 64), i, oa, ob), alpha=0.2, res1=rows(b.slice(0, 64), i, oa, ob),
                               res1_pre=False, band=bd)
                    else:
                        E.conv(convs[4], x, rows(nxt.slice(0, 64), i, oa, ob), alpha=0.2, res1=rows(b.slice(0, 64), i, oa, ob),
                               res1_pre=False, res2=rows(x_rrdb.slice(0, 64), i, oa, ob), alpha2=0.2, band=bd)

    TAIL_BAND = int(os.environ.get("FCP_RRDB_TAIL_BAND", "256"))    # x4-resolution output rows per band

    def _tail(self, fea2: E.Act, f: int) -> E.Act:
        """upconv2(up2(fea2)) -> HRconv -> conv_last over bands of x4-resolution rows (rrdb.py:72-74 of the reference).

        A band [r0, r1) of output rows needs HRconv rows [r0-1, r1+1), upconv2 rows [r0-2, r1+2) and the x2-resolution
        rows [(r0-3)/2, ..): every band runs the three convs on a view with >= 4 rows of margin on interior edges, where
        the zero padding a conv applies at the edge of its VIEW only spoils rows that are thrown away; at the true image
        border the view ends there and the zero padding is the right one.  Rows [r0, r1) of the last conv are copied out."""
        p = self._p
        n, h2, w2, dev = fea2.n, fea2.h, fea2.w, fea2.buf.device
        h4, w4, band = 2 * h2, 2 * w2, max(8, self.TAIL_BAND & ~1)
        out = E.Act.empty(n, h4, w4, 4, dev)
        rows_max = min(h4, band + 10)
        sa = torch.empty((1, rows_max, w4, 64), dtype=torch.float32, device=dev)
        sb = torch.empty((1, rows_max, w4, 64), dtype=torch.float32, device=dev)
        sc = torch.empty((1, rows_max, w4, 8), dtype=torch.float32, device=dev)
        for i in range(n):
            for r0 in range(0, h4, band):
                r1 = min(h4, r0 + band)
                a, b = max(0, r0 // 2 - 2), min(h2, (r1 + 1) // 2 + 2)
                rows = 2 * (b - a)
                src = E.Act(fea2.buf[i:i + 1, a:b], fea2.c0, fea2.c, fea2.fmt)
                ta, tb, tc = (E.Act(t[:, :rows], fmt=ff) for t, ff in ((sa, f), (sb, f), (sc, 0)))
                E.conv(p["upconv2"], src, ta, act_slope=0.2, in_up2=True)
                E.conv(p["HRconv"], ta, tb, act_slope=0.2)
                E.conv(p["conv_last"], tb, tc)
                out.buf[i, r0:r1, :, :3].copy_(sc[0, r0 - 2 * a:r1 - 2 * a, :, :3])
        return out

    def enhance_u8(self, images_u8: torch.Tensor, which) -> torch.Tensor:
        """Enhance in place the listed images of a (n,h,w,3) uint8 device batch."""
        n, h, w, _ = images_u8.shape
        for i in which:
            x4 = E.u8_to_nhwc4(images_u8[i:i + 1], sub=(0.0, 0.0, 0.0), div=255.0)   # `.div(255)`, rrdb.py:142
            y = self.forward(x4)
            N.check(N.lib().fcp_bicubic_down4_u8(y.ptr(), h, w, y.ld, N.ptr(images_u8, i * h * w * 3),
                                                 N.stream_ptr()), "fcp_bicubic_down4_u8")
        return images_u8

    def gate(self, n_images, h, w, landmarks, indices):
        """rrdb.py:124-140: enhance iff mean((x4-x0)*(y4-y0)/(H*W)) <= min_face_factor."""
        out = []
        for i in range(n_images):
            if landmarks is None or indices is None:
                out.append(i)
                continue
            lm = landmarks[[idx == i for idx in indices]]
            if len(lm) == 0:
                continue
            wv, hv = (lm[:, 4] - lm[:, 0]).T
            if (wv * hv / (h * w)).mean() <= self.min_face_factor:
                out.append(i)
        return out

    @torch.no_grad()
    def predict(self, images, landmarks: np.ndarray | None, indices: list | None):
        """Reference signature (rrdb.py:84-146).  ``images``: (N,H,W,3) uint8 device batch (fast
        path, modified in place and returned) or (N,3,H,W) float 0..255 (returned as float)."""
        with torch.cuda.device(self.device):
            if isinstance(images, torch.Tensor) and images.dtype == torch.uint8:
                images = images.to(self.device)
                return self.enhance_u8(images, self.gate(len(images), images.shape[1], images.shape[2],
                                                         landmarks, indices))
            as_list = isinstance(images, list)
            outs = []
            h0, w0 = images[0].shape[1], images[0].shape[2]
            todo = set(self.gate(len(images), h0, w0, landmarks, indices))
            for i in range(len(images)):
                img = images[i]
                if i in todo:
                    u8 = img.permute(1, 2, 0).to(self.device).to(torch.uint8).contiguous()[None]
                    img = self.enhance_u8(u8, [0])[0].permute(2, 0, 1).float().to(img.device)
                outs.append(img)
            return outs if as_list else torch.stack(outs)
