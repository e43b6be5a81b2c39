"""GPU batch builder (mirror of the reference's ``utils.as_batch``, utils.py:273-342).

The ragged list of decoded images is packed into one pinned staging blob, copied to
the device once, and resized + padded into the (N,H,W,3) uint8 batch by a single
launch of ``fcp_build_batch_u8`` (OpenCV's INTER_AREA / INTER_CUBIC uint8 arithmetic
restated, see ``csrc/fcp_batch.hip``).  The batch stays on the device: the detector,
the enhancer and the warp all read it there, so no resized pixel crosses PCIe.
"""
from __future__ import annotations

import threading

import numpy as np
import torch

from . import _native as N
from .align import border_code

ITEM_DTYPE = np.dtype([("src_off", "<i8"), ("sh", "<i4"), ("sw", "<i4"), ("dh", "<i4"), ("dw", "<i4"),
                       ("top", "<i4"), ("left", "<i4"), ("interp", "<i4"), ("reserved", "<i4")])
assert ITEM_DTYPE.itemsize == 40                                   # sizeof(fcp_batch_item)

_tls = threading.local()


def batch_geometry(h: int, w: int, size):
    """utils.py:316-331 for one image -> (ww, hh, [t,b,l,r], unscale, interp 0 cubic | 1 area)."""
    m = max(h, w)
    interp = 1 if m > max(size) else 0
    ratio_w, ratio_h = size[0] / w, size[1] / h
    if ratio_w < ratio_h:
        unscale = ratio_w
        ww, hh = size[0], int(h * ratio_w)
        padding = [(size[1] - hh) // 2, (size[1] - hh + 1) // 2, 0, 0]
    else:
        unscale = ratio_h
        ww, hh = int(w * ratio_h), size[1]
        padding = [0, 0, (size[0] - ww) // 2, (size[0] - ww + 1) // 2]
    return ww, hh, padding, unscale, interp


def _staging(nbytes: int) -> torch.Tensor:
    """Per-thread pinned staging buffer (process_dir's ThreadPool workers build batches concurrently)."""
    buf = getattr(_tls, "buf", None)
    ev = getattr(_tls, "event", None)
    if ev is not None:
        ev.synchronize()                                           # previous upload has left the buffer
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        _tls.buf = buf
    return buf


def build_batch(images, size=512, padding_mode: str = "constant", device="cuda:0", pinned=None):
    """list of (h,w,3) uint8 RGB arrays -> (batch (N,H,W,3) uint8 *device* tensor,
    unscales (N,) float64, paddings (N,4) int64 [t,b,l,r]).  ``pinned``: optional per-image flags — True for an array
    that already lives in page-locked memory registered with HIP (the decode workers' shared-memory rings,
    ``_io_pool.py``): it is uploaded straight from where it is; the others are packed into the pinned staging blob
    first.  The caller keeps such arrays alive and unchanged until it has synchronised with the stream."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("face_crop_plus_amd runs on an AMD GPU only; there is no CPU fallback")
    lib = N.lib()
    size = (size, size) if isinstance(size, int) else tuple(size)
    n = len(images)
    items = np.zeros(n, ITEM_DTYPE)
    unscales, paddings, off = [], [], 0
    for i, image in enumerate(images):
        if image.ndim != 3 or image.shape[2] != 3 or image.dtype != np.uint8:
            raise ValueError(f"image {i}: expected an (h,w,3) uint8 array, got {image.shape} {image.dtype}")
        h, w = image.shape[:2]
        ww, hh, padding, unscale, interp = batch_geometry(h, w, size)
        if ww < 1 or hh < 1:
            raise ValueError(f"image {i} ({w}x{h}) collapses to {ww}x{hh} at batch size {size}")  # cv2.resize asserts
        items[i] = (off, h, w, hh, ww, padding[0], padding[2], interp, 0)
        off += h * w * 3
        unscales.append(np.array(unscale))
        paddings.append(np.array(padding))
    with torch.cuda.device(device):
        out = torch.empty((n, size[1], size[0], 3), dtype=torch.uint8, device=device)
        if n == 0:
            return out, np.zeros((0,)), np.zeros((0, 4), np.int64)
        blob = torch.empty(off, dtype=torch.uint8, device=device)
        direct = [bool(pinned[i]) and images[i].flags.c_contiguous for i in range(n)] if pinned is not None else [False] * n
        staged = [i for i in range(n) if not direct[i]]
        if staged:
            # staged images are packed back to back into the pinned staging blob; when ALL are staged that is the device
            # layout itself and one copy moves the lot
            stage = _staging(sum(images[i].size for i in staged))
            view, so = stage.numpy(), 0
            for i in staged:
                image = images[i]
                view[so:so + image.size] = np.ascontiguousarray(image).reshape(-1)
                if len(staged) < n:
                    o = int(items[i]["src_off"])
                    blob[o:o + image.size].copy_(stage[so:so + image.size], non_blocking=True)
                so += image.size
            if len(staged) == n:
                blob.copy_(stage[:off], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            _tls.event = ev
        for i in range(n):
            if direct[i]:                                          # DMA straight out of the registered ring
                o, image = int(items[i]["src_off"]), images[i]
                blob[o:o + image.size].copy_(torch.from_numpy(image.reshape(-1)), non_blocking=True)
        items_dev = torch.from_numpy(items.view(np.uint8)).to(device)
        N.check(lib.fcp_build_batch_u8(N.ptr(blob), off, items.ctypes.data, N.ptr(items_dev), n, size[1], size[0],
                                       border_code(padding_mode), N.ptr(out), N.stream_ptr()), "fcp_build_batch_u8")
    return out, np.stack(unscales), np.stack(paddings)
