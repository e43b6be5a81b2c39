"""Achieved HBM GB/s of the bandwidth-bound (non-conv) kernels at bench-like sizes — profiling helper.
Bytes = algorithmic bytes read + written."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from face_crop_plus_amd import engine as E, align, _native as N
from face_crop_plus_amd.batch import build_batch
from face_crop_plus_amd.cropper import landmarks_target
dev = torch.device("cuda:0")

def t(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def report(name, ms, nbytes):
    print(f"{name:34s} {ms * 1e3:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)

img = torch.randint(0, 256, (64, 640, 640, 3), dtype=torch.uint8, device=dev)
report("u8_to_nhwc4 (64x640^2)", t(lambda: E.u8_to_nhwc4(img, sub=(123.0, 117.0, 104.0))), img.numel() + img.numel() // 3 * 16)
x = E.Act.empty(64, 320, 320, 64, dev, 1); x.buf.normal_()
report("maxpool3x3s2 split32 (320^2x64)", t(lambda: E.maxpool3x3s2(x)), x.buf.numel() * 4 * 1.25)
xf = E.Act(torch.randn(64, 160, 160, 64, device=dev))
report("f32_to_split32 (160^2x64)", t(lambda: E.f32_to_split32(xf)), xf.buf.numel() * 8)
# warp: 4096 crops of 256^2 from 64 images (bytes: crops written + ~source footprint read)
F = 4096
lm = torch.tensor(landmarks_target((256, 256), 0.65)).repeat(F, 1, 1) * 1.7 + 60 + torch.rand(F, 1, 2) * 200
idx = torch.arange(F, dtype=torch.int32) % 64
tgt = landmarks_target((256, 256), 0.65)
ms = t(lambda: align.crop_align(img, idx, lm.to(dev), tgt, (256, 256)))
report("estimate + warp_affine (4096 crops)", ms, F * 256 * 256 * 3 * (1 + 1.7 ** 2))
# batch builder: 16 4K frames -> 1024^2
frames = [np.random.default_rng(i).integers(0, 256, (2160, 3840, 3), dtype=np.uint8) for i in range(4)] * 4
blob_bytes = sum(f.size for f in frames)
from face_crop_plus_amd.batch import ITEM_DTYPE, batch_geometry
items = np.zeros(len(frames), ITEM_DTYPE); off = 0
for i, f in enumerate(frames):
    ww, hh, pad, _, interp = batch_geometry(2160, 3840, (1024, 1024)); items[i] = (off, 2160, 3840, hh, ww, pad[0], pad[2], interp, 0); off += f.size
blob = torch.from_numpy(np.concatenate([f.reshape(-1) for f in frames])).to(dev)
items_dev = torch.from_numpy(items.view(np.uint8)).to(dev)
out = torch.empty((len(frames), 1024, 1024, 3), dtype=torch.uint8, device=dev)
ms = t(lambda: N.check(N.lib().fcp_build_batch_u8(N.ptr(blob), off, items.ctypes.data, N.ptr(items_dev), len(frames), 1024, 1024, 0, N.ptr(out), N.stream_ptr())))
report("build_batch INTER_AREA (16 4K->1024)", ms, blob_bytes + out.numel())
# RRDB tail: bicubic x0.25
y = E.Act(torch.rand(1, 4096, 4096, 4, device=dev)); o = torch.empty((1024, 1024, 3), dtype=torch.uint8, device=dev)
ms = t(lambda: N.check(N.lib().fcp_bicubic_down4_u8(y.ptr(), 1024, 1024, 4, N.ptr(o), N.stream_ptr())))
report("bicubic_down4 (4096^2 -> 1024^2)", ms, 4096 * 4096 * 16 + o.numel())
