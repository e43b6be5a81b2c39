"""``torch.ops.fcp.*``: the TORCH_LIBRARY registration over the C ABI (SURVEY.md 8b / north_star "custom ops")."""
import os

import numpy as np
import pytest
import torch


def test_ops_are_registered_with_a_gpu_kernel():
    """CPU-only: the op library loads, every op has a schema and a CUDA(HIP)-key kernel, and a CPU tensor is refused."""
    from face_crop_plus_amd import torch_ops as T
    ops = T.load()
    for name in T.OPS:
        assert hasattr(ops, name), name
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"fcp::{name}", "CUDA"), name
    schema = str(torch.ops.fcp.warp_affine_u8.default._schema)
    assert "Tensor images" in schema and "int border" in schema
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.bicubic_down4_round(torch.zeros(8, 8, 4))             # no CPU implementation: there is no fallback


@pytest.mark.gpu
def test_detector_and_align_through_ops_equal_ctypes_path(device):
    """The default boundary (FCP_BOUNDARY=auto | torch) routes every conv launch, the fused bottleneck chain, decode / NMS /
    gather and estimate + warp through the registered ops: all outputs must equal the ctypes path bit for bit."""
    from face_crop_plus_amd import weights, align, torch_ops as T
    from face_crop_plus_amd.retinaface import RetinaFace
    from face_crop_plus_amd.cropper import landmarks_target
    det = RetinaFace("all", 0.55).load(device, weights.generate_state_dict("retinaface"))
    g = torch.Generator().manual_seed(4)
    img = torch.randint(0, 256, (4, 160, 200, 3), generator=g, dtype=torch.uint8).to(device)
    pads = torch.tensor([[0, 0, 0, 0], [3, 2, 0, 0], [0, 0, 4, 1], [0, 0, 0, 0]], dtype=torch.int32)
    tgt = landmarks_target((64, 48), 0.65)
    res = {}
    prev = T.ENABLED
    for mode in (False, True):
        T.ENABLED = mode
        try:
            r = det.detect(img, paddings=pads)
            nf = int(r["face_offset"][-1])
            crops, ok, mat = align.crop_align(img, r["img_idx"][:nf], r["landmarks"][:nf], tgt, (64, 48), 2, False, pads)
            torch.cuda.synchronize()
            res[mode] = (r, crops, ok, mat, nf)
        finally:
            T.ENABLED = prev
    (a, ca, oa, ma, na), (b, cb, ob, mb, nb) = res[False], res[True]
    assert na == nb and na > 4
    for k in ("landmarks", "img_idx", "face_offset", "cand_count", "keep_count", "sel_count"):
        assert torch.equal(a[k], b[k]), k
    for i, c in enumerate(a["cand_count"].tolist()):          # compacted arrays: rows beyond the count are never written
        for k in ("cand_prior", "cand_score", "cand_box", "cand_ldm"):
            assert torch.equal(a[k][i, :c], b[k][i, :c]), (k, i)
    for ha, hb in zip(a["heads"], b["heads"]):
        assert torch.equal(ha.buf, hb.buf)
    assert torch.equal(ca, cb) and torch.equal(oa, ob) and torch.equal(ma.view(-1), mb.reshape(-1))


@pytest.mark.gpu
def test_tail_ops_equal_ctypes(device):
    from face_crop_plus_amd import _native as N, torch_ops as T
    ops = T.load()
    g = torch.Generator().manual_seed(1)
    x4 = (torch.rand(64, 48, 4, generator=g) * 1.4 - 0.2).to(device)
    out = torch.empty((16, 12, 3), dtype=torch.uint8, device=device)
    N.check(N.lib().fcp_bicubic_down4_u8(N.ptr(x4), 16, 12, 4, N.ptr(out), N.stream_ptr()))
    assert torch.equal(ops.bicubic_down4_round(x4), out)
    logits = torch.randn(2, 64, 64, 32, generator=g).to(device)
    labels = torch.empty((2, 100, 80), dtype=torch.uint8, device=device)
    counts = torch.empty((2, 19), dtype=torch.int32, device=device)
    N.check(N.lib().fcp_parse_tail(N.ptr(logits), 2, 64, 64, 32, 19, 512, 512, 100, 80, N.ptr(labels), N.ptr(counts), N.stream_ptr()))
    l2, c2 = ops.parse_argmax_hist(logits, 19, 512, 512, 100, 80)
    assert torch.equal(l2, labels) and torch.equal(c2, counts)
    with pytest.raises(RuntimeError, match="dtype"):
        ops.bicubic_down4_round(x4.double())


@pytest.mark.gpu
def test_ops_reject_bad_shapes_and_declare_aliasing(device):
    """Misuse gives RuntimeError (TORCH_CHECK), not an out-of-bounds device read; the in-place variant of the conv op
    declares its write, the allocating one returns a fresh tensor."""
    from face_crop_plus_amd import torch_ops as T
    ops = T.load()
    assert "Tensor(a!) out" in str(torch.ops.fcp.conv2d_out.default._schema)
    assert "(a!)" not in str(torch.ops.fcp.conv2d.default._schema)
    h = [torch.zeros(2, -(-96 // s), -(-128 // s), 32, device=device) for s in (8, 16, 32)]
    ops.retina_decode(h[0], h[1], h[2], 96, 128, 0.6, 0.1, 0.2)
    with pytest.raises(RuntimeError, match="head map of stride 16"):
        ops.retina_decode(h[0], h[2], h[2], 96, 128, 0.6, 0.1, 0.2)
    with pytest.raises(RuntimeError, match="head map of stride 8"):
        ops.retina_decode(h[0], h[1], h[2], 192, 128, 0.6, 0.1, 0.2)           # heads too small for this image size
    score = torch.zeros(2, 100, device=device)
    count = torch.zeros(2, dtype=torch.int32, device=device)
    with pytest.raises(RuntimeError, match="cand_box"):
        ops.nms_select(score, torch.zeros(2, 50, 4, device=device), count, 0.4, 0)
    with pytest.raises(RuntimeError, match="cand_count"):
        ops.nms_select(score, torch.zeros(2, 100, 4, device=device), count[:1], 0.4, 0)
    sel = torch.zeros(2, 100, dtype=torch.int32, device=device)
    with pytest.raises(RuntimeError, match="cand_ldm"):
        ops.gather_faces(torch.zeros(2, 50, 10, device=device), sel, count, None, 4)
    with pytest.raises(RuntimeError, match="max_faces"):
        ops.gather_faces(torch.zeros(2, 100, 10, device=device), sel, count, None, 0)
    with pytest.raises(RuntimeError, match="mat must be"):
        ops.warp_affine_u8(torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device=device), torch.zeros(2, dtype=torch.int32, device=device),
                           torch.zeros(1, 2, 3, dtype=torch.float64, device=device), None, None, 4, 4, 0)
