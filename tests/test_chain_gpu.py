"""Fused bottleneck chain (conv2 -> conv3 + identity -> next conv1) vs the three stand-alone convolutions:
the fused launch must reproduce them BIT FOR BIT (same K order, same epilogue arithmetic, same hi/lo splits),
and both must match a torch fp32 reference of the three ops to fp32 roundoff."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bn(c, g):
    return dict(weight=torch.rand(c, generator=g) * 0.6 + 0.6, bias=torch.randn(c, generator=g) * 0.1,
                running_mean=torch.randn(c, generator=g) * 0.1, running_var=torch.rand(c, generator=g) + 0.5)


def _ref_bn(x, bn):
    return F.batch_norm(x, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5)


@pytest.mark.parametrize("n,h,w,cn", [(2, 37, 45, 64), (1, 16, 16, 128), (3, 64, 50, 128), (1, 5, 3, 64), (4, 160, 160, 64), (2, 8, 16, 64),
                                      (1, 9, 17, 128), (3, 23, 31, 64)])
def test_chain_equals_three_convs(n, h, w, cn, device):
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(n * 1000 + h)
    w2 = torch.randn(64, 64, 3, 3, generator=g) * (2 / 576) ** 0.5
    w3 = torch.randn(256, 64, 1, 1, generator=g) * (2 / 64) ** 0.5
    w1 = torch.randn(cn, 256, 1, 1, generator=g) * (2 / 256) ** 0.5
    bn2, bn3, bn1 = _bn(64, g), _bn(256, g), _bn(cn, g)
    bn3["weight"] = bn3["weight"] * 0.4
    t1 = F.relu(torch.randn(n, 64, h, w, generator=g))
    x = F.relu(torch.randn(n, 256, h, w, generator=g))
    with E.default_precision("f16x3"):
        pc2 = E.pack_conv(w2, None, bn2, 1, 1, device)
        pc3 = E.pack_conv(w3, None, bn3, 1, 0, device)
        pc1 = E.pack_conv(w1, None, bn1, 1, 0, device)
    assert E.chain_supported(pc2, pc3, pc1)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(device)
    t1a, xa = E.f32_to_split32(E.Act(nhwc(t1))), E.f32_to_split32(E.Act(nhwc(x)))
    # stand-alone convolutions (every tile choice of the engine gives the same bits)
    o2 = E.conv(pc2, t1a, act_slope=0.0, out_fmt=1)
    o3 = E.conv(pc3, o2, act_slope=0.0, res1=xa, res1_pre=True, out_fmt=1)
    o1 = E.conv(pc1, o3, act_slope=0.0, out_fmt=1)
    out, t1n = E.bottleneck_chain(pc2, pc3, pc1, t1a, xa)                 # default form: 8 x 16 patches, staged halo
    torch.cuda.synchronize()
    assert torch.equal(out.buf, o3.buf), "fused conv3 output differs from the stand-alone kernels"
    assert torch.equal(t1n.buf, o1.buf), "fused next-conv1 output differs from the stand-alone kernels"
    for tm in (128, 16, 256, 32):                                          # linear 4-wave tiles, 8 x 16 patches, 8-wave tiles, 16 x 16 patches: same bits
        o_t, t_t = E.bottleneck_chain(pc2, pc3, pc1, t1a, xa, tile_m=tm)
        assert torch.equal(o_t.buf, o3.buf) and torch.equal(t_t.buf, o1.buf), f"tile_m={tm}"
    # and against torch fp32
    r2 = F.relu(_ref_bn(F.conv2d(t1, w2, None, 1, 1), bn2))
    r3 = F.relu(_ref_bn(F.conv2d(r2, w3), bn3) + x)
    r1 = F.relu(_ref_bn(F.conv2d(r3, w1), bn1))
    for got, ref in ((out, r3), (t1n, r1)):
        err = (got.nchw().cpu() - ref).abs().max().item()
        assert err <= 3e-5 * float(ref.abs().max()) + 1e-6, err


def test_chain_into_channel_slices(device):
    """Outputs may be channel slices of wider buffers (ld > c); untouched channels stay untouched."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(5)
    n, h, w = 1, 20, 24
    with E.default_precision("f16x3"):
        pc2 = E.pack_conv(torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g), None, 1, 1, device)
        pc3 = E.pack_conv(torch.randn(256, 64, 1, 1, generator=g) / 8, torch.randn(256, generator=g), None, 1, 0, device)
        pc1 = E.pack_conv(torch.randn(64, 256, 1, 1, generator=g) / 16, torch.randn(64, generator=g), None, 1, 0, device)
    t1w = E.Act(torch.randn(n, h, w, 128, generator=g).to(device))
    t1w = E.f32_to_split32(t1w)
    xa = E.f32_to_split32(E.Act(torch.randn(n, h, w, 256, generator=g).to(device)))
    wide_out = E.Act(torch.full((n, h, w, 320), 7.0, device=device), fmt=1)
    wide_t1n = E.Act(torch.full((n, h, w, 128), 9.0, device=device), fmt=1)
    ref_out, ref_t1n = E.bottleneck_chain(pc2, pc3, pc1, t1w.slice(64, 64), xa)
    E.bottleneck_chain(pc2, pc3, pc1, t1w.slice(64, 64), xa, wide_out.slice(64, 256), wide_t1n.slice(64, 64))
    torch.cuda.synchronize()
    assert torch.equal(wide_out.buf[..., 64:320], ref_out.buf) and torch.equal(wide_t1n.buf[..., 64:], ref_t1n.buf)
    assert float((wide_out.buf[..., :64] - 7.0).abs().max()) == 0 and float((wide_t1n.buf[..., :64] - 9.0).abs().max()) == 0


def test_detector_same_bits_with_and_without_chain(device):
    from face_crop_plus_amd import weights
    from face_crop_plus_amd.retinaface import RetinaFace
    det = RetinaFace("all", 0.55).load(device, weights.generate_state_dict("retinaface"))
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (5, 200, 168, 3), generator=g, dtype=torch.uint8).to(device)
    assert det.fused_chain
    a = det.detect(img)
    det.fused_chain = False
    b = det.detect(img)
    torch.cuda.synchronize()
    for ha, hb in zip(a["heads"], b["heads"]):
        assert torch.equal(ha.buf, hb.buf)
    assert torch.equal(a["landmarks"], b["landmarks"]) and torch.equal(a["face_offset"], b["face_offset"])


@pytest.mark.parametrize("n,h,w,residual,c", [(2, 37, 45, True, 128), (1, 16, 16, True, 128), (3, 80, 80, True, 128), (2, 33, 29, False, 128),
                                              (1, 160, 160, False, 128), (2, 21, 19, True, 256), (1, 8, 16, True, 256), (5, 40, 40, True, 256), (2, 27, 30, True, -128), (1, 80, 80, True, -128)])
def test_pair_equals_two_convs(n, h, w, residual, c, device):
    """The pair forms (no conv2): conv3 (+ identity) + next conv1 on 128-channel inputs — layer-2 identity blocks
    (128 -> 512 -> 128, residual) and layer1.0's K-concatenated conv3 + downsample with layer1.1.conv1
    (128 -> 256 -> 64, no residual); layer-3 identity blocks on 256-channel inputs (256 -> 1024 -> 256, residual; the
    operand tile aliases the chunk buffers there).  Bit-identical to the two stand-alone convolutions."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(n * 100 + h + int(residual))
    wide_next = c < 0                                            # -128: layer 2's last block with layer3.0.conv1 (512 -> 256)
    c = abs(c)
    nout, cn = ((512, 256 if wide_next else 128) if residual else (256, 64)) if c == 128 else (1024, 256)
    w3 = torch.randn(nout, c, 1, 1, generator=g) * (2 / c) ** 0.5
    w1 = torch.randn(cn, nout, 1, 1, generator=g) * (2 / nout) ** 0.5
    bn3, bn1 = _bn(nout, g), _bn(cn, g)
    t = F.relu(torch.randn(n, c, h, w, generator=g))
    x = F.relu(torch.randn(n, nout, h, w, generator=g)) if residual else None
    with E.default_precision("f16x3"):
        pc3 = E.pack_conv(w3, None, bn3, 1, 0, device)
        pc1 = E.pack_conv(w1, None, bn1, 1, 0, device)
    assert E.chain_supported(None, pc3, pc1, residual)
    nhwc = lambda v: v.permute(0, 2, 3, 1).contiguous().to(device)
    ta = E.f32_to_split32(E.Act(nhwc(t)))
    xa = E.f32_to_split32(E.Act(nhwc(x))) if residual else None
    o3 = E.conv(pc3, ta, act_slope=0.0, res1=xa, res1_pre=True, out_fmt=1)
    o1 = E.conv(pc1, o3, act_slope=0.0, out_fmt=1)
    out, t1n = E.bottleneck_chain(None, pc3, pc1, ta, xa)
    torch.cuda.synchronize()
    assert torch.equal(out.buf, o3.buf) and torch.equal(t1n.buf, o1.buf)
    r3 = _ref_bn(F.conv2d(t, w3), bn3)
    r3 = F.relu(r3 + x) if residual else F.relu(r3)
    r1 = F.relu(_ref_bn(F.conv2d(r3, w1), bn1))
    for got, ref in ((out, r3), (t1n, r1)):
        assert (got.nchw().cpu() - ref).abs().max().item() <= 3e-5 * float(ref.abs().max()) + 1e-6


def test_chain_rejects_unsupported_shapes(device):
    from face_crop_plus_amd import engine as E
    with E.default_precision("f16x3"):
        pc3 = E.pack_conv(torch.randn(2048, 512, 1, 1), torch.zeros(2048), None, 1, 0, device)
        pc1 = E.pack_conv(torch.randn(512, 2048, 1, 1), torch.zeros(512), None, 1, 0, device)
    assert not E.chain_supported(None, pc3, pc1)


@pytest.mark.parametrize("n,h,w", [(2, 37, 45), (1, 16, 32), (3, 64, 50)])
def test_chain_out_even_only(n, h, w, device):
    """A block whose output only a stride-2 consumer reads (ResNet-50 layer1.2 -> layer2.0's 1x1 / 2 downsample) stores
    `out` at even (y, x) only: those pixels and the complete t1n carry the bits of the full launch; nothing else is written."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(n * 77 + h)
    mk = lambda co, ci, k: E.pack_conv(torch.randn(co, ci, k, k, generator=g) * (2 / (ci * k * k)) ** 0.5, torch.randn(co, generator=g) * 0.1,
                                       None, 1, k // 2, device, precision="f16x3")
    pc2, pc3, pc1 = mk(64, 64, 3), mk(256, 64, 1), mk(128, 256, 1)
    t1 = E.f32_to_split32(E.Act(torch.randn(n, h, w, 64, generator=g).relu().to(device)))
    x = E.f32_to_split32(E.Act(torch.randn(n, h, w, 256, generator=g).relu().to(device)))
    full, t1n_full = E.bottleneck_chain(pc2, pc3, pc1, t1, x)
    out = E.Act(torch.full((n, h, w, 256), -7.0, device=device), fmt=1)            # sentinel: untouched pixels keep it
    t1n = E.Act.empty(n, h, w, 128, device, 1)
    E.bottleneck_chain(pc2, pc3, pc1, t1, x, out, t1n, out_even_only=True)
    torch.cuda.synchronize()
    assert torch.equal(t1n.buf, t1n_full.buf)
    assert torch.equal(out.buf[:, ::2, ::2], full.buf[:, ::2, ::2])
    odd = torch.ones((h, w), dtype=torch.bool)
    odd[::2, ::2] = False
    assert bool((out.buf[:, odd.to(device)] == -7.0).all()), "a pixel the stride-2 consumer never reads was written"
    # the stride-2 two-source consumer gives the same bits from the sparse tensor
    pcd = mk(512, 128 + 256, 1)
    o = E.f32_to_split32(E.Act(torch.randn(n, (h + 1) // 2, (w + 1) // 2, 128, generator=g).relu().to(device)))
    a = E.conv(pcd, o, act_slope=0.0, out_fmt=1, x2=full, x2_stride=2)
    b = E.conv(pcd, o, act_slope=0.0, out_fmt=1, x2=out, x2_stride=2)
    assert torch.equal(a.buf, b.buf)


@pytest.mark.parametrize("n,h,w", [(2, 20, 24), (1, 7, 5), (3, 33, 41), (2, 80, 80), (1, 1, 1)])
def test_two_source_pair_equals_the_two_convs(n, h, w, device):
    """Round 5: layer2.0's conv3 + stride-2 downsample (one 1x1 conv over the K concatenation of TWO tensors, [conv2 out |
    x(::2, ::2)]) and layer2.1's conv1 as ONE pair launch whose operand fragments are loaded straight from the two tensors
    (no LDS tile): bit-identical to the two-source conv followed by the conv1 launch, odd sizes (x has 2h or 2h - 1 rows) and
    ragged tiles included; and both match a torch fp32 reference of bn3(conv3(o)) + bn_d(down(x)) -> relu -> conv1 -> relu."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(n * 100 + h)
    hx, wx = 2 * h - (h % 2), 2 * w - (w % 3 == 0)                      # the strided grid covers (h, w) either way
    o = F.relu(torch.randn(n, 128, h, w, generator=g))
    x = F.relu(torch.randn(n, 256, hx, wx, generator=g))
    w3 = torch.randn(512, 128, 1, 1, generator=g) * (2 / 128) ** 0.5
    wd = torch.randn(512, 256, 1, 1, generator=g) * (2 / 256) ** 0.5
    w1 = torch.randn(128, 512, 1, 1, generator=g) * (2 / 512) ** 0.5
    bn3, bnd, bn1 = _bn(512, g), _bn(512, g), _bn(128, g)
    with E.default_precision("f16x3"):
        (w3f, b3f), (wdf, bdf) = E.fold_bn(w3.numpy(), {k: v.numpy() for k, v in bn3.items()}, None), \
            E.fold_bn(wd.numpy(), {k: v.numpy() for k, v in bnd.items()}, None)
        pc3 = E.pack_conv(np.concatenate([w3f, wdf], 1), b3f + bdf, None, 1, 0, device)      # retinaface.py::_pack "c3ds"
        pc1 = E.pack_conv(w1, None, bn1, 1, 0, device)
    assert E.chain_supported(None, pc3, pc1, residual=False, cb=256)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(device)
    oa, xa = E.f32_to_split32(E.Act(nhwc(o))), E.f32_to_split32(E.Act(nhwc(x)))
    ref_out = E.conv(pc3, oa, act_slope=0.0, out_fmt=1, x2=xa, x2_stride=2)
    ref_t1n = E.conv(pc1, ref_out, act_slope=0.0, out_fmt=1)
    out, t1n = E.bottleneck_chain(None, pc3, pc1, oa, None, t1b=xa, t1b_stride=2)
    torch.cuda.synchronize()
    assert torch.equal(out.buf, ref_out.buf), "fused two-source conv3 output differs from the stand-alone kernel"
    assert torch.equal(t1n.buf, ref_t1n.buf), "fused next-conv1 output differs from the stand-alone kernel"
    again = E.bottleneck_chain(None, pc3, pc1, oa, None, t1b=xa, t1b_stride=2)
    assert torch.equal(again[0].buf, out.buf) and torch.equal(again[1].buf, t1n.buf)
    r3 = F.relu(_ref_bn(F.conv2d(o, w3), bn3) + _ref_bn(F.conv2d(x, wd, None, 2), bnd)[:, :, :h, :w])
    r1 = F.relu(_ref_bn(F.conv2d(r3, w1), bn1))
    for got, ref in ((out, r3), (t1n, r1)):
        err = (got.nchw().cpu() - ref).abs().max().item()
        assert err <= 3e-5 * float(ref.abs().max()) + 1e-6, err
    # channel-slice views of wider buffers on both sources
    wide_o, wide_x = E.Act.empty(n, h, w, 192, device, 1), E.Act.empty(n, hx, wx, 320, device, 1)
    wide_o.buf.copy_(torch.randn_like(wide_o.buf)); wide_x.buf.copy_(torch.randn_like(wide_x.buf))
    wide_o.buf[..., 64:].copy_(oa.buf); wide_x.buf[..., 32:288].copy_(xa.buf)
    o2, t2 = E.bottleneck_chain(None, pc3, pc1, wide_o.slice(64, 128), None, t1b=wide_x.slice(32, 256), t1b_stride=2)
    assert torch.equal(o2.buf, out.buf) and torch.equal(t2.buf, t1n.buf)
    # a second source whose grid does not cover the output grid is rejected by the library
    from face_crop_plus_amd import _native as N
    import ctypes as C
    small = E.Act.empty(n, max(1, (h - 1) * 2), wx, 256, device, 1)      # one row short: (h - 1) * 2 is not < its height
    d = N.ChainDesc()
    d.t1, d.out, d.t1n, d.t1b = oa.ptr(), out.ptr(), t1n.ptr(), small.ptr()
    d.w3, d.ws3, d.b3 = N.ptr(pc3.w), N.ptr(pc3.wscale), N.ptr(pc3.bias)
    d.w1n, d.ws1n, d.b1n = N.ptr(pc1.w), N.ptr(pc1.wscale), N.ptr(pc1.bias)
    d.n, d.h, d.w, d.c, d.cn, d.nout = n, h, w, 384, 128, 512
    d.t1_ld, d.out_ld, d.t1n_ld = 128, 512, 128
    d.cb, d.t1b_ld, d.t1b_h, d.t1b_w, d.t1b_stride = 256, 256, small.h, small.w, 2
    if h > 1:
        with pytest.raises(RuntimeError, match="t1b"):
            N.check(N.lib().fcp_bottleneck_chain_f16x3(C.byref(d), N.stream_ptr()), "fcp_bottleneck_chain_f16x3")


@pytest.mark.parametrize("n,h,w", [(2, 20, 24), (1, 7, 5), (3, 33, 41), (4, 40, 40), (1, 1, 1)])
def test_expand_form_equals_the_conv(n, h, w, device):
    """Round 5: conv3 + identity of a layer-3 block alone (1x1 256 -> 1024 + residual, ReLU) on the expand form of the chain
    kernel — operand fragments straight from global memory into registers, filters streamed through LDS, two workgroups per CU,
    no conv1' — is bit-identical to the stand-alone convolution with the fused residual epilogue, ragged tiles and channel-slice
    views included."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(n * 10 + w)
    t = F.relu(torch.randn(n, 256, h, w, generator=g))
    x = F.relu(torch.randn(n, 1024, h, w, generator=g))
    w3 = torch.randn(1024, 256, 1, 1, generator=g) * (2 / 256) ** 0.5
    bn3 = _bn(1024, g)
    with E.default_precision("f16x3"):
        pc3 = E.pack_conv(w3, None, bn3, 1, 0, device)
    assert E.chain_supported(None, pc3, None)
    nhwc = lambda a: a.permute(0, 2, 3, 1).contiguous().to(device)
    ta, xa = E.f32_to_split32(E.Act(nhwc(t))), E.f32_to_split32(E.Act(nhwc(x)))
    ref = E.conv(pc3, ta, act_slope=0.0, res1=xa, res1_pre=True, out_fmt=1)
    out, none = E.bottleneck_chain(None, pc3, None, ta, xa)
    torch.cuda.synchronize()
    assert none is None and torch.equal(out.buf, ref.buf)
    again, _ = E.bottleneck_chain(None, pc3, None, ta, xa)
    assert torch.equal(again.buf, out.buf)
    r = F.relu(_ref_bn(F.conv2d(t, w3), bn3) + x)
    assert (out.nchw().cpu() - r).abs().max().item() <= 3e-5 * float(r.abs().max()) + 1e-6
    wide_t, wide_o = E.Act.empty(n, h, w, 320, device, 1), E.Act.empty(n, h, w, 1088, device, 1)
    wide_t.buf.copy_(torch.randn_like(wide_t.buf)); wide_o.buf.zero_()
    wide_t.buf[..., 32:288].copy_(ta.buf)
    E.bottleneck_chain(None, pc3, None, wide_t.slice(32, 256), xa, out=wide_o.slice(64, 1024))
    assert torch.equal(wide_o.buf[..., 64:], out.buf) and wide_o.buf[..., :64].abs().max().item() == 0


def test_expand_form_through_the_registered_op_equals_ctypes(device):
    """FCP_BOUNDARY=torch: the expand form (no next conv1: w1n / ws1n / b1n = None, cn = 0) goes through ``fcp::bottleneck_chain``
    like the other chain forms — same bits as the C-ABI call; giving a conv1' filter with cn = 0 (or none with cn > 0) is refused."""
    from face_crop_plus_amd import engine as E, torch_ops as T
    g = torch.Generator().manual_seed(77)
    n, h, w = 2, 13, 17
    t = F.relu(torch.randn(n, 256, h, w, generator=g))
    x = F.relu(torch.randn(n, 1024, h, w, generator=g))
    with E.default_precision("f16x3"):
        pc3 = E.pack_conv(torch.randn(1024, 256, 1, 1, generator=g) * (2 / 256) ** 0.5, None, _bn(1024, g), 1, 0, device)
    nhwc = lambda a: a.permute(0, 2, 3, 1).contiguous().to(device)
    ta, xa = E.f32_to_split32(E.Act(nhwc(t))), E.f32_to_split32(E.Act(nhwc(x)))
    prev, outs = T.ENABLED, {}
    try:
        for mode in (False, True):
            T.ENABLED = mode
            outs[mode], none = E.bottleneck_chain(None, pc3, None, ta, xa)
            assert none is None
    finally:
        T.ENABLED = prev
    torch.cuda.synchronize()
    assert torch.equal(outs[False].buf, outs[True].buf)
    ops = T.load()
    with pytest.raises(RuntimeError, match="cn > 0"):
        ops.bottleneck_chain(ta.buf, 0, xa.buf, 0, None, None, None, pc3.w, pc3.wscale, pc3.bias, pc3.w, pc3.wscale, pc3.bias,
                             256, 1024, 0, 0, 0, None, 0, 0, 1)
    with pytest.raises(RuntimeError, match="cn > 0"):
        ops.bottleneck_chain(ta.buf, 0, xa.buf, 0, None, None, None, pc3.w, pc3.wscale, pc3.bias, None, None, None,
                             256, 1024, 256, 0, 0, None, 0, 0, 1)
