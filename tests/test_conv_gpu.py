"""HIP conv engine vs a plain PyTorch fp32 (CPU) reference of the same op."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f16x3"], autouse=True)
def precision(request):
    """Every conv test runs on both arithmetic paths (exact fp32 MFMA / split-fp16 MFMA)."""
    from face_crop_plus_amd import engine as E
    with E.default_precision(request.param):
        yield request.param


def _nhwc(x, device):  # NCHW cpu -> NHWC device
    return x.permute(0, 2, 3, 1).contiguous().to(device)


def _tol(ref):
    return 2e-5 * float(ref.abs().max()) + 1e-6


CASES = [
    # n, h, w, cin, cout, k, stride, pad
    (2, 17, 23, 64, 64, 1, 1, 0),
    (1, 40, 40, 64, 256, 1, 1, 0),
    (2, 20, 20, 256, 64, 3, 1, 1),
    (1, 33, 29, 128, 128, 3, 2, 1),
    (2, 16, 16, 256, 512, 1, 2, 0),
    (1, 12, 12, 512, 32, 1, 1, 0),
    (1, 24, 24, 96, 32, 3, 1, 1),
    (1, 24, 24, 192, 64, 3, 1, 1),
    (1, 16, 16, 256, 19, 1, 1, 0),
    (1, 32, 32, 64, 3, 3, 1, 1),
    (3, 10, 10, 256, 192, 3, 1, 1),
    (1, 64, 64, 256, 256, 3, 1, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_plain(case, device):
    from face_crop_plus_amd import engine as E
    n, h, w, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, stride, pad)
    pc = E.pack_conv(wt, b, None, stride, pad, device)
    for tile_n in ([32] if cout <= 32 else [64, 128]):
        out = E.conv(pc, E.Act(_nhwc(x, device)), tile_n=tile_n).nchw().cpu()
        assert out.shape == ref.shape
        err = (out - ref).abs().max().item()
        assert err <= _tol(ref), f"tile_n={tile_n} err={err}"


def test_conv_bn_relu_residual_pre(device):
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 19, 21, generator=g)
    wt = torch.randn(256, 64, 1, 1, generator=g) / 8
    bn = dict(weight=torch.rand(256, generator=g) + 0.5, bias=torch.randn(256, generator=g),
              running_mean=torch.randn(256, generator=g), running_var=torch.rand(256, generator=g) + 0.5)
    res = torch.randn(2, 256, 19, 21, generator=g)
    ref = F.relu(F.batch_norm(F.conv2d(x, wt), bn["running_mean"], bn["running_var"], bn["weight"],
                              bn["bias"], False, 0.0, 1e-5) + res)
    pc = E.pack_conv(wt, None, bn, 1, 0, device)
    out = E.conv(pc, E.Act(_nhwc(x, device)), act_slope=0.0, res1=E.Act(_nhwc(res, device)), res1_pre=True)
    assert (out.nchw().cpu() - ref).abs().max().item() <= _tol(ref)


def test_conv_residual_post_nearest(device):
    """FPN: lrelu(bn(conv1x1(x))) + nearest_up(res) with a non-2x size ratio."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 512, 13, 15, generator=g)
    wt = torch.randn(256, 512, 1, 1, generator=g) / 22
    res = torch.randn(1, 256, 7, 8, generator=g)
    ref = F.leaky_relu(F.conv2d(x, wt), 0.0) + F.interpolate(res, size=(13, 15), mode="nearest")
    pc = E.pack_conv(wt, None, None, 1, 0, device)
    out = E.conv(pc, E.Act(_nhwc(x, device)), act_slope=0.0, res1=E.Act(_nhwc(res, device)), res1_pre=False)
    assert (out.nchw().cpu() - ref).abs().max().item() <= _tol(ref)


def test_conv_rrdb_epilogue_and_slices(device):
    """Dense-block style: read a channel slice of a wide buffer, write another
    slice, LeakyReLU(0.2); then conv5-style ((acc+b)*0.2 + x)*0.2 + y."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(3)
    n, h, w = 1, 18, 22
    feat = torch.randn(n, 192, h, w, generator=g)
    w2 = torch.randn(32, 96, 3, 3, generator=g) / 30
    b2 = torch.randn(32, generator=g)
    buf = E.Act(_nhwc(feat, device))
    ref2 = F.leaky_relu(F.conv2d(feat[:, :96], w2, b2, 1, 1), 0.2)
    pc2 = E.pack_conv(w2, b2, None, 1, 1, device)
    E.conv(pc2, buf.slice(0, 96), buf.slice(96, 32), act_slope=0.2)
    got = buf.slice(96, 32).nchw().cpu()
    assert (got - ref2).abs().max().item() <= _tol(ref2)
    feat2 = buf.nchw().cpu()
    w5 = torch.randn(64, 192, 3, 3, generator=g) / 40
    b5 = torch.randn(64, generator=g)
    xin = feat2[:, :64]
    y = torch.randn(n, 64, h, w, generator=g)
    ref5 = (F.conv2d(feat2, w5, b5, 1, 1) * 0.2 + xin) * 0.2 + y
    pc5 = E.pack_conv(w5, b5, None, 1, 1, device)
    out = E.conv(pc5, buf, alpha=0.2, res1=buf.slice(0, 64), res1_pre=False, res2=E.Act(_nhwc(y, device)),
                 alpha2=0.2)
    assert (out.nchw().cpu() - ref5).abs().max().item() <= _tol(ref5)


def test_conv_upsampled_input(device):
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 64, 9, 11, generator=g)
    wt = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2), wt, b, 1, 1), 0.2)
    pc = E.pack_conv(wt, b, None, 1, 1, device)
    out = E.conv(pc, E.Act(_nhwc(x, device)), act_slope=0.2, in_up2=True)
    assert (out.nchw().cpu() - ref).abs().max().item() <= _tol(ref)


@pytest.mark.f16x3_only
@pytest.mark.parametrize("cin,cout,hw,n", [(64, 64, (9, 11), 1), (64, 32, (20, 37), 2), (128, 128, (12, 40), 1), (64, 64, (70, 130), 2)])
def test_conv_in_up2_halo_tiles(cin, cout, hw, n, device, precision):
    """Nearest x2 fused into the operand fetch (RRDB's upconv1 / upconv2) on the halo-tile kernels: the staged halo row of a
    logical pixel is the physical pixel (y / 2, x / 2).  Bit-identical to the implicit-GEMM tile, which has had it since round 1."""
    if precision != "f16x3":
        pytest.skip("split32 tensors exist only on the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(cin + cout + hw[0])
    h, w = hw
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2), wt, b, 1, 1), 0.2)
    pc = E.pack_conv(wt, b, None, 1, 1, device)
    xs = E.f32_to_split32(E.Act(_nhwc(x, device)))
    base = E.conv(pc, xs, act_slope=0.2, in_up2=True, tile_m=128, tile_n=64)
    assert (base.nchw().cpu() - ref).abs().max().item() <= _tol(ref)
    tiles = [(1, 32)] if cout <= 64 else []
    tiles += [(1, 64)] if 32 < cout <= 64 else ([(1, 128)] if cout > 64 else [])
    for tm, tn in tiles:
        out = E.conv(pc, xs, act_slope=0.2, in_up2=True, tile_m=tm, tile_n=tn)
        assert torch.equal(out.buf, base.buf), (tm, tn)


@pytest.mark.parametrize("k,stride,pad", [(7, 2, 3), (3, 1, 1)])
def test_conv_stem_cin4(k, stride, pad, device):
    """u8 image -> NHWC4 (mean subtraction fused) -> cin4-mode stem conv."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (2, 37, 45, 3), generator=g, dtype=torch.uint8)
    wt = torch.randn(64, 3, k, k, generator=g) / (3 * k * k) ** 0.5
    mean = torch.tensor([104.0, 117.0, 123.0])
    x = img.permute(0, 3, 1, 2).float()
    ref = F.relu(F.conv2d(x[:, [2, 1, 0]] - mean.view(3, 1, 1), wt, None, stride, pad))
    # kernel keeps RGB order: permute the filter's input channels and the means instead
    a = E.u8_to_nhwc4(img.to(device), sub=(123.0, 117.0, 104.0))
    pc = E.pack_conv(wt, None, None, stride, pad, device, cin_perm=[2, 1, 0])
    out = E.conv(pc, a, act_slope=0.0)
    assert (out.nchw().cpu() - ref).abs().max().item() <= _tol(ref)


def test_maxpool(device):
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 64, 37, 41, generator=g)
    ref = F.max_pool2d(x, 3, 2, 1)
    out = E.maxpool3x3s2(E.Act(_nhwc(x, device))).nchw().cpu()
    assert torch.equal(out, ref)


def test_split32_roundtrip_and_maxpool(device):
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 21, 19, generator=g) * 30
    a = E.Act(_nhwc(x, device))
    sp = E.f32_to_split32(a)
    back = sp.nchw().cpu()
    assert (back - x).abs().max().item() <= 2.0 ** -20 * x.abs().max().item()
    # idempotent: re-encoding a decoded tensor reproduces the same values (bytes may differ in the sign of a zero lo)
    again = E.f32_to_split32(E.split32_to_f32(sp))
    assert torch.equal(again.nchw(), sp.nchw())
    mp = E.maxpool3x3s2(sp)
    assert mp.fmt == 1 and torch.equal(mp.nchw().cpu(), F.max_pool2d(back, 3, 2, 1))


def test_split32_bits_vs_numpy(device):
    """The hi / lo split and its inverse, bit for bit against a numpy restatement (hi = x rounded TOWARD ZERO to binary16,
    lo = (x - hi) rounded toward zero; value = float(hi) + float(lo)), over normal, subnormal-binary16, tiny and large
    magnitudes: pins the arithmetic of split8 / join8 (one v_fma_mix_f32 per value) in every kernel epilogue."""
    from face_crop_plus_amd import engine as E
    rng = np.random.default_rng(5)
    mags = np.concatenate([rng.normal(0, 1, 20000), rng.normal(0, 300, 20000), rng.normal(0, 1e-4, 20000), rng.normal(0, 3e-6, 20000),
                           rng.normal(0, 1e-9, 8000), rng.uniform(-65000, 65000, 8000), [0.0, -0.0, 1.0, -1.0, 65504.0, 2.0 ** -24, 2.0 ** -14]])
    n = (len(mags) // 64) * 64
    x = mags[:n].astype(np.float32).reshape(1, 1, n // 64, 64)

    def rtz16(v):
        h = v.astype(np.float16)
        over = np.abs(h.astype(np.float32)) > np.abs(v)                 # round-to-nearest went away from zero: step back
        h[over] = np.nextafter(h[over], np.float16(0))
        return h
    hi = rtz16(x)
    lo = rtz16(x - hi.astype(np.float32))
    sp = E.f32_to_split32(E.Act(torch.from_numpy(x).to(device)))
    raw = sp.buf.cpu().numpy().view(np.uint16).reshape(1, 1, n // 64, 2, 2, 32)      # [group of 32 channels][hi | lo][32]
    # a zero lo part may carry either sign (x - hi = +0 or -0): compare values for zeros, bits otherwise
    got_hi, got_lo = raw[..., 0, :].reshape(x.shape), raw[..., 1, :].reshape(x.shape)
    assert np.array_equal(got_hi, hi.view(np.uint16))
    nz = lo != 0
    assert np.array_equal(got_lo[nz], lo.view(np.uint16)[nz]) and not got_lo.view(np.float16)[~nz].any()
    back = E.split32_to_f32(sp).buf.cpu().numpy()
    assert np.array_equal(back, hi.astype(np.float32) + lo.astype(np.float32))
    assert np.abs(back - x).max() <= 2.0 ** -21 * np.abs(x).max()


@pytest.mark.f16x3_only
@pytest.mark.parametrize("k,stride,cin,cout", [(1, 1, 64, 256), (3, 1, 128, 128), (3, 2, 64, 64), (1, 2, 256, 512)])
def test_conv_split32_in_out_with_residual(k, stride, cin, cout, device, precision):
    """split32 activations end to end: split input, split residual, split output (fp16x3 path only)."""
    if precision != "f16x3":
        pytest.skip("split32 tensors exist only on the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(k * 100 + cin)
    n, h, w = 2, 18, 22
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, stride, k // 2)
    res = torch.randn(*ref.shape, generator=g)
    ref = F.relu(ref + res)
    pc = E.pack_conv(wt, b, None, stride, k // 2, device)
    xs = E.f32_to_split32(E.Act(_nhwc(x, device)))
    rs = E.f32_to_split32(E.Act(_nhwc(res, device)))
    for tile_n in (64, 128):
        out = E.conv(pc, xs, act_slope=0.0, res1=rs, res1_pre=True, out_fmt=1, tile_n=tile_n)
        assert out.fmt == 1
        assert (out.nchw().cpu() - ref).abs().max().item() <= _tol(ref)
        # mixed: split input, fp32 residual and output
        out2 = E.conv(pc, xs, act_slope=0.0, res1=E.Act(_nhwc(res, device)), res1_pre=True, tile_n=tile_n)
        assert out2.fmt == 0 and (out2.nchw().cpu() - ref).abs().max().item() <= _tol(ref)


@pytest.mark.f16x3_only
def test_conv_split32_channel_slices(device, precision):
    """SSH-style: read a 32-aligned channel slice of a split32 buffer, write another slice of it."""
    if precision != "f16x3":
        pytest.skip("split32 tensors exist only on the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(11)
    feat = torch.randn(1, 384, 14, 17, generator=g)
    buf = E.f32_to_split32(E.Act(_nhwc(feat, device)))
    w = torch.randn(128, 64, 3, 3, generator=g) / 24
    b = torch.randn(128, generator=g)
    ref = F.relu(F.conv2d(feat[:, :64], w, b, 1, 1))
    E.conv(E.pack_conv(w, b, None, 1, 1, device), buf.slice(0, 64), buf.slice(256, 128), act_slope=0.0)
    full = buf.nchw().cpu()
    assert (full[:, 256:384] - ref).abs().max().item() <= _tol(ref)
    assert (full[:, :256] - feat[:, :256]).abs().max().item() <= 2.0 ** -20 * feat.abs().max().item()   # untouched


@pytest.mark.f16x3_only
@pytest.mark.parametrize("k,stride,cin,cout,hw", [(1, 1, 64, 256, (18, 22)), (3, 1, 128, 128, (18, 22)), (3, 2, 64, 384, (19, 21)),
                                                  (1, 2, 256, 512, (18, 22)), (1, 1, 1024, 256, (9, 7)), (3, 1, 32, 200, (16, 16))])
def test_conv_256_row_tiles(k, stride, cin, cout, hw, device, precision):
    """The 8-wave 256-row kernel (tile_m=256, tile_n 128 / 256): same operands and accumulation order as the
    128-row LDS-DMA kernel => bit-identical outputs; also checked against the fp32 reference.  Covers ragged
    M (not a multiple of 256), cout that is not a multiple of the N tile, residuals in both formats, one slice."""
    if precision != "f16x3":
        pytest.skip("split32 tensors exist only on the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(k * 1000 + cin + cout)
    n, (h, w) = 3, hw
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, stride, k // 2)
    res = torch.randn(*ref.shape, generator=g)
    ref = F.relu(ref + res)
    pc = E.pack_conv(wt, b, None, stride, k // 2, device)
    xs = E.f32_to_split32(E.Act(_nhwc(x, device)))
    rs32 = E.Act(_nhwc(res, device))
    base = E.conv(pc, xs, act_slope=0.0, res1=rs32, res1_pre=True, tile_n=128, tile_m=128)
    assert (base.nchw().cpu() - ref).abs().max().item() <= _tol(ref)
    for tile_n in (128, 256):
        out = E.conv(pc, xs, act_slope=0.0, res1=rs32, res1_pre=True, tile_n=tile_n, tile_m=256)
        assert out.fmt == 0 and torch.equal(out.buf, base.buf), tile_n
        if cout % 32 == 0:
            outs = E.conv(pc, xs, act_slope=0.0, res1=E.f32_to_split32(rs32), res1_pre=True, out_fmt=1, tile_n=tile_n,
                          tile_m=256)
            assert outs.fmt == 1 and (outs.nchw().cpu() - ref).abs().max().item() <= _tol(ref)


def test_conv_256_row_tiles_rejects_unsupported(device, precision):
    from face_crop_plus_amd import engine as E
    wt, b = torch.randn(128, 64, 1, 1), torch.zeros(128)
    pc = E.pack_conv(wt, b, None, 1, 0, device)
    x = E.Act(torch.randn(1, 8, 8, 64, device=device))
    with pytest.raises(RuntimeError, match="256-row tiles"):      # fp32-format input (and the fp32 path) are not eligible
        E.conv(pc, x, tile_m=256, tile_n=128)


@pytest.mark.f16x3_only
@pytest.mark.parametrize("c1,c2,cout,stride,hw", [(64, 64, 256, 1, (18, 22)), (128, 256, 512, 2, (20, 24)),
                                                  (256, 512, 1024, 2, (14, 10)), (32, 96, 72, 3, (19, 22))])
def test_conv_two_sources(c1, c2, cout, stride, hw, device, precision):
    """1x1 conv over the K concatenation of two tensors, the second sampled at a stride (bottleneck conv3 +
    downsample in one launch): every tile shape, vs the sum of two fp32 reference convs."""
    if precision != "f16x3":
        pytest.skip("two-source convs exist only on the fp16x3 / split32 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(c1 + c2)
    n, (h, w) = 3, hw
    H2, W2 = (h - 1) * stride + 1 + (stride > 1), (w - 1) * stride + 1 + (stride > 1)     # even sizes like ResNet's
    a = torch.randn(n, c1, h, w, generator=g)
    b = torch.randn(n, c2, H2, W2, generator=g)
    wa = torch.randn(cout, c1, 1, 1, generator=g) / (c1 + c2) ** 0.5
    wb = torch.randn(cout, c2, 1, 1, generator=g) / (c1 + c2) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(a, wa, bias) + F.conv2d(b, wb, None, stride))
    pc = E.pack_conv(torch.cat([wa, wb], 1), bias, None, 1, 0, device)
    xa = E.f32_to_split32(E.Act(_nhwc(a, device)))
    wide = E.f32_to_split32(E.Act(_nhwc(torch.cat([torch.randn(n, 32, H2, W2, generator=g), b], 1), device)))
    xb = wide.slice(32, c2)                                        # a channel slice as second source
    outs = []
    for tm, tn in ((128, 64), (128, 128), (256, 128), (256, 256)):
        if tm == 256 and cout % 8:
            continue
        out = E.conv(pc, xa, act_slope=0.0, x2=xb, x2_stride=stride, tile_m=tm, tile_n=tn)
        assert (out.nchw().cpu() - ref).abs().max().item() <= _tol(ref), (tm, tn)
        outs.append(out.buf)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    with pytest.raises(RuntimeError, match="second source"):
        pc3 = E.pack_conv(torch.randn(cout, c1 + c2, 3, 3), None, None, 1, 1, device)
        E.conv(pc3, xa, x2=xb, x2_stride=stride)


@pytest.mark.f16x3_only
@pytest.mark.parametrize("cin,cout,hw,n", [(64, 32, (16, 64), 2), (96, 32, (19, 45), 1), (160, 24, (8, 32), 3), (64, 32, (5, 7), 2),
                                           (192, 8, (33, 70), 1), (192, 64, (40, 70), 2), (64, 64, (9, 33), 1), (96, 40, (17, 20), 2),
                                           (96, 32, (256, 384), 3), (128, 64, (200, 300), 2)])
def test_conv_halo_tiles(cin, cout, hw, n, device, precision):
    """The halo-tile 3x3 kernel (tile_m=1; RRDB's 32- and 64-filter convs): bit-identical to the implicit-GEMM
    kernel; ragged patches (H % 8, W % 32 != 0), image borders, odd and even slice counts, one and two filter passes,
    residual + LeakyReLU epilogue, and images with more patches than CUs (the persistent tile loop, where the next
    tile's first DMA overlaps the epilogue)."""
    if precision != "f16x3":
        pytest.skip("split32 tensors exist only on the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(cin * 7 + cout)
    h, w = hw
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    ref = F.leaky_relu(F.conv2d(x, wt, b, 1, 1), 0.2) * 0.5 + res
    pc = E.pack_conv(wt, b, None, 1, 1, device)
    xs = E.f32_to_split32(E.Act(_nhwc(x, device)))
    rs = E.Act(_nhwc(res, device))
    tn = 32 if cout <= 32 else 64
    base = E.conv(pc, xs, act_slope=0.2, alpha=0.5, res1=rs, res1_pre=False, tile_m=128, tile_n=tn)
    assert (base.nchw().cpu() - ref).abs().max().item() <= _tol(ref)
    out = E.conv(pc, xs, act_slope=0.2, alpha=0.5, res1=rs, res1_pre=False, tile_m=1, tile_n=32)
    assert torch.equal(out.buf, base.buf)
    again = E.conv(pc, xs, act_slope=0.2, alpha=0.5, res1=rs, res1_pre=False, tile_m=1, tile_n=32)
    assert torch.equal(again.buf, out.buf)                         # no race between a tile's epilogue and the next tile's DMA
    if cout % 32 == 0:                                             # dense-block style: write a slice of a wider split32 buffer
        wide = E.Act.empty(n, h, w, 64 + cout, device, 1)
        wide.buf.zero_()
        E.conv(pc, xs, wide.slice(64, cout), act_slope=0.2, tile_m=1, tile_n=32)
        ref2 = F.leaky_relu(F.conv2d(x, wt, b, 1, 1), 0.2)
        got = wide.nchw().cpu()
        assert (got[:, 64:] - ref2).abs().max().item() <= _tol(ref2) and got[:, :64].abs().max().item() == 0
        rs2 = E.f32_to_split32(E.Act(_nhwc(torch.randn(n, cout, h, w, generator=g), device)))   # conv5-style double residual
        a5 = E.conv(pc, xs, alpha=0.2, res1=rs, res1_pre=False, res2=rs2, alpha2=0.2, out_fmt=1, tile_m=128, tile_n=tn)
        b5 = E.conv(pc, xs, alpha=0.2, res1=rs, res1_pre=False, res2=rs2, alpha2=0.2, out_fmt=1, tile_m=1, tile_n=32)
        assert torch.equal(a5.buf, b5.buf)
    with pytest.raises(RuntimeError, match="halo-tile"):
        E.conv(E.pack_conv(torch.randn(96, cin, 3, 3), None, None, 1, 1, device), xs, tile_m=1, tile_n=64)


@pytest.mark.f16x3_only
@pytest.mark.parametrize("cin,cout,hw,n", [(64, 128, (16, 64), 2), (128, 128, (19, 45), 1), (192, 96, (8, 32), 3), (64, 72, (5, 7), 2),
                                           (256, 128, (40, 70), 2), (128, 128, (200, 300), 2), (64, 64, (9, 33), 1), (128, 40, (17, 20), 2),
                                           (192, 64, (130, 260), 2)])
def test_conv_halo_wide_tiles(cin, cout, hw, n, device, precision):
    """The wide halo-tile kernel (tile_m=1, tile_n 64 / 128: pixel fragments kept across the column tiles, filters through a
    four-slot tap ring, one barrier per tap): bit-identical to the implicit-GEMM tiles and to itself on a second run (no race
    between the ring refills and the just-in-time fragment reads); 2 / 4 / 6 / 8 slices, ragged patches, more patches than
    CUs (tiles chained across the persistent loop), residual epilogues for <= 64 filters, a rejected residual above."""
    if precision != "f16x3":
        pytest.skip("split32 tensors exist only on the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(cin * 11 + cout)
    h, w = hw
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    pc = E.pack_conv(wt, b, None, 1, 1, device)
    xs = E.f32_to_split32(E.Act(_nhwc(x, device)))
    tn = 64 if cout <= 64 else 128
    ref = F.leaky_relu(F.conv2d(x, wt, b, 1, 1), 0.2) * 0.5
    for fmt in ((0, 1) if cout % 32 == 0 else (0,)):
        base = E.conv(pc, xs, act_slope=0.2, alpha=0.5, tile_m=128, tile_n=64, out_fmt=fmt)
        assert (base.nchw().cpu() - ref).abs().max().item() <= _tol(ref)
        out = E.conv(pc, xs, act_slope=0.2, alpha=0.5, tile_m=1, tile_n=tn, out_fmt=fmt)
        assert torch.equal(out.buf, base.buf)
        again = E.conv(pc, xs, act_slope=0.2, alpha=0.5, tile_m=1, tile_n=tn, out_fmt=fmt)
        assert torch.equal(again.buf, out.buf)
    res = E.Act(_nhwc(torch.randn(n, cout, h, w, generator=g), device))
    if cout <= 64:
        rs2 = E.Act(_nhwc(torch.randn(n, cout, h, w, generator=g), device))
        if cout % 32 == 0:
            rs2 = E.f32_to_split32(rs2)
        a5 = E.conv(pc, xs, alpha=0.2, res1=res, res1_pre=False, res2=rs2, alpha2=0.2, out_fmt=int(cout % 32 == 0), tile_m=128, tile_n=64)
        b5 = E.conv(pc, xs, alpha=0.2, res1=res, res1_pre=False, res2=rs2, alpha2=0.2, out_fmt=int(cout % 32 == 0), tile_m=1, tile_n=64)
        assert torch.equal(a5.buf, b5.buf)
    else:
        with pytest.raises(RuntimeError, match="wide halo-tile"):
            E.conv(pc, xs, res1=res, tile_m=1, tile_n=128)
    with pytest.raises(RuntimeError, match="wide halo-tile"):    # cin must be a multiple of 64
        E.conv(E.pack_conv(torch.randn(cout, 96, 3, 3), None, None, 1, 1, device),
               E.f32_to_split32(E.Act(_nhwc(torch.randn(1, 96, 8, 8), device))), tile_m=1, tile_n=tn)


@pytest.mark.parametrize("cin,cout,hw,tiles", [
    (64, 32, (40, 70), [(128, 32), (1, 32)]),                     # halo kernel, one pass
    (192, 64, (37, 45), [(128, 64), (1, 32), (1, 64)]),           # halo kernel two passes, wide halo (64 filters, residuals)
    (128, 128, (33, 64), [(128, 128), (128, 64), (1, 128), (256, 128)]),   # wide halo, 256-row kernel
    (64, 256, (70, 40), [(128, 128), (256, 256)]),
])
def test_conv_row_bands(cin, cout, hw, tiles, device, precision):
    """``band`` (fcp_conv_desc.band_top / band_bottom): rows [a, b) of a 3x3 conv computed from the row view [a - 1, b + 1) of
    its input (one real row above / below instead of the zero padding at the view's edge; at the image's edges the view ends
    there) are the bits of the whole-image launch, for every kernel family, with residual views of the same rows."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(cin + cout)
    h, w = hw
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    res = torch.randn(1, cout, h, w, generator=g)
    pc = E.pack_conv(wt, b, None, 1, 1, device)
    xa, ra = E.Act(_nhwc(x, device)), E.Act(_nhwc(res, device))
    split = precision == "f16x3"
    if split:
        xa = E.f32_to_split32(xa)
    rows = lambda t, a, e: E.Act(t.buf[:, a:e], t.c0, t.c, t.fmt)
    for tm, tn in (tiles if split else [(128, 64)]):
        with_res = not (tm == 1 and tn == 128)                     # the wide halo kernel takes no residual above 64 filters
        kw = dict(act_slope=0.2, alpha=0.5, tile_m=tm, tile_n=tn, out_fmt=int(split and cout % 32 == 0))
        whole = E.conv(pc, xa, res1=ra if with_res else None, res1_pre=False, **kw)
        got = E.Act.empty(1, h, w, cout, device, whole.fmt)
        got.buf.fill_(float("nan"))
        for a, e in ((0, 9), (9, 10), (10, 27), (27, h)):          # bands of 9, 1, 17 and the rest
            ia, ib = max(0, a - 1), min(h, e + 1)
            E.conv(pc, rows(xa, ia, ib), rows(got, a, e), res1=rows(ra, a, e) if with_res else None, res1_pre=False,
                   band=(a - ia, ib - e), **kw)
        assert torch.equal(got.buf, whole.buf), (tm, tn)
    with pytest.raises(RuntimeError, match="band_top"):
        E.conv(pc, rows(xa, 0, 12), band=(2, 0))


@pytest.mark.f16x3_only
@pytest.mark.parametrize("n,h,w", [(2, 77, 91), (1, 64, 64), (3, 50, 130), (1, 9, 200), (2, 640, 640)])
def test_fused_stem_pool(n, h, w, device, precision):
    """uint8 -> (x - mean) -> 7x7/2 conv + BN + ReLU -> max-pool 3x3/2 in one launch (RetinaFace stem, fp16x3 path):
    vs torch fp32 ops; odd sizes, partial patches, image borders, both output formats, slice output."""
    if precision != "f16x3":
        pytest.skip("the fused stem belongs to the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(n * 1000 + h + w)
    img = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12
    bn = {"weight": torch.rand(64, generator=g) + 0.5, "bias": torch.randn(64, generator=g) * 0.1,
          "running_mean": torch.randn(64, generator=g) * 0.1, "running_var": torch.rand(64, generator=g) + 0.5}
    mean = (123, 117, 104)
    x = img.permute(0, 3, 1, 2).float() - torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    y = F.conv2d(x, wt, None, 2, 3)
    y = F.batch_norm(y, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, 1e-5)
    ref = F.max_pool2d(F.relu(y), 3, 2, 1)
    ps = E.pack_stem_fused(wt, bn, device)
    out = E.stem_relu_pool_u8(ps, img.to(device), mean_rgb=mean, out_fmt=0)
    assert out.fmt == 0 and tuple(out.nchw().shape) == tuple(ref.shape)
    assert (out.nchw().cpu() - ref).abs().max().item() <= _tol(ref)
    wide = E.Act.empty(n, ref.shape[2], ref.shape[3], 128, device, 1)
    wide.buf.zero_()
    E.stem_relu_pool_u8(ps, img.to(device), wide.slice(64, 64), mean_rgb=mean)
    got = wide.nchw().cpu()
    assert (got[:, 64:] - ref).abs().max().item() <= _tol(ref) and got[:, :64].abs().max().item() == 0
    # same thing through the separate kernels of the generic path
    pc = E.pack_conv(wt, None, bn, 2, 3, device)
    sep = E.maxpool3x3s2(E.conv(pc, E.u8_to_nhwc4(img.to(device), sub=mean), act_slope=0.0, out_fmt=1))
    assert (sep.nchw().cpu() - out.nchw().cpu()).abs().max().item() <= _tol(ref)


@pytest.mark.parametrize("n,h,w", [(1, 64, 64), (2, 75, 131), (3, 160, 96), (1, 9, 11), (2, 128, 640)])
def test_fused_stem_pool_conv1(n, h, w, device):
    """The stem launch continued by layer1.0.conv1 (1x1 64 -> 64 + BN + ReLU on the pooled map): the pooled map and t1 must
    have the bits of the two separate launches (same hi / lo operand bytes, same K and term order, same epilogue expressions);
    odd sizes, partial patches, a slice output for the pooled map."""
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(n * 977 + h * 3 + w)
    img = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8).to(device)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12
    bn = lambda c: {"weight": torch.rand(c, generator=g) + 0.5, "bias": torch.randn(c, generator=g) * 0.1,
                    "running_mean": torch.randn(c, generator=g) * 0.1, "running_var": torch.rand(c, generator=g) + 0.5}
    ps = E.pack_stem_fused(wt, bn(64), device)
    pc1 = E.pack_conv(torch.randn(64, 64, 1, 1, generator=g) / 8, None, bn(64), 1, 0, device, precision="f16x3")
    assert E.stem_conv1_supported(pc1)
    hp, wp = ((h - 1) // 2) // 2 + 1, ((w - 1) // 2) // 2 + 1
    cat_a, cat_b = E.Act.empty(n, hp, wp, 128, device, 1), E.Act.empty(n, hp, wp, 128, device, 1)
    cat_a.buf.zero_(); cat_b.buf.zero_()
    pooled = E.stem_relu_pool_u8(ps, img, cat_a.slice(64, 64))
    t1_sep = E.conv(pc1, pooled, act_slope=0.0, out_fmt=1)
    pooled_f, t1_f = E.stem_relu_pool_u8(ps, img, cat_b.slice(64, 64), conv1=pc1)
    torch.cuda.synchronize()
    assert torch.equal(cat_a.buf, cat_b.buf), "pooled stem map differs"
    assert t1_f.fmt == 1 and torch.equal(t1_f.buf, t1_sep.buf), "fused conv1 differs from the separate launch"
    assert float(t1_f.nchw().abs().max()) > 0


def test_conv_randomized_shapes(device, precision):
    """Seeded sweep over geometry / epilogue / format combinations (every tile the shape admits) against torch fp32."""
    from face_crop_plus_amd import engine as E
    rng = np.random.default_rng(2024)
    f16 = precision == "f16x3"
    for it in range(36):
        k = int(rng.choice([1, 3, 5]))
        stride = int(rng.choice([1, 2]))
        pad = int(rng.choice([0, k // 2]))
        cin = int(rng.choice([32, 64, 96, 160, 256]))
        cout = int(rng.choice([8, 24, 32, 64, 72, 128, 192, 256, 320]))
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(k, 40)), int(rng.integers(k, 40))
        g = torch.Generator().manual_seed(it)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) if rng.random() < 0.7 else None
        ref = F.conv2d(x, wt, b, stride, pad)
        use_res, pre, slope, alpha = rng.random() < 0.5, bool(rng.random() < 0.5), float(rng.choice([0.0, 0.2, 1.0])), float(rng.choice([1.0, 0.2]))
        res = torch.randn(*ref.shape, generator=g) if use_res else None
        y = ref + res if (use_res and pre) else ref
        y = F.leaky_relu(y, slope) * alpha
        y = y + res if (use_res and not pre) else y
        pc = E.pack_conv(wt, b, None, stride, pad, device)
        split_in = f16 and bool(rng.random() < 0.7)
        split_out = f16 and cout % 32 == 0 and bool(rng.random() < 0.5)
        xa = E.Act(_nhwc(x, device))
        xa = E.f32_to_split32(xa) if split_in else xa
        ra = None
        if use_res:
            ra = E.Act(_nhwc(res, device))
            ra = E.f32_to_split32(ra) if (f16 and cout % 32 == 0 and rng.random() < 0.5) else ra
        tiles = [(128, 32), (128, 64), (128, 128)]
        if split_in and cout % 8 == 0:
            tiles += [(256, 128), (256, 256)]
            if (k, stride, pad) == (3, 1, 1) and cout <= 32:
                tiles.append((1, 32))
        if (split_in or split_out or (ra is not None and ra.fmt)) and cout % 8:
            continue                                              # split32 epilogue needs cout % 8 == 0
        for tm, tn in tiles:
            out = E.conv(pc, xa, act_slope=slope, alpha=alpha, res1=ra, res1_pre=pre, out_fmt=int(split_out), tile_m=tm, tile_n=tn)
            err = (out.nchw().cpu() - y).abs().max().item()
            assert err <= _tol(y) * (3 if not f16 else 1) + 1e-6, (it, k, stride, pad, cin, cout, n, h, w, tm, tn, err)


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, stride, pad, tile_n, extras
    (1, 20, 24, 64, 64, 3, 1, 1, 64, "lrelu"),        # HRconv's shape (rrdb.py:80)
    (1, 20, 24, 64, 3, 3, 1, 1, 32, "slice3of4"),     # conv_last's shape: 3 channels into an NHWC4 buffer
    (2, 13, 17, 128, 160, 1, 2, 0, 128, "res"),
    (1, 37, 45, 3, 64, 7, 2, 3, 64, "cin4"),
])
def test_conv_flat_addressing(case, device):
    """The 64-bit flat-addressing variant of the fp32 kernel (`launch<BN, *, false>`): on its own it only runs
    for tensors >= 4 GiB (RRDB's x4 tail at 1024^2 inputs); FCP_CONV_FLAT_ADDR forces it at test sizes.  Must
    equal the buffer-addressed variant bit for bit and the torch fp32 reference to roundoff."""
    from face_crop_plus_amd import engine as E
    n, h, w, cin, cout, k, stride, pad, tile_n, extra = case
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, stride, pad)
    pc = E.pack_conv(wt, b, None, stride, pad, device, precision="f32")
    if extra == "cin4":
        xin = E.Act(_nhwc(F.pad(x, (0, 0, 0, 0, 0, 1)), device))
    else:
        xin = E.Act(_nhwc(x, device))
    kw = {}
    if extra == "lrelu":
        kw["act_slope"], ref = 0.2, F.leaky_relu(ref, 0.2)
    if extra == "res":
        r = torch.randn(ref.shape, generator=g)
        kw.update(res1=E.Act(_nhwc(r, device)), res1_pre=True, act_slope=0.0)
        ref = F.relu(ref + r)
    outs = []
    for flat in (False, True):
        if extra == "slice3of4":
            buf = E.Act(torch.zeros(n, ref.shape[2], ref.shape[3], 4, device=device))
            outs.append(E.conv(pc, xin, buf.slice(0, 3), tile_n=tile_n, flat=flat, **kw).nchw().cpu())
            assert float(buf.buf[..., 3].abs().max()) == 0.0, "the pad channel of the NHWC4 buffer was written"
        else:
            outs.append(E.conv(pc, xin, tile_n=tile_n, flat=flat, **kw).nchw().cpu())
    assert torch.equal(outs[0], outs[1]), "flat and buffer addressing disagree"
    assert (outs[1] - ref).abs().max().item() <= _tol(ref)


def test_conv_flat_flag_rejected_on_f16x3(device):
    from face_crop_plus_amd import engine as E
    wt = torch.randn(64, 64, 1, 1)
    pc = E.pack_conv(wt, None, None, 1, 0, device, precision="f16x3")
    with pytest.raises(RuntimeError, match="FCP_CONV_FLAT_ADDR"):
        E.conv(pc, E.Act(torch.zeros(1, 4, 4, 64, device=device)), flat=True)


@pytest.mark.f16x3_only
def test_autotune_does_not_corrupt_in_place_ops(device, precision):
    """An op whose output aliases a residual (RRDB conv5 of the third dense block: out and res2 are the same
    tensor) must give the same result whether or not its shape is being autotuned (trial launches go to scratch)."""
    if precision != "f16x3":
        pytest.skip("autotuned tiles exist on the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(9)
    n, h, w = 1, 24, 40
    wt = torch.randn(64, 192, 3, 3, generator=g) / 42
    pc = E.pack_conv(wt, torch.randn(64, generator=g), None, 1, 1, device)
    src = E.f32_to_split32(E.Act(torch.randn(n, h, w, 192, generator=g).to(device)))
    base = torch.randn(n, h, w, 192, generator=g).to(device)
    outs = []
    for tuned in (False, True):
        tgt = E.f32_to_split32(E.Act(base.clone()))
        prev, E.Autotune.enabled = E.Autotune.enabled, tuned
        saved, E.Autotune.cache = E.Autotune.cache, {}
        try:
            E.conv(pc, src, tgt.slice(0, 64), alpha=0.2, res1=src.slice(0, 64), res1_pre=False, res2=tgt.slice(0, 64), alpha2=0.2)
        finally:
            E.Autotune.enabled, E.Autotune.cache = prev, saved
        outs.append(tgt.buf.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.f16x3_only
@pytest.mark.parametrize("n,h,w", [(2, 77, 91), (1, 64, 64), (3, 50, 130), (1, 9, 200), (2, 512, 512)])
def test_fused_stem_pool_f32_input(n, h, w, device, precision):
    """Round 5: the fused stem on a normalised fp32 NHWC4 input (BiSeNet's ResNet-18 stem): conv 7x7 / 2 (BatchNorm folded) ->
    ReLU -> max-pool 3x3 / 2 in one launch, the activation split hi + lo while it is staged.  Against torch fp32 of the three
    ops, against the unfused engine path (conv + max-pool: same accuracy class, different summation order), split32 and fp32
    outputs equal after decoding, odd sizes and border patches, and a channel slice of a wider buffer."""
    if precision != "f16x3":
        pytest.skip("the fused stem exists on the fp16x3 path")
    from face_crop_plus_amd import engine as E
    g = torch.Generator().manual_seed(n + h * 3 + w)
    x = torch.randn(n, 3, h, w, generator=g) * 1.3 + 0.2                 # (x / 255 - mean) / std has this range
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12
    bn = {"weight": torch.rand(64, generator=g) + 0.5, "bias": torch.randn(64, generator=g) * 0.1,
          "running_mean": torch.randn(64, generator=g) * 0.1, "running_var": torch.rand(64, generator=g) + 0.5}
    ref = F.max_pool2d(F.relu(F.batch_norm(F.conv2d(x, wt, None, 2, 3), bn["running_mean"], bn["running_var"], bn["weight"],
                                           bn["bias"], False, 0.0, 1e-5)), 3, 2, 1)
    x4 = torch.zeros(n, h, w, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    x4[..., 3] = 7.0                                                      # channel 3 must be ignored
    xa = E.Act(x4.to(device))
    ps = E.pack_stem_fused(wt, bn, device)
    out1 = E.stem_relu_pool_f32(ps, xa, out_fmt=1)
    out0 = E.stem_relu_pool_f32(ps, xa, out_fmt=0)
    assert out1.fmt == 1 and tuple(out1.nchw().shape) == tuple(ref.shape)
    # the split32 image is the fp32 result cut to hi + lo (22 significant bits)
    assert (out1.nchw() - out0.nchw()).abs().max().item() <= 2.0 ** -20 * float(ref.abs().max())
    err = (out0.nchw().cpu() - ref).abs().max().item()
    assert err <= _tol(ref), err
    x4z = x4.clone(); x4z[..., 3] = 0.0                                   # the generic cin4 kernel multiplies channel 3 by zero weights
    pc = E.pack_conv(wt, None, bn, 2, 3, device)
    unfused = E.maxpool3x3s2(E.conv(pc, E.Act(x4z.to(device)), act_slope=0.0, out_fmt=1))
    assert (unfused.nchw().cpu() - out1.nchw().cpu()).abs().max().item() <= _tol(ref)
    wide = E.Act.empty(n, ref.shape[2], ref.shape[3], 128, device, 1)
    wide.buf.zero_()
    E.stem_relu_pool_f32(ps, xa, wide.slice(64, 64))
    got = wide.nchw()
    assert torch.equal(got[:, 64:], out1.nchw()) and got[:, :64].abs().max().item() == 0
    again = E.stem_relu_pool_f32(ps, xa, out_fmt=1)
    assert torch.equal(again.buf, out1.buf)
