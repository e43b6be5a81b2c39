"""Fused bottleneck chain vs the three stand-alone convs at the bench shape (batch 64, 160x160): us per launch.
FCP_CHAIN_TILE_M=256 times the 8-wave / 256-pixel tiles (where supported) instead of the 4-wave ones.
FCP_CHAIN_ABLATE (profiling builds: FCP_BUILD_PROFILING=1 python face-crop-plus_amd/build_native.py --force)
attributes the chain's time: 1 no out stores, 2 no residual loads, 4 no phase-1 loop, 8 no chunk loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
b, h = int(os.environ.get("B", 64)), int(os.environ.get("H", 160))
g = torch.Generator().manual_seed(0)
mk = lambda co, ci, k: E.pack_conv(torch.randn(co, ci, k, k, generator=g) * (2 / (ci * k * k)) ** 0.5, torch.randn(co, generator=g) * 0.1,
                                   None, 1, k // 2, dev, precision="f16x3")
pc2, pc3 = mk(64, 64, 3), mk(256, 64, 1)
t1 = E.f32_to_split32(E.Act(torch.randn(b, h, h, 64, device=dev).relu()))
x = E.f32_to_split32(E.Act(torch.randn(b, h, h, 256, device=dev).relu()))


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for cn in (64, 128):
    pc1 = mk(cn, 256, 1)
    out, t1n = E.Act.empty(b, h, h, 256, dev, 1), E.Act.empty(b, h, h, cn, dev, 1)
    o2 = E.Act.empty(b, h, h, 64, dev, 1)
    tc = timeit(lambda: E.bottleneck_chain(pc2, pc3, pc1, t1, x, out, t1n))
    tl = timeit(lambda: E.bottleneck_chain(pc2, pc3, pc1, t1, x, out, t1n, tile_m=128))
    tp = timeit(lambda: E.bottleneck_chain(pc2, pc3, pc1, t1, x, out, t1n, tile_m=16))
    tp2 = timeit(lambda: E.bottleneck_chain(pc2, pc3, pc1, t1, x, out, t1n, tile_m=32))
    tl2 = timeit(lambda: E.bottleneck_chain(pc2, pc3, pc1, t1, x, out, t1n, tile_m=256))
    print(f"cn={cn}: linear 128-pixel tiles {tl:7.1f} us, 8 x 16 patches {tp:7.1f} us, linear 256 {tl2:7.1f} us, 16 x 16 patches {tp2:7.1f} us", flush=True)
    t2 = timeit(lambda: E.conv(pc2, t1, o2, act_slope=0.0))
    t3 = timeit(lambda: E.conv(pc3, o2, out, act_slope=0.0, res1=x))
    t1_ = timeit(lambda: E.conv(pc1, out, t1n, act_slope=0.0))
    m = b * h * h
    gb = m * (64 + 256 + 256 + cn) * 4 / 1e9
    fl = (pc2.flops_per_pixel + pc3.flops_per_pixel + pc1.flops_per_pixel) * m
    print(f"cn={cn}: chain {tc:7.1f} us ({gb / tc * 1e6:5.0f} GB/s algorithmic, {fl / tc / 1e6:5.0f} TFLOP/s) | separate "
          f"{t2:6.1f} + {t3:6.1f} + {t1_:6.1f} = {t2 + t3 + t1_:7.1f} us  ablate={os.environ.get('FCP_CHAIN_ABLATE', '0')}", flush=True)

# ---- pair forms (no conv2)
for (hh, nout, cn, residual) in ((80, 512, 128, True), (80, 512, 256, True), (160, 256, 64, False)):
    pc3 = mk(nout, 128, 1)
    pc1 = mk(cn, nout, 1)
    t = E.f32_to_split32(E.Act(torch.randn(b, hh, hh, 128, device=dev).relu()))
    xr = E.f32_to_split32(E.Act(torch.randn(b, hh, hh, nout, device=dev).relu())) if residual else None
    out, t1n = E.Act.empty(b, hh, hh, nout, dev, 1), E.Act.empty(b, hh, hh, cn, dev, 1)
    tc = timeit(lambda: E.bottleneck_chain(None, pc3, pc1, t, xr, out, t1n))
    t3 = timeit(lambda: E.conv(pc3, t, out, act_slope=0.0, res1=xr))
    t1_ = timeit(lambda: E.conv(pc1, out, t1n, act_slope=0.0))
    m = b * hh * hh
    gb = m * (128 + (nout if residual else 0) + nout + cn) * 4 / 1e9
    fl = (pc3.flops_per_pixel + pc1.flops_per_pixel) * m
    print(f"pair {hh}x{hh} 128->{nout}->{cn} res={residual}: {tc:7.1f} us ({gb / tc * 1e6:5.0f} GB/s algorithmic, {fl / tc / 1e6:5.0f} TFLOP/s) | "
          f"separate {t3:6.1f} + {t1_:6.1f} = {t3 + t1_:7.1f} us  ablate={os.environ.get('FCP_CHAIN_ABLATE', '0')}", flush=True)

# ---- layer-3 pair (256 -> 1024 -> 256 at 40x40)
if True:
    hh, nout, cn = 40, 1024, 256
    pc3 = mk(nout, 256, 1)
    pc1 = mk(cn, nout, 1)
    t = E.f32_to_split32(E.Act(torch.randn(b, hh, hh, 256, device=dev).relu()))
    xr = E.f32_to_split32(E.Act(torch.randn(b, hh, hh, nout, device=dev).relu()))
    out, t1n = E.Act.empty(b, hh, hh, nout, dev, 1), E.Act.empty(b, hh, hh, cn, dev, 1)
    t3 = timeit(lambda: E.conv(pc3, t, out, act_slope=0.0, res1=xr))
    t1_ = timeit(lambda: E.conv(pc1, out, t1n, act_slope=0.0))
    tc = timeit(lambda: E.bottleneck_chain(None, pc3, pc1, t, xr, out, t1n)) if E.chain_supported(None, pc3, pc1) else float("nan")
    print(f"pair {hh}x{hh} 256->{nout}->{cn} res=True: {tc:7.1f} us | separate {t3:6.1f} + {t1_:6.1f} = {t3 + t1_:7.1f} us", flush=True)
