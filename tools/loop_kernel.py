"""Run ONE conv-engine launch in a loop for ~N seconds (power / clock sampling with tools/smi_sample.sh).
    python tools/loop_kernel.py chain|pair2|pair3|expand3|big1x1res|pair2s|c3ds|big3x3|big1x1|dma1x1|dma3x3s2|halo|wide|stem|copy [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
what = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
g = torch.Generator().manual_seed(0)
mk = lambda co, ci, k: E.pack_conv(torch.randn(co, ci, k, k, generator=g) * (2 / (ci * k * k)) ** 0.5, torch.randn(co, generator=g) * 0.1,
                                   None, 1, k // 2, dev, precision="f16x3")
sp = lambda n, h, c: E.f32_to_split32(E.Act(torch.randn(n, h, h, c, device=dev).relu()))
b = 64
if what == "chain":
    pc2, pc3, pc1 = mk(64, 64, 3), mk(256, 64, 1), mk(64, 256, 1)
    t1, x = sp(b, 160, 64), sp(b, 160, 256)
    out, t1n = E.Act.empty(b, 160, 160, 256, dev, 1), E.Act.empty(b, 160, 160, 64, dev, 1)
    f = lambda: E.bottleneck_chain(pc2, pc3, pc1, t1, x, out, t1n)
elif what in ("pair2", "pair3"):
    c, nout, cn, h = (128, 512, 128, 80) if what == "pair2" else (256, 1024, 256, 40)
    pc3, pc1 = mk(nout, c, 1), mk(cn, nout, 1)
    t, x = sp(b, h, c), sp(b, h, nout)
    out, t1n = E.Act.empty(b, h, h, nout, dev, 1), E.Act.empty(b, h, h, cn, dev, 1)
    f = lambda: E.bottleneck_chain(None, pc3, pc1, t, x, out, t1n)
elif what == "big3x3":
    pc = mk(256, 256, 3); x = sp(b, 80, 256); out = E.Act.empty(b, 80, 80, 256, dev, 1)
    f = lambda: E.conv(pc, x, out, act_slope=0.0, tile_m=256, tile_n=256)
elif what == "halo":
    pc = mk(32, 160, 3); x = sp(1, 1024, 192); 
    f = lambda: E.conv(pc, x.slice(0, 160), x.slice(160, 32), act_slope=0.2, tile_m=1, tile_n=32)
elif what == "dma1x1":       # 128-row LDS-DMA kernel: 1x1 512 -> 128 @80x80 (layer 2's conv1)
    pc = mk(128, 512, 1); x = sp(b, 80, 512); out = E.Act.empty(b, 80, 80, 128, dev, 1)
    f = lambda: E.conv(pc, x, out, act_slope=0.0, tile_m=128, tile_n=128)
elif what == "dma3x3s2":     # 128-row LDS-DMA kernel: 3x3 / 2 128 -> 128 (layer2.0.conv2)
    pc = E.pack_conv(torch.randn(128, 128, 3, 3, generator=g) * (2 / 1152) ** 0.5, torch.randn(128, generator=g) * 0.1, None, 2, 1, dev, precision="f16x3")
    x = sp(b, 160, 128); out = E.Act.empty(b, 80, 80, 128, dev, 1)
    f = lambda: E.conv(pc, x, out, act_slope=0.0, tile_m=128, tile_n=128)
elif what == "big1x1":       # 256-row kernel, short K: 1x1 1024 -> 256 @40x40
    pc = mk(256, 1024, 1); x = sp(b, 40, 1024); out = E.Act.empty(b, 40, 40, 256, dev, 1)
    f = lambda: E.conv(pc, x, out, act_slope=0.0, tile_m=256, tile_n=256)
elif what == "wide":         # wide halo-tile kernel: 3x3 128 -> 128 @80x80 (layer 2's conv2)
    pc = mk(128, 128, 3); x = sp(b, 80, 128); out = E.Act.empty(b, 80, 80, 128, dev, 1)
    f = lambda: E.conv(pc, x, out, act_slope=0.0, tile_m=1, tile_n=128)
elif what == "stem":         # uint8 -> stem conv + pool + layer1.0.conv1
    from face_crop_plus_amd import weights
    sd = weights.generate_state_dict("retinaface")
    ps = E.pack_stem_fused(sd["body.conv1.weight"], E.bn_of(sd, "body.bn1"), dev, cin_perm=[2, 1, 0])
    c1 = E.pack_conv(sd["body.layer1.0.conv1.weight"], None, E.bn_of(sd, "body.layer1.0.bn1"), 1, 0, dev)
    imgs = torch.randint(0, 256, (b, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)
    o = E.Act.empty(b, 160, 160, 64, dev, 1); t1 = E.Act.empty(b, 160, 160, 64, dev, 1)
    f = lambda: E.stem_relu_pool_u8(ps, imgs, o, conv1=c1, t1=t1)
elif what in ("expand3", "big1x1res"):   # conv3 + identity of a layer-3 block alone: the expand form / the 256-row kernel
    pc3 = mk(1024, 256, 1); t, x = sp(b, 40, 256), sp(b, 40, 1024); out = E.Act.empty(b, 40, 40, 1024, dev, 1)
    f = (lambda: E.bottleneck_chain(None, pc3, None, t, x, out)) if what == "expand3" else \
        (lambda: E.conv(pc3, t, out, act_slope=0.0, res1=x, res1_pre=True, tile_m=256, tile_n=256, balance_tail=True))
elif what in ("pair2s", "c3ds"):         # layer2.0: two-source conv3 + downsample (+ layer2.1.conv1 in the pair form)
    pc3, pc1 = mk(512, 384, 1), mk(128, 512, 1)
    t, xb = sp(b, 80, 128), sp(b, 160, 256)
    out, t1n = E.Act.empty(b, 80, 80, 512, dev, 1), E.Act.empty(b, 80, 80, 128, dev, 1)
    f = (lambda: E.bottleneck_chain(None, pc3, pc1, t, None, out, t1n, t1b=xb, t1b_stride=2)) if what == "pair2s" else \
        (lambda: E.conv(pc3, t, out, act_slope=0.0, x2=xb, x2_stride=2, tile_m=256, tile_n=256))
elif what == "copy":
    a = torch.empty(1 << 28, device=dev); c = torch.empty_like(a)
    f = lambda: c.copy_(a)
else:
    raise SystemExit(what)
for _ in range(3): f()
torch.cuda.synchronize()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50): f()
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
print(f"{what}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch over {n} launches", flush=True)
