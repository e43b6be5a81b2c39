import sys; sys.path.insert(0, '.')
import torch
from face_crop_plus_amd import engine as E
dev = torch.device('cuda:0')
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = E.Act(torch.randn(64, 160, 160, 256, device=dev))     # 1.68 GB
gb = x.buf.numel() * 4 / 1e9
ms = timeit(lambda: E.f32_to_split32(x)); print(f"f32_to_split32 1.68GB in+out: {ms:.3f} ms  {2*gb/ms:.2f} TB/s")
y = torch.empty_like(x.buf)
ms = timeit(lambda: y.copy_(x.buf)); print(f"torch copy: {ms:.3f} ms {2*gb/ms:.2f} TB/s")
ms = timeit(lambda: torch.add(x.buf, y, out=y)); print(f"torch add (2 reads 1 write): {ms:.3f} ms {3*gb/ms:.2f} TB/s")
ms = timeit(lambda: y.zero_()); print(f"torch fill (write only): {ms:.3f} ms {gb/ms:.2f} TB/s")
# conv c3-type with residual
with E.default_precision("f16x3"):
    xin = E.f32_to_split32(E.Act(torch.randn(64, 160, 160, 64, device=dev)))
    res = E.f32_to_split32(x)
    pc = E.pack_conv(torch.randn(256, 64, 1, 1) / 8, torch.zeros(256), None, 1, 0, dev)
    out = E.conv(pc, xin, act_slope=0.0, res1=res, res1_pre=True, out_fmt=1)
    for tn in (64, 128):
        ms = timeit(lambda: E.conv(pc, xin, out, act_slope=0.0, res1=res, res1_pre=True, tile_n=tn))
        print(f"c3 64->256 +res tile{tn}: {ms:.3f} ms {(0.42+1.68+1.68)/ms:.2f} TB/s")
        ms = timeit(lambda: E.conv(pc, xin, out, act_slope=0.0, tile_n=tn))
        print(f"c3 64->256 no res tile{tn}: {ms:.3f} ms {(0.42+1.68)/ms:.2f} TB/s")
    pc1 = E.pack_conv(torch.randn(64, 256, 1, 1) / 16, torch.zeros(64), None, 1, 0, dev)
    o1 = E.conv(pc1, res, act_slope=0.0, out_fmt=1)
    ms = timeit(lambda: E.conv(pc1, res, o1, act_slope=0.0, tile_n=64)); print(f"c1 256->64: {ms:.3f} ms {(1.68+0.42)/ms:.2f} TB/s")
