"""sha256 of the fused stem kernel's outputs on seeded inputs (odd sizes, borders, both formats, with and without conv1):
run before and after a change that must keep every bit.  python tools/stem_hash.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
wt = torch.randn(64, 3, 7, 7, generator=g) / 12
bn = {"weight": torch.rand(64, generator=g) + 0.5, "bias": torch.randn(64, generator=g) * 0.1,
      "running_mean": torch.randn(64, generator=g) * 0.1, "running_var": torch.rand(64, generator=g) + 0.5}
ps = E.pack_stem_fused(wt, bn, dev)
c1 = E.pack_conv(torch.randn(64, 64, 1, 1, generator=g) / 8, None, bn, 1, 0, dev, precision="f16x3")
hs = hashlib.sha256()
for (n, h, w) in ((1, 1, 4), (1, 7, 9), (2, 203, 317), (3, 64, 64), (1, 33, 1000), (2, 641, 639), (4, 640, 640)):
    img = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, generator=g).to(dev)
    for fmt in (0, 1):
        out = E.stem_relu_pool_u8(ps, img, out_fmt=fmt)
        torch.cuda.synchronize()
        hs.update(out.buf.cpu().numpy().tobytes())
    out, t1 = E.stem_relu_pool_u8(ps, img, conv1=c1)
    torch.cuda.synchronize()
    hs.update(out.buf.cpu().numpy().tobytes()); hs.update(t1.buf.cpu().numpy().tobytes())
    print(n, h, w, hs.hexdigest()[:16], flush=True)
print("stem hash", hs.hexdigest())
