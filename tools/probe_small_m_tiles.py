"""Small-M / large-K convs of BiSeNet's layer 4 and ARM (batch 32 faces @512^2 -> 16x16 maps, M = 8192): which tile?
   python tools/probe_small_m_tiles.py"""
import os, sys
os.environ["FCP_BOUNDARY"] = "ctypes"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
E.Autotune.enabled = False
mk = lambda co, ci, k, s=1: E.pack_conv(torch.randn(co, ci, k, k, generator=g) * (2 / (ci * k * k)) ** 0.5, torch.randn(co, generator=g) * 0.1, None, s, k // 2, dev, precision="f16x3")
sp = lambda n, h, c: E.f32_to_split32(E.Act(torch.randn(n, h, h, c, device=dev).relu()))
cases = [("3x3 512->512 @16", mk(512, 512, 3), sp(32, 16, 512), [(128, 128), (128, 64), (128, 32), (256, 128), (256, 256)]),
         ("3x3 512->128 @16", mk(128, 512, 3), sp(32, 16, 512), [(128, 128), (128, 64), (128, 32), (1, 128), (256, 128)]),
         ("3x3 256->256 @32", mk(256, 256, 3), sp(32, 32, 256), [(128, 128), (128, 64), (128, 32), (256, 128), (256, 256)]),
         ("3x3/2 256->512 @16", mk(512, 256, 3, 2), sp(32, 32, 256), [(128, 128), (128, 64), (128, 32), (256, 128), (256, 256)])]
for name, pc, x, tiles in cases:
    ref = None
    for t in tiles:
        try:
            f = lambda: E.conv(pc, x, act_slope=0.0, out_fmt=1, tile_m=t[0], tile_n=t[1])
            y = f(); f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                f()
            e1.record(); torch.cuda.synchronize()
            same = "" if ref is None else (" same bits" if torch.equal(y.buf, ref) else " DIFFERENT BITS")
            if ref is None:
                ref = y.buf.clone()
            print(f"{name} tile {t}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us{same}", flush=True)
        except Exception as ex:
            print(f"{name} tile {t}: {type(ex).__name__}: {str(ex)[:90]}")
