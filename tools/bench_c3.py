"""c3-type (1x1, K = N/4, + residual) layers in isolation, split32, with profiling ablations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with E.default_precision("f16x3"):
    for name, hw, planes in (("l1", 160, 64), ("l2", 80, 128), ("l3", 40, 256)):
        xin = E.f32_to_split32(E.Act(torch.randn(64, hw, hw, planes, device=dev)))
        res = E.f32_to_split32(E.Act(torch.randn(64, hw, hw, planes * 4, device=dev)))
        pc = E.pack_conv(torch.randn(planes * 4, planes, 1, 1) / planes ** 0.5, torch.zeros(planes * 4), None, 1, 0, dev)
        out = E.conv(pc, xin, act_slope=0.0, res1=res, res1_pre=True, out_fmt=1)
        gb = (xin.buf.numel() + 2 * res.buf.numel()) * 4 / 1e9
        for tn in (64, 128):
            ms = timeit(lambda: E.conv(pc, xin, out, act_slope=0.0, res1=res, res1_pre=True, tile_n=tn))
            print(f"ablate={os.environ.get('FCP_CONV_ABLATE','0'):>3s} {name}.c3 +res tile{tn}: {ms:.3f} ms {gb/ms:.2f} TB/s", flush=True)
