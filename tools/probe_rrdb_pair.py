"""RRDB pair fusion, decided by measurement of its main loop with the kernels that exist.

A fused (conv_k, conv_k+1) launch of a dense block stages every earlier 32-channel slice ONCE and
runs two 32-filter passes over it — conv_k (on the 10 x 34 ring-extended patch: 11 row tiles instead of 8) and conv_k+1's partial
sums (8 row tiles) — then conv_k+1's last slice from conv_k's output kept on chip.  Its main loop is therefore exactly a
64-FILTER convolution over the shared slices, on 19 / 16 of the row tiles, which the halo-tile kernels already run in two forms:
two passes over one staged patch (tile 1x32 with cout 64) and the wide form (column tiles inner, filters through a tap ring: tile
1x64).  This probe times, each alone in a steady loop with clock / power sampled (bench.Telemetry -> W, MHz, joules per launch):

    S3  = conv 128 -> 32  (conv3 of a dense block: 4 slices)              S4 = conv 160 -> 32 (conv4: 5 slices)
    P2  = conv 128 -> 64, two passes over the shared patch                 PW = the same in the wide form
    T1  = conv 64 -> 32 scaled to one slice (the fused launch's tail: conv4's last slice)

    pair lower bound = min(P2, PW) x 19 / 16 + T1 / 2          against          S3 + S4   (what runs today)

(the bound leaves out conv_k's epilogue into LDS and the drain between the two phases).  Same for the (conv1, conv2) pair with 64 / 96
input channels.   python tools/probe_rrdb_pair.py [size=1024] [seconds=1.0]"""
import os
import sys
import time

os.environ["FCP_BOUNDARY"] = "ctypes"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from face_crop_plus_amd import engine as E

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
E.Autotune.enabled = False
mk = lambda co, ci: E.pack_conv(torch.randn(co, ci, 3, 3, generator=g) * (1 / (ci * 9)) ** 0.5, torch.randn(co, generator=g) * 0.05,
                                None, 1, 1, dev, precision="f16x3")
buf = E.f32_to_split32(E.Act(torch.randn(1, size, size, 256, device=dev) * 0.5))      # [earlier slices | outputs]
tele = bench.Telemetry(0, period_s=0.005)


def run(label, pc, cin, cout, tile):
    x, out = buf.slice(0, cin), buf.slice(192, cout)
    f = lambda: E.conv(pc, x, out, act_slope=0.2, tile_m=tile[0], tile_n=tile[1])
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
    reps = max(20, int(secs * 1e3 / (e0.elapsed_time(e1) / 10)))
    with tele:
        ta = time.perf_counter()
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        tb = time.perf_counter()
    s = tele.summary(ta + min(0.1, 0.25 * (tb - ta)), tb)
    us = e0.elapsed_time(e1) / reps * 1e3
    j = s["mean_power_w"] * us * 1e-6 if s["mean_power_w"] else float("nan")
    gflop = 2 * 9 * cin * cout * size * size / 1e9
    print(f"{label:34s} {us:8.1f} us  {s['mean_sclk_mhz']} MHz  {s['mean_power_w']} W  {j:.3f} J/launch  {gflop / us * 1e3:.0f} TFLOP/s algorithmic", flush=True)
    return us, j


res = {}
for name, cin in (("pair (conv3, conv4)", 128), ("pair (conv1, conv2)", 64)):
    print(f"== {name}: shared slices = {cin // 32}, image {size}x{size}")
    sA = run(f"S_k   {cin}->32 tile 1x32", mk(32, cin), cin, 32, (1, 32))
    sB = run(f"S_k+1 {cin + 32}->32 tile 1x32", mk(32, cin + 32), cin + 32, 32, (1, 32))
    p2 = run(f"P2    {cin}->64 tile 1x32 (2 passes)", mk(64, cin), cin, 64, (1, 32))
    pw = run(f"PW    {cin}->64 tile 1x64 (wide)", mk(64, cin), cin, 64, (1, 64))
    t1 = run("T     64->32 tile 1x32 (2 slices)", mk(32, 64), 64, 32, (1, 32))
    today, today_j = sA[0] + sB[0], sA[1] + sB[1]
    best = min(p2, pw)
    bound, bound_j = best[0] * 19 / 16 + t1[0] / 2, best[1] * 19 / 16 + t1[1] / 2
    print(f"   today S_k + S_k+1 = {today:.1f} us, {today_j:.3f} J   |   fused pair >= {bound:.1f} us, {bound_j:.3f} J "
          f"(main loop {best[0]:.1f} x 19/16 + tail {t1[0] / 2:.1f})   ->  {'NEGATIVE' if bound >= today else 'positive'}: "
          f"{(bound / today - 1) * 100:+.1f} % time, {(bound_j / today_j - 1) * 100:+.1f} % joules")
    print(f"   sharing the staged patch between two 32-filter passes: P2 / (2 x S_k) = {p2[0] / (2 * sA[0]):.3f}, PW / (2 x S_k) = {pw[0] / (2 * sA[0]):.3f}")
print("telemetry:", tele.source)
