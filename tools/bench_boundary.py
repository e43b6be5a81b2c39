"""Host cost of the two boundaries: ctypes over the C ABI vs the registered torch.ops.fcp custom ops.
For each: host microseconds to ENQUEUE one detect + align + crop step (device idle before the call, timer around the call
only), and the resulting step time, at BASELINE configs[1] (batch 64 @640^2) and at a launch-bound batch 8 @320^2.
    python tools/bench_boundary.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from face_crop_plus_amd import torch_ops as T, weights, engine as E

dev = torch.device("cuda:0")
sd = weights.generate_state_dict("retinaface")
T.load()
for batch, size in ((64, 640), (8, 320)):
    for mode in ("ctypes", "torch"):
        T.ENABLED = mode == "torch"
        p = bench.Pipeline(dev, sd, full=False, batch=batch, size=size, out_size=256, strategy="largest", precision="f16x3",
                           enhance="none", streams=2, seed=1)
        E.Autotune.enabled = True
        p.step(); torch.cuda.synchronize()
        E.Autotune.enabled = False
        for _ in range(5):
            p.step()
        torch.cuda.synchronize()
        host = []
        for _ in range(30):
            torch.cuda.synchronize()
            t = time.perf_counter()
            p.step()
            host.append(time.perf_counter() - t)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 40
        for _ in range(n):
            p.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / n * 1e3
        host.sort()
        print(f"batch {batch:3d} @{size}: {mode:6s} host enqueue {host[len(host) // 2] * 1e6:8.0f} us/step (min {host[0] * 1e6:.0f}), "
              f"step {ms:7.3f} ms = {batch / ms * 1e3:7.0f} faces/s", flush=True)
        T.ENABLED = False
        del p
