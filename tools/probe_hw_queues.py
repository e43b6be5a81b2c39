"""Which torch streams share a hardware queue?  HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) queues; two streams on
one queue run their kernels one after the other.  A one-workgroup spin kernel (torch.cuda._sleep) on stream i and on stream j:
~1 x its duration when the two overlap, ~2 x when they share a queue.   python tools/probe_hw_queues.py [n_streams=8]"""
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
cycles = 20_000_000
torch.cuda._sleep(cycles)
torch.cuda.synchronize()
t = time.perf_counter(); torch.cuda._sleep(cycles); torch.cuda.synchronize(); one = time.perf_counter() - t
print(f"one spin kernel: {one * 1e3:.1f} ms; default stream = D")
names = ["D"] + [str(i) for i in range(n)]
alls = [torch.cuda.default_stream(dev)] + streams
print("     " + " ".join(f"{x:>4s}" for x in names))
for i, si in enumerate(alls):
    row = []
    for j, sj in enumerate(alls):
        if j <= i:
            row.append("   .")
            continue
        torch.cuda.synchronize()
        t = time.perf_counter()
        with torch.cuda.stream(si):
            torch.cuda._sleep(cycles)
        with torch.cuda.stream(sj):
            torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        row.append(f"{(time.perf_counter() - t) / one:4.1f}")
    print(f"{names[i]:>4s} " + " ".join(row))
