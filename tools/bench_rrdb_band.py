"""RRDB enhancer at 1024^2 (BASELINE configs[2] geometry): whole-image dense-block launches against the band-major
schedule (rrdb.py::_dense_block_banded), in ONE process so that the comparison shares a box and a clock state.

    python tools/bench_rrdb_band.py [bands ...]        # default: 0 64 128 192 256   (0 = whole image)

Prints ms per enhanced image (full forward incl. the x4 tail + bicubic) and ms of the trunk alone; with FCP_SMI=1 the
socket power is sampled through rocm-smi during each timed loop (joules per image = W x ms).
"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import engine as E, weights
from face_crop_plus_amd.rrdb import RRDBNet

dev = torch.device("cuda:0")
bands = [int(a) for a in sys.argv[1:]] or [0, 64, 128, 192, 256]
S = int(os.environ.get("FCP_BENCH_SIZE", "1024"))
REPS = int(os.environ.get("FCP_BENCH_REPS", "3"))
m = RRDBNet(0.001).load(dev, weights.generate_state_dict("rrdb"), "f16x3")
g = torch.Generator(device="cpu").manual_seed(7)
img = torch.randint(0, 256, (1, S, S, 3), generator=g, dtype=torch.uint8).to(dev)


class Smi(threading.Thread):
    """Average socket power while the timed loop runs (rocm-smi text output, ~4 samples / s)."""

    def __init__(self):
        super().__init__(daemon=True)
        self.watts, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                for line in out.splitlines():
                    if "Power" in line and "W" in line:
                        self.watts.append(float(line.split(":")[-1].strip().split()[0]))
                        break
            except Exception:
                pass
            time.sleep(0.05)


ref = None
for band in bands:
    RRDBNet.TRUNK_BAND = band
    E.Autotune.enabled = True
    work = img.clone()
    m.enhance_u8(work, [0])                      # tunes the band shapes
    torch.cuda.synchronize()
    E.Autotune.enabled = False
    if ref is None:
        ref = work.clone()
    same = bool(torch.equal(work, ref))
    smi = Smi() if os.environ.get("FCP_SMI") == "1" else None
    if smi:
        smi.start()
    work = img.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        work.copy_(img)
        m.enhance_u8(work, [0])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / REPS * 1e3
    if smi:
        smi.stop = True
        smi.join()
    w = sum(smi.watts) / len(smi.watts) if smi and smi.watts else float("nan")
    print(f"band {band:4d}: {ms:8.2f} ms per enhanced {S}x{S} image   same bytes as the first schedule: {same}"
          + (f"   {w:6.0f} W  {w * ms / 1e3:7.1f} J" if smi else ""), flush=True)
