"""Why does a workload measured SECOND in a bench.py process run 4-10 % slower than the same workload measured first
(r06: configs[1] 3454 faces/s as the headline, 3107 as an `extra` behind the 1024^2 headline)?  One process, sequences of
Pipelines, ms per step of each;  python tools/probe_second_pipeline.py <mode>
  modes: AB (1024 then 640), BB (640 twice), BB_keep (no empty_cache between), BB_samestreams (second detector reuses the first one's streams),
         BB_nogc"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from face_crop_plus_amd import weights

mode = sys.argv[1] if len(sys.argv) > 1 else "BB"
dev = torch.device("cuda:0")
sd = weights.generate_state_dict("retinaface")
bench.Telemetry.disabled = True
geo = {"A": (32, 1024), "B": (64, 640)}
skip = int(mode[4:]) if mode.startswith("skip") else 0
dummies = [torch.cuda.Stream(device=dev) for _ in range(skip)]          # shift the detector's streams along torch's stream pool
if skip:
    mode_key = "B"
seq = {"AB": "AB", "BA": "BA", "BB": "BB", "BB_keep": "BB", "BB_samestreams": "BB", "BBB": "BBB", "AA": "AA", "BBBBBB": "BBBBBB"}.get(mode, "B")
prev_streams = None
for i, g in enumerate(seq):
    b, s = geo[g]
    p = bench.Pipeline(dev, sd, full=False, batch=b, size=s, out_size=256, strategy="largest", precision="f16x3", enhance="none",
                       streams=2, seed=1234 + i)
    if mode == "BB_samestreams" and prev_streams is not None:
        p.det._tls.__dict__["streams"] = prev_streams
    el, faces = bench.time_pipeline(p, 20, 5)
    print(f"{mode} #{i} {g}: {el / 20 * 1e3:.3f} ms/step, {int(faces.item()) / el:.1f} faces/s, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
    prev_streams = p.det._tls.__dict__.get("streams")
    del p
    gc.collect()
    if mode != "BB_keep":
        torch.cuda.empty_cache()
