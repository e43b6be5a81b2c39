"""Print the range guard's report for the generated weights (numbers quoted in profiles/r04_probes.md section 7)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from face_crop_plus_amd import weights
from face_crop_plus_amd.retinaface import RetinaFace
from face_crop_plus_amd.bise import BiSeNet
from face_crop_plus_amd.rrdb import RRDBNet
dev = torch.device("cuda:0")
for name, mk in (("retinaface", lambda sd: RetinaFace("largest", 0.6).load(dev, sd)), ("bisenet", lambda sd: BiSeNet(None, None, 8).load(dev, sd)),
                 ("rrdb", lambda sd: RRDBNet(0.001).load(dev, sd))):
    sd = weights.generate_state_dict(name)
    import time
    m = mk(sd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rep = m.selfcheck(sd)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    peak = torch.cuda.max_memory_allocated() / 2**20
    print(f"{name}: selfcheck took {dt:.2f} s (startup cost of load() for checkpoint files), peak device memory {peak:.0f} MiB, "
          f"after it {torch.cuda.memory_allocated() / 2**20:.0f} MiB")
    rows = rep["launch_absmax"]
    top = sorted(rows, key=lambda r: -r[1])[:3]
    print(name, len(rows), "launches; largest |x|:", [(l, round(v, 1)) for l, v in top], {k: v for k, v in rep.items() if k.endswith("rel_diff")})
