import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from face_crop_plus_amd import weights
dev = torch.device("cuda:0"); sd = weights.generate_state_dict("retinaface"); bench.Telemetry.disabled = True
for batch in (8, 12, 16):
    for mi in (8, 4, 2):
        p = bench.Pipeline(dev, sd, full=False, batch=batch, size=1024, out_size=256, strategy="largest", precision="f16x3", enhance="none", streams=2, seed=1)
        p.det.min_images_per_stream = mi
        el, faces = bench.time_pipeline(p, 30, 5)
        print(f"batch {batch} min_images_per_stream {mi}: {el / 30 * 1e3:.3f} ms/step, {int(faces.item()) / el:.1f} faces/s", flush=True)
        del p
