"""Run the benchmark geometries once with tuning on and write the tuner's picks as the table shipped in
face-crop-plus_amd/tuned/ (run on the MI355X: `python tools/dump_autotune.py gpurun_out/tuned.json`, then copy the file
into face-crop-plus_amd/tuned/).  Picks only select among tiles that return identical bits."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FCP_TUNE_CACHE"] = "0"          # measure everything afresh, write only the file named on the command line

import torch

from face_crop_plus_amd import engine as E, weights
from face_crop_plus_amd.bise import BiSeNet
from face_crop_plus_amd.retinaface import RetinaFace
from face_crop_plus_amd.rrdb import RRDBNet

dev = torch.device("cuda:0")
det = RetinaFace("largest", 0.6).load(dev, weights.generate_state_dict("retinaface"))
g = torch.Generator().manual_seed(1)
E.Autotune.enabled = True
for n, s in ((64, 640), (32, 1024), (8, 1024), (4, 1024), (2, 1024), (8, 640)):
    imgs = torch.randint(0, 256, (n, s, s, 3), generator=g, dtype=torch.uint8).to(dev)
    for streams in (2, 1):
        det.streams = streams
        det.detect(imgs, max_faces=n)
        torch.cuda.synchronize()
    print("detect", n, s, len(E.Autotune.cache), flush=True)
par = BiSeNet({"glasses": [6]}, {"eyes": [4, 5]}, 32).load(dev, weights.generate_state_dict("bisenet"))
for f in (32, 64, 8):
    par.parse(torch.randint(0, 256, (f, 256, 256, 3), generator=g, dtype=torch.uint8).to(dev))
    torch.cuda.synchronize()
print("parse", len(E.Autotune.cache), flush=True)
enh = RRDBNet(0.001).load(dev, weights.generate_state_dict("rrdb"))
imgs = torch.randint(0, 256, (1, 1024, 1024, 3), generator=g, dtype=torch.uint8).to(dev)
enh.enhance_u8(imgs, [0])
torch.cuda.synchronize()
print("enhance", len(E.Autotune.cache), flush=True)
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tuned.json"
print(E.Autotune.save(out), E.Autotune.section())
