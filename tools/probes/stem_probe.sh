#!/bin/bash
# GPU box: cycle attribution of the fused stem kernel's patch loop (experiment build, device printf)
cd $GRAFT_REPO_ROOT
FCP_BUILD_DEFINES="FCP_STEM_PROBE=1" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
img = torch.randint(0, 256, (64, 640, 640, 3), dtype=torch.uint8, device=dev)
wt = torch.randn(64, 3, 7, 7) / 12
bn = {"weight": torch.ones(64), "bias": torch.zeros(64), "running_mean": torch.zeros(64), "running_var": torch.ones(64)}
ps = E.pack_stem_fused(wt, bn, dev)
cat = E.Act.empty(64, 160, 160, 128, dev, 1)
print("== stem + pool", flush=True)
E.stem_relu_pool_u8(ps, img, cat.slice(64, 64)); torch.cuda.synchronize()
c1 = E.pack_conv(torch.randn(64, 64, 1, 1) / 8, None, bn, 1, 0, dev, precision="f16x3")
t1 = E.Act.empty(64, 160, 160, 64, dev, 1)
print("== stem + pool + conv1", flush=True)
E.stem_relu_pool_u8(ps, img, cat.slice(64, 64), conv1=c1, t1=t1); torch.cuda.synchronize()
PY
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
