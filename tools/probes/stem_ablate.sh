#!/bin/bash
# GPU box: ablation builds of the fused stem kernel's MFMA phase (wrong results, cycle probes only)
cd $GRAFT_REPO_ROOT
for d in "FCP_STEM_PROBE=1" "FCP_STEM_PROBE=1 FCP_STEM_ABLATE_AREAD=1" "FCP_STEM_PROBE=1 FCP_STEM_ABLATE_STAGE=1" "FCP_STEM_PROBE=1 FCP_STEM_ABLATE_AREAD=1 FCP_STEM_ABLATE_STAGE=1"; do
  echo "== $d"
  FCP_BUILD_DEFINES="$d" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
  python - <<'PY' 2>&1 | grep "wave [05]:" | head -2
import sys, torch
sys.path.insert(0, ".")
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
img = torch.randint(0, 256, (64, 640, 640, 3), dtype=torch.uint8, device=dev)
wt = torch.randn(64, 3, 7, 7) / 12
bn = {"weight": torch.ones(64), "bias": torch.zeros(64), "running_mean": torch.zeros(64), "running_var": torch.ones(64)}
ps = E.pack_stem_fused(wt, bn, dev)
cat = E.Act.empty(64, 160, 160, 128, dev, 1)
E.stem_relu_pool_u8(ps, img, cat.slice(64, 64)); torch.cuda.synchronize()
PY
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
