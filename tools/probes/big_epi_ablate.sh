#!/bin/bash
# GPU box: prologue / main loop / epilogue cycles of the 256-row kernel's workgroups (experiment build, cycle probes).
# (The staged epilogue this was written for took 36-39 k cycles per 256 x 256 tile, 36 k with its stores ablated.)
cd $GRAFT_REPO_ROOT
for d in "FCP_BIG_PROBE=1"; do
  echo "== $d"
  FCP_BUILD_DEFINES="$d" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
  python - <<'PY' 2>&1 | grep "workgroup [1-9]"
import sys, torch
sys.path.insert(0, ".")
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
for (n, h, w, cin, cout, k) in ((64, 40, 40, 768, 1024, 1),):
    x = E.f32_to_split32(E.Act(torch.randn(n, h, w, cin, device=dev)))
    pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, 1, k // 2, dev, precision="f16x3")
    out = E.Act.empty(n, h, w, cout, dev, 1)
    E.conv(pc, x, out, act_slope=0.0, tile_n=256, tile_m=256)
    torch.cuda.synchronize()
PY
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
