#!/bin/bash
# GPU box: cycle attribution inside the halo-tile kernel (experiment build with s_memtime probes + device printf)
cd $GRAFT_REPO_ROOT
FCP_BUILD_DEFINES="FCP_HALO_PROBE=1 ${EXTRA_DEFINES}" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
H = W = 1024
buf = E.f32_to_split32(E.Act(torch.randn(1, H, W, 192, device=dev)))
for cin, cout in ((160, 32), (192, 64)):
    pc = E.pack_conv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.zeros(cout), None, 1, 1, dev, precision="f16x3")
    out = E.Act.empty(1, H, W, cout, dev, 1)
    print(f"== {cin}->{cout}", flush=True)
    for _ in range(2):
        E.conv(pc, buf.slice(0, cin), out, act_slope=0.2, tile_n=32, tile_m=1)
        torch.cuda.synchronize()
PY
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
