#!/bin/bash
# GPU box: cycle attribution inside the 256-row kernel (experiment build with s_memtime probes + device printf)
cd $GRAFT_REPO_ROOT
FCP_BUILD_DEFINES="FCP_BIG_PROBE=1" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
for (n, h, w, cin, cout, k) in ((64, 40, 40, 256, 256, 3), (64, 40, 40, 1024, 256, 1), (64, 80, 80, 256, 256, 3), (64, 40, 40, 768, 1024, 1), (64, 20, 20, 1536, 2048, 1)):
    x = E.f32_to_split32(E.Act(torch.randn(n, h, w, cin, device=dev)))
    pc = E.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), None, 1, k // 2, dev, precision="f16x3")
    out = E.Act.empty(n, h, w, cout, dev, 1)
    print(f"== {cin}->{cout} {k}x{k} @{h}x{w}", flush=True)
    E.conv(pc, x, out, act_slope=0.0, tile_n=256, tile_m=256)
    torch.cuda.synchronize()
PY
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
