#!/bin/bash
# GPU box: cycle attribution of the bottleneck-chain kernel's chunk loop (experiment build, device printf)
cd $GRAFT_REPO_ROOT
FCP_BUILD_DEFINES="FCP_CHAIN_PROBE=1" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from face_crop_plus_amd import engine as E
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
mk = lambda co, ci, k: E.pack_conv(torch.randn(co, ci, k, k, generator=g) * (2 / (ci * k * k)) ** 0.5, torch.randn(co, generator=g) * 0.1,
                                   None, 1, k // 2, dev, precision="f16x3")
b = 64
for (hh, c, nout, cn, has_c2, residual) in ((160, 64, 256, 64, True, True), (160, 64, 256, 128, True, True), (80, 128, 512, 128, False, True), (160, 128, 256, 64, False, False)):
    pc2 = mk(c, c, 3) if has_c2 else None
    pc3, pc1 = mk(nout, c, 1), mk(cn, nout, 1)
    t = E.f32_to_split32(E.Act(torch.randn(b, hh, hh, c, device=dev).relu()))
    xr = E.f32_to_split32(E.Act(torch.randn(b, hh, hh, nout, device=dev).relu())) if residual else None
    out, t1n = E.Act.empty(b, hh, hh, nout, dev, 1), E.Act.empty(b, hh, hh, cn, dev, 1)
    print(f"== c={c} nout={nout} cn={cn} conv2={has_c2} residual={residual}", flush=True)
    E.bottleneck_chain(pc2, pc3, pc1, t, xr, out, t1n)
    torch.cuda.synchronize()
PY
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
