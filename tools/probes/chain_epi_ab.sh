#!/bin/bash
# GPU box: bottleneck-chain kernels with the staged (default) and the register epilogue (FCP_CHAIN_REG_EPI build), one call
cd $GRAFT_REPO_ROOT
for d in "" "FCP_CHAIN_REG_EPI=1"; do
  echo "== build defines: '$d'"
  FCP_BUILD_DEFINES="$d" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
  python tools/bench_chain.py 2>&1 | grep "chain\|pair\|layer" | cut -c1-80
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
