#!/bin/bash
# GPU box: cache-policy sweep of the LDS-DMA operand loads (A = activations, B = filters; 0 default, 2 nt, 16 sc1, 18 both).
cd $GRAFT_REPO_ROOT
for ab in ${PAIRS:-"0 0" "0 2" "0 16" "2 0" "16 0" "2 2"}; do
  set -- $ab
  FCP_BUILD_DEFINES="FCP_AUX_A=$1 FCP_AUX_B=$2" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
  echo "== AUX_A=$1 AUX_B=$2"
  eval "${CMD:-python tools/bench_big.py 256 64 0 | tail -11}"
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
