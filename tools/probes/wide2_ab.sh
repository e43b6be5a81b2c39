#!/bin/bash
# GPU box: taps per barrier of the 64-filter wide halo-tile form (experiment builds), against the pass-per-32-filters kernel
cd $GRAFT_REPO_ROOT
for bt in 1 2 3; do
  echo "== FCP_WIDE2_BT=$bt"
  FCP_BUILD_DEFINES="FCP_WIDE2_BT=$bt" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
  python tools/bench_halo_wide.py 2>&1 | grep -A3 "l1 64->64\|rrdb conv5\|ssh 64->64" | grep "1x32\|1x64\|same"
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
