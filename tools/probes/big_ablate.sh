#!/bin/bash
# GPU box: attribute the 256-row kernel's time (experiment builds, wrong results; the shipped .so is rebuilt at the end)
cd $GRAFT_REPO_ROOT
for a in ${ABL:-0 1 2 3 4 12}; do
  FCP_BUILD_DEFINES="FCP_BIG_ABLATE=$a" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
  echo "== FCP_BIG_ABLATE=$a: $(python tools/bench_big.py 256 | awk '{printf "%s %s; ", $1, $2}')"
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
