#!/bin/bash
# GPU box: is the halo-tile kernel bound by LDS fragment reads?  Experiment builds that skip part of them (wrong
# results, timing only); the shipped .so is rebuilt at the end.
cd $GRAFT_REPO_ROOT; ABL="${ABL:-0 4 8 12 16 28}"
for a in $ABL; do
  FCP_BUILD_DEFINES="FCP_HALO_ABLATE=$a" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
  echo "== FCP_HALO_ABLATE=$a"
  python tools/bench_rrdb_layers.py 1024 1024 1 | cut -c1-90
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
