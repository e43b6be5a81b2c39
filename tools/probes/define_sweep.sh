#!/bin/bash
# GPU box: sweep one compile-time define of the native library:  bash tools/define_sweep.sh FCP_BIG_NP2 "8 6 4" 'python tools/bench_big.py ...'
cd $GRAFT_REPO_ROOT
for v in $2; do
  FCP_BUILD_DEFINES="$1=$v" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
  echo "== $1=$v"
  eval "$3"
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
