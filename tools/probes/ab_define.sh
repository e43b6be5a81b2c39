#!/bin/bash
# GPU box: A/B of a compile-time define of the native library within one call (interleaved, twice):
#   bash tools/ab_define.sh FCP_BIG_V1 'python bench.py --no-extra --no-cpu-baseline'
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for d in "" "$1"; do
    FCP_BUILD_DEFINES="$d" python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
    echo "== defines: '$d' (run $r)"
    eval "$2"
  done
done
python face-crop-plus_amd/build_native.py --force > /dev/null 2>&1
