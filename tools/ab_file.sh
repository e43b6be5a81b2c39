#!/bin/bash
# GPU box: A/B of one kernel source file in the tree against a saved previous version, within one call (twice):
#   bash tools/ab_file.sh face-crop-plus_amd/csrc/fcp_bneck_chain.hip tools/probes/chain_prev.hip.txt 'python tools/bench_chain.py'
cd $GRAFT_REPO_ROOT
K=$1; P=$2
cp $K /tmp/ab_new.hip
for r in 1 2; do
  for v in new prev; do
    if [ $v = new ]; then cp /tmp/ab_new.hip $K; else cp $P $K; fi
    python face-crop-plus_amd/build_native.py > /dev/null 2>&1
    echo "== $v (run $r)"
    eval "$3"
  done
done
cp /tmp/ab_new.hip $K
python face-crop-plus_amd/build_native.py > /dev/null 2>&1
