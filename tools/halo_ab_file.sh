#!/bin/bash
# GPU box: A/B of the halo kernel source in the tree against tools/probes/halo_prev.hip.txt within one call
cd $GRAFT_REPO_ROOT
K=face-crop-plus_amd/csrc/fcp_conv_f16x3_halo.hip
cp $K /tmp/halo_new.hip
for r in 1 2; do
  for v in new prev; do
    if [ $v = new ]; then cp /tmp/halo_new.hip $K; else cp tools/probes/halo_prev.hip.txt $K; fi
    python face-crop-plus_amd/build_native.py > /dev/null 2>&1
    echo "== $v (run $r)"
    python tools/bench_rrdb_layers.py 1024 1024 1 | cut -c1-60
    python tools/bench_rrdb_tail.py 1024 f16x3 256 | tail -1
  done
done
cp /tmp/halo_new.hip $K
python face-crop-plus_amd/build_native.py > /dev/null 2>&1
