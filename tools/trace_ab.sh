#!/bin/bash
# kernel-trace the bench with and without the 256-row tiles; outputs under gpurun_out/trace_ab/{big,nobig}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  FCP_BIG_TILES=$v rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_ab/big$v -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $R/gpurun_out/trace_ab_$v.log 2>&1
  tail -1 $R/gpurun_out/trace_ab_$v.log | cut -c1-200
done
