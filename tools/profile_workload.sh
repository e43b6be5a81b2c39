#!/bin/bash
# GPU box: rocprofv3 evidence for one bench workload under gpurun_out/prof_<tag>/ :
#   bash tools/profile_workload.sh <tag> [trace|pmc|all] <bench.py workload flags...>
#   e.g. bash tools/profile_workload.sh c3det all --workload detect --batch 32 --size 1024
# Counter passes are separate runs with --pmc only (no tracing domains), as gpurun requires.  Counters and the per-launch
# trace use --streams 1 (kernels own the device, launches of a step are contiguous); `trace2` is the same step on the
# product's two detector streams (overlap evidence).
TAG=$1; WHAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BASE="python $R/bench.py --no-cpu-baseline --no-extra $*"
if [ "$WHAT" = trace ] || [ "$WHAT" = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- $BASE --steps 3 --warmup 2 --streams 1 > $O/bench_trace.log 2>&1
  grep '^{"metric"' $O/bench_trace.log | tail -1 > $O/bench_line_under_trace.json
  rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -o b -- $BASE --steps 3 --warmup 2 --streams 2 > $O/bench_trace2.log 2>&1
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  CMD1="$BASE --steps 1 --warmup 1 --streams 1"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- $CMD1 > $O/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- $CMD1 > $O/write.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o b -- $CMD1 > $O/sq.log 2>&1
  grep '^{"metric"' $O/sq.log | tail -1 > $O/bench_line_under_pmc.json
fi
# keep what the reducers need, drop the bulky rest (gpurun_out is capped at 64 MiB)
find $O -name "*agent_info.csv" -delete
ls $O
