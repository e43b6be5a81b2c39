#!/bin/bash
# Regenerate the rocprofv3 evidence under gpurun_out/prof_<tag>/ for one precision (run on the GPU box):
#   bash tools/profile_round.sh f16x3      (then tools/summarize_trace.py / tools/summarize_pmc.py reduce it)
# Counter passes are separate runs with --pmc only (no tracing domains), as gpurun requires.  Traces and counters
# are taken with --streams 1 (kernels own the device, launches of a step are contiguous); the headline line of the
# same binary (two detector streams, extras) is recorded last, un-profiled.
P=${1:-f16x3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$P
mkdir -p $O
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra --streams 1 --precision $P"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- $CMD > $O/bench_trace.log 2>&1
grep '^{"metric"' $O/bench_trace.log | tail -1 > $O/bench_line_under_trace.json
CMD1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --streams 1 --precision $P"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- $CMD1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- $CMD1 > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o b -- $CMD1 > $O/sq.log 2>&1
python $R/bench.py --precision $P 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_line.json
find $O -name "*.csv" | head -20
cut -c1-260 $O/bench_line.json
