#!/bin/bash
# GPU box: the round's rocprofv3 evidence for the headline command, reduced on the box into gpurun_out/r04/ (small files
# only; copy them into profiles/ afterwards):  bash tools/profile_r04.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
bash $R/tools/profile_workload.sh f16x3 all > $O/profile_workload.log 2>&1
cd $R
python tools/summarize_trace.py gpurun_out/prof_f16x3/trace $O/r04_f16x3_bench_step_trace.csv 2>&1 | tee $O/summarize_trace.log
cp $(find gpurun_out/prof_f16x3/trace -name "*kernel_stats.csv" | head -1) $O/r04_f16x3_bench_kernel_stats.csv
python tools/summarize_trace2.py gpurun_out/prof_f16x3/trace2 $O/r04_f16x3_bench_step_trace_2streams.csv 2>&1 | tee $O/summarize_trace2.log
python tools/summarize_pmc.py gpurun_out/prof_f16x3 $O/r04_f16x3_pmc_conv.json 53 2>&1 | tee $O/summarize_pmc.log
cp gpurun_out/prof_f16x3/bench_line_under_trace.json $O/bench_line_under_trace.json
python bench.py --launch-table $O/r04_f16x3_launch_table.csv > $O/bench_line.json 2> $O/bench.err
rm -rf gpurun_out/prof_f16x3          # bulky raw traces: not needed once reduced
ls -la $O
