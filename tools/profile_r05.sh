#!/bin/bash
# GPU box: the round's rocprofv3 evidence for the headline command, reduced on the box into gpurun_out/r05/ (small files
# only; copy them into profiles/ afterwards):  bash tools/profile_r05.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
bash $R/tools/profile_workload.sh f16x3 all > $O/profile_workload.log 2>&1
cd $R
python tools/summarize_trace.py gpurun_out/prof_f16x3/trace $O/r05_f16x3_bench_step_trace.csv 2>&1 | tee $O/summarize_trace.log
cp $(find gpurun_out/prof_f16x3/trace -name "*kernel_stats.csv" | head -1) $O/r05_f16x3_bench_kernel_stats.csv
python tools/summarize_trace2.py gpurun_out/prof_f16x3/trace2 $O/r05_f16x3_bench_step_trace_2streams.csv 2>&1 | tee $O/summarize_trace2.log
L=$(python -c "import json;print(json.load(open('gpurun_out/prof_f16x3/bench_line_under_trace.json'))['roofline']['launches_per_step'])")
python tools/summarize_pmc.py gpurun_out/prof_f16x3 $O/r05_f16x3_pmc_conv.json $L 2>&1 | tee $O/summarize_pmc.log
cp gpurun_out/prof_f16x3/bench_line_under_trace.json $O/bench_line_under_trace.json
python bench.py --launch-table $O/r05_f16x3_launch_table.csv > $O/bench_line.json 2> $O/bench.err
rm -rf gpurun_out/prof_f16x3          # bulky raw traces: not needed once reduced
# the north-star geometry (batch 32 @1024^2) and configs[2] without RRDB: counter passes only
bash $R/tools/profile_workload.sh c3det pmc --workload detect --batch 32 --size 1024 > $O/profile_c3det.log 2>&1
L=$(python -c "import json;print(json.load(open('gpurun_out/prof_c3det/bench_line_under_pmc.json'))['roofline']['launches_per_step'])")
python tools/summarize_pmc.py gpurun_out/prof_c3det $O/r05_c3det_pmc.json $L 32 1024 "bench.py --workload detect --batch 32 --size 1024 --steps 1 --warmup 1 --streams 1" 2>&1 | tee $O/summarize_pmc_c3det.log
rm -rf gpurun_out/prof_c3det
bash $R/tools/profile_workload.sh c3 pmc --workload full --enhance none > $O/profile_c3.log 2>&1
L=$(python -c "import json;print(json.load(open('gpurun_out/prof_c3/bench_line_under_pmc.json'))['roofline']['launches_per_step'])")
python tools/summarize_pmc.py gpurun_out/prof_c3 $O/r05_c3_pmc.json $L 32 1024 "bench.py --workload full --enhance none --steps 1 --warmup 1 --streams 1" 2>&1 | tee $O/summarize_pmc_c3.log
rm -rf gpurun_out/prof_c3
ls -la $O
