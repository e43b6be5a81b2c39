#!/bin/bash
# GPU box: the round's rocprofv3 evidence in one call — traces (one and two streams) + PMC passes of the headline (configs[1]),
# of the 1024^2 detect step, of configs[2] without enhancement, and PMC passes of the RRDB workload (two images).
cd $GRAFT_REPO_ROOT
bash tools/profile_workload.sh c2 all
bash tools/profile_workload.sh c3det all --workload detect --batch 32 --size 1024
bash tools/profile_workload.sh c3 all --workload full --enhance none
bash tools/profile_workload.sh rrdb pmc --workload full --enhance all --batch 2
python bench.py --launch-table gpurun_out/launch_table.csv 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/bench_line_final.json
cut -c1-200 gpurun_out/bench_line_final.json
du -sh gpurun_out
