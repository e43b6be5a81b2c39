#!/bin/bash
# GPU box: THE round's evidence in one call —  bash tools/profile_round.sh r06   (results, reduced on the box, in
# gpurun_out/<tag>/; copy them into profiles/).  Regenerates every file bench.py quotes as `roofline.traffic`
# (bench.TRAFFIC_KEYS: c3det_pmc = the headline, f16x3_pmc_conv = configs[1], c3_pmc, rrdb_pmc, f32_pmc_conv), each
# stamped with the ABI version and the conv-launch count it was recorded at, plus the rocprofv3 kernel trace / --stats of the
# headline step (one and two detector streams), the per-launch table with its HBM / MFMA floors, the per-launch
# clock / power / joule ledger, and the bench lines of the same build.
# Counter passes are separate runs with --pmc only (no tracing domains), as gpurun requires.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
launches() { python -c "import json,sys;print(json.load(open(sys.argv[1]))['roofline']['launches_per_step'])" $1; }
pmc() {   # key, prof tag, cal batch, cal side, description, bench flags...
  local key=$1 tag=$2 n=$3 side=$4 what=$5; shift 5
  bash tools/profile_workload.sh $tag pmc "$@" > $O/profile_$tag.log 2>&1
  local L=$(launches gpurun_out/prof_$tag/bench_line_under_pmc.json)
  python tools/summarize_pmc.py gpurun_out/prof_$tag $O/${TAG}_$key.json $L $n $side "$what --steps 1 --warmup 1 --streams 1" > $O/summarize_$key.log 2>&1 || tail -3 $O/summarize_$key.log
  cp gpurun_out/prof_$tag/bench_line_under_pmc.json $O/bench_line_under_pmc_$tag.json
}
# ---- headline: detect + align + crop, batch 32 @1024^2 (bench.py's default): traces + counters
bash tools/profile_workload.sh c3det trace > $O/profile_c3det_trace.log 2>&1
python tools/summarize_trace.py gpurun_out/prof_c3det/trace $O/${TAG}_c3det_step_trace.csv 2>&1 | tee $O/summarize_trace.log
cp $(find gpurun_out/prof_c3det/trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_c3det_kernel_stats.csv
python tools/summarize_trace2.py gpurun_out/prof_c3det/trace2 $O/${TAG}_c3det_step_trace_2streams.csv 2>&1 | tee $O/summarize_trace2.log
cp gpurun_out/prof_c3det/bench_line_under_trace.json $O/bench_line_under_trace_c3det.json
rm -rf gpurun_out/prof_c3det/trace gpurun_out/prof_c3det/trace2
pmc c3det_pmc c3det 32 1024 "bench.py (default: detect, batch 32 @1024x1024)"
rm -rf gpurun_out/prof_c3det
# ---- BASELINE configs[1]: batch 64 @640^2 (trace of one stream + counters)
bash tools/profile_workload.sh c2 trace --batch 64 --size 640 > $O/profile_c2_trace.log 2>&1
python tools/summarize_trace.py gpurun_out/prof_c2/trace $O/${TAG}_f16x3_bench_step_trace.csv 2>&1 | tee -a $O/summarize_trace.log
cp $(find gpurun_out/prof_c2/trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_f16x3_bench_kernel_stats.csv
rm -rf gpurun_out/prof_c2/trace gpurun_out/prof_c2/trace2
pmc f16x3_pmc_conv c2 64 640 "bench.py --batch 64 --size 640" --batch 64 --size 640
rm -rf gpurun_out/prof_c2
# ---- configs[2] without RRDB, RRDB on every image (batch 2), exact-fp32 mode: counters only
pmc c3_pmc c3 32 1024 "bench.py --workload full --enhance none" --workload full --enhance none
rm -rf gpurun_out/prof_c3
pmc rrdb_pmc rrdb 2 1024 "bench.py --workload full --enhance all --batch 2" --workload full --enhance all --batch 2
rm -rf gpurun_out/prof_rrdb
pmc f32_pmc_conv f32 64 640 "bench.py --batch 64 --size 640 --precision f32" --batch 64 --size 640 --precision f32
rm -rf gpurun_out/prof_f32
# ---- per-launch table (in-situ events + floors) of the headline and of configs[1]; the full bench line of this build
python bench.py --launch-table $O/${TAG}_c3det_launch_table.csv > $O/bench_line.json 2> $O/bench.err
python bench.py --batch 64 --size 640 --no-extra --no-cpu-baseline --launch-table $O/${TAG}_f16x3_launch_table.csv > $O/bench_line_c2.json 2> $O/bench_c2.err
# ---- per-launch ledger: every launch of the step alone in a steady loop, clock / power sampled -> W, MHz, joules, floors
python tools/launch_ledger.py $O/${TAG}_c3det_launch_ledger.csv 32 1024 0.4 > $O/${TAG}_c3det_launch_ledger.txt 2>&1
python tools/launch_ledger.py $O/${TAG}_f16x3_launch_ledger.csv 64 640 0.4 > $O/${TAG}_f16x3_launch_ledger.txt 2>&1
tail -12 $O/${TAG}_c3det_launch_ledger.txt
grep -h '^{"metric"' $O/bench_line.json $O/bench_line_c2.json | cut -c1-400 > $O/${TAG}_bench_lines.jsonl
ls -la $O; du -sh $R/gpurun_out
