#!/bin/bash
# Per-stage ROCTx ranges next to the kernel trace: full pipeline, 2 images with RRDB on both.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_markers
mkdir -p $O
rocprofv3 --kernel-trace --marker-trace --output-format csv -d $O/trace -o m -- \
  python $R/bench.py --workload full --enhance all --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --streams 1 > $O/run.log 2>&1
find $O -name "*.csv"
python - <<PY
import csv, glob, collections
mk = glob.glob("$O/trace/**/*marker_api_trace.csv", recursive=True)
kt = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(mk[0]))) if mk else []
ker = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"])) if kt else []
print("marker rows", len(rows), "kernel rows", len(ker))
with open("$O/stage_summary.csv", "w") as f:
    f.write("range,start_ns,end_ns,host_span_us,kernels_launched_inside,device_time_of_those_kernels_us\n")
    # the last occurrence of every stage = the timed step
    last = collections.OrderedDict()
    for r in rows:
        last[r["Function"]] = r
    corr = {}
    for name, r in last.items():
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        # kernels whose launch (approximated by their correlation order / start time) falls after the range opened and
        # before the next range opened cannot be told apart without API tracing: report kernels that START inside [a, b + 50 ms)
        inside = [k for k in ker if a <= int(k["Start_Timestamp"]) < b + 50_000_000]
        f.write(f'{name},{a},{b},{(b - a) / 1e3:.1f},{len(inside)},{sum(int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) for k in inside) / 1e3:.1f}\n')
print(open("$O/stage_summary.csv").read())
PY
