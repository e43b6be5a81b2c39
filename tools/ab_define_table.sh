#!/bin/bash
# In-call A/B of a build-time define of the native library: the headline bench (with its per-launch table) on the product build
# and on a build with -D<DEFINE>, twice each, interleaved, inside ONE gpurun call (boxes of the pool differ by a few percent).
#
#   bash tools/ab_define.sh <DEFINE> <outdir> [extra bench.py flags ...]
#
# e.g. FCP_EPI_SERIAL = the round-3/4 register epilogue (one set at a time) against round 5's grouped one.
set -u
def=$1
out=${2:-gpurun_out/ab}
shift 2
mkdir -p "$out"
run() {   # tag [bench flags]
  local tag=$1; shift
  python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --launch-table "$out/launch_$tag.csv" "$@" > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
  python - "$out/bench_$tag.json" "$tag" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], "faces/s", d["value"], "ms/step", d["ms_per_step"], "conv_ms", r["conv_ms_per_step"], "frac", r["frac"], "frac_timed", r["frac_timed"], flush=True)
P
}
extra=("$@")
run_t() { local t=$1; run "$t" "${extra[@]}"; }
run_t new1
FCP_BUILD_DEFINES=$def python face-crop-plus_amd/build_native.py --force > "$out/build_old.log" 2>&1
run_t old1
python face-crop-plus_amd/build_native.py --force > "$out/build_new.log" 2>&1
run_t new2
FCP_BUILD_DEFINES=$def python face-crop-plus_amd/build_native.py --force >> "$out/build_old.log" 2>&1
run_t old2
python face-crop-plus_amd/build_native.py --force >> "$out/build_new.log" 2>&1
python - "$out" "$def" <<'P'
import csv, sys
o = sys.argv[1]
rd = lambda t: {r["launch"] + " #" + str(i): float(r["us"]) for i, r in enumerate(csv.DictReader(open(f"{o}/launch_{t}.csv")))}
n1, o1, n2, o2 = rd("new1"), rd("old1"), rd("new2"), rd("old2")
print(f"{'launch (single-stream roofline passes)':74s} {'-D' + sys.argv[2]:>16s} {'product':>8s}  delta")
for k in n1:
    a, b = (o1[k] + o2[k]) / 2, (n1[k] + n2[k]) / 2
    if abs(a - b) > 0.03 * a:
        print(f"{k:74s} {a:16.1f} {b:8.1f} {100 * (b - a) / a:+6.1f} %")
print("conv ms per step: define", round(sum(o1.values()) / 1e3, 3), round(sum(o2.values()) / 1e3, 3), " product", round(sum(n1.values()) / 1e3, 3), round(sum(n2.values()) / 1e3, 3))
P
