"""Reduce a rocprofv3 --kernel-trace CSV of bench.py to (a) the per-launch durations of the last full step
and (b) nothing else — the --stats CSV is copied as is.

    python tools/summarize_trace.py gpurun_out/prof_f16x3/trace profiles/r01_f16x3_bench_step_trace.csv
"""
import csv, glob, re, sys

src, dst = sys.argv[1], sys.argv[2]
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # u8_to_nhwc4 launches per step (= --streams)
path = glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
first = "stem_pool_kernel" if any("stem_pool_kernel" in r["Kernel_Name"] for r in rows) else "u8_to_nhwc4"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]      # first launch of a detection step
# take the last complete timed step (the markers after it belong to the next step / an optional roofline pass)
a, b = starts[-2 * per_step], starts[-per_step]
seg = rows[a:b]
with open(dst, "w") as f:
    f.write("kernel,workgroups,duration_us\n")
    for r in seg:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        f.write(f'"{name}",{wg},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f}\n')
wall = (int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
is_conv = lambda r: any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv3x3_halo", "stem_pool_kernel", "bneck_chain"))
conv = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if is_conv(r)) / 1e6
print(f"step: {len(seg)} launches, {wall:.3f} ms wall, conv engine {conv:.3f} ms in {sum(map(is_conv, seg))} launches")
