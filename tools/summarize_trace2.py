"""Reduce a rocprofv3 --kernel-trace CSV of the TWO-STREAM bench step to per-launch (queue, start, duration) rows of the last
complete step and to overlap statistics: how much of the step has kernels of both detector streams resident at once.

    python tools/summarize_trace2.py gpurun_out/prof_c2/trace2 profiles/r03_f16x3_bench_step_trace_2streams.csv
"""
import csv, glob, re, sys

src, dst = sys.argv[1], sys.argv[2]
path = glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "stem_pool_kernel" in r["Kernel_Name"]]
# a step launches two stems (one per half batch); the last complete timed step = between the 3rd-last pair and the last pair
# (the roofline's single-stream passes come after the timed steps and launch ONE stem each: walk back to the last PAIR)
pairs = [(a, b) for a, b in zip(stems, stems[1:]) if rows[a]["Queue_Id"] != rows[b]["Queue_Id"] and
         int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) < 2_000_000]
a, b = pairs[-2][0], pairs[-1][0]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
with open(dst, "w") as f:
    f.write("kernel,queue,start_us,duration_us,workgroups\n")
    for r in seg:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(
            1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        f.write(f'"{name}",{r["Queue_Id"]},{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},'
                f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f},{wg}\n')
ev = []
for r in seg:
    ev += [(int(r["Start_Timestamp"]), 1), (int(r["End_Timestamp"]), -1)]
ev.sort()
depth, last, busy = 0, ev[0][0], {0: 0, 1: 0, 2: 0}
for t, d in ev:
    busy[min(depth, 2)] += t - last
    depth, last = depth + d, t
wall = int(rows[b]["Start_Timestamp"]) - t0
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"step: {len(seg)} launches on queues {sorted({r['Queue_Id'] for r in seg})}, {wall / 1e6:.3f} ms wall, sum of kernel durations "
      f"{tot / 1e6:.3f} ms; time with >= 2 kernels resident {busy[2] / 1e6:.3f} ms ({busy[2] / wall:.0%}), exactly one {busy[1] / 1e6:.3f} ms, "
      f"none {(wall - busy[1] - busy[2]) / 1e6:.3f} ms")
