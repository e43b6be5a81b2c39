"""Effective shader clock and matrix-pipe utilisation per conv launch of the last step of a `--pmc SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE` pass (tools/profile_workload.sh <tag> pmc ...): clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration
(MI355X_MICROARCH.md, DVFS: "effective clock = GRBM_GUI_ACTIVE / kernel wall time"); busy = MFMA busy cycles / (1024 SIMDs x
cycles).  Launch durations under counter collection are longer than un-profiled ones; the clock estimate is a ratio of two
quantities of the same launch.

    python tools/summarize_clock.py gpurun_out/prof_c2 53 profiles/r03_f16x3_clock_per_launch.csv"""
import collections, csv, glob, re, sys

src, launches, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
d = collections.OrderedDict()
for r in csv.DictReader(open(glob.glob(f"{src}/sq/**/*counter_collection.csv", recursive=True)[0])):
    e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
conv = [v for v in d.values() if any(k in v["name"] for k in ("conv_igemm", "conv3x3_halo", "stem_pool_kernel", "bneck_chain"))][-launches:]
with open(dst, "w") as f:
    f.write("kernel,duration_us,effective_clock_ghz,mfma_busy_frac\n")
    tot_t = tot_c = tot_b = 0.0
    for v in conv:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", v["name"]).split("(")[0]
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        f.write(f'"{name}",{v["t"] / 1e3:.1f},{cyc / v["t"]:.3f},{v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024):.3f}\n')
        tot_t += v["t"]; tot_c += cyc; tot_b += v["SQ_VALU_MFMA_BUSY_CYCLES"]
print(f"{len(conv)} launches: {tot_t / 1e6:.3f} ms, mean effective clock {tot_c / tot_t:.3f} GHz, matrix pipe busy {tot_b / (tot_c * 1024):.3f}")
