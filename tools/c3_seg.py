import csv, glob
rows = sorted(csv.DictReader(open(glob.glob("gpurun_out/prof_c3/trace/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "stem_pool_kernel" in r["Kernel_Name"]]
seg = rows[stems[-2]:stems[-1]]
for r in seg:
    if any(k in r["Kernel_Name"] for k in ("avgpool", "label_hist", "parse_tail", "bise_pre")):
        print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
