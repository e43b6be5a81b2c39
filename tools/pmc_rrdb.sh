#!/bin/bash
# rocprofv3 --pmc passes over tools/bench_rrdb_layers.py (one counter group per pass; no tracing domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_rrdb
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
i=0
for grp in "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python $R/tools/bench_rrdb_layers.py ${1:-1024} > $O/p$i.log 2>&1
  tail -3 $O/p$i.log
done
python - <<PY
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "conv" not in r["Kernel_Name"]:
            continue
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"] if "Grid_Size" in r else "")
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
# keep the last timed launch of every (kernel, grid) config: dispatch ids repeat the same pattern in every pass
seen = collections.OrderedDict()
for (d, n, g), v in rows.items():
    seen[(d,)] = (n, g, v)
with open("$O/summary.txt", "w") as out:
    for (d,), (n, g, v) in seen.items():
        out.write(f"{d} {n} {g} " + " ".join(f"{k}={x:.4g}" for k, x in v.items()) + "\n")
print(open("$O/summary.txt").read()[-6000:])
PY
