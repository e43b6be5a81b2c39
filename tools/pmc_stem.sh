#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_stem/p$i -- python $R/tools/bench_stem.py > $R/gpurun_out/pmc_stem_p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for p in sorted(glob.glob(R+'/gpurun_out/pmc_stem/p*/**/*counter_collection.csv', recursive=True)):
    d=collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        if 'stem_pool' in r['Kernel_Name']:
            d.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
    k=list(d)[-1]
    print({a:f"{b:.4g}" for a,b in d[k].items()})
PY
