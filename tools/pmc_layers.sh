#!/bin/bash
# rocprofv3 --pmc passes over tools/bench_layers.py (one counter group per pass; no tracing domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-l3c1,l3c2,l3c3,l4c1}
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_layers/p$i -- python $R/tools/bench_layers.py $L 2 > $R/gpurun_out/pmc_layers_p$i.log 2>&1
  tail -2 $R/gpurun_out/pmc_layers_p$i.log
done
find $R/gpurun_out/pmc_layers -name "*counter_collection.csv" | head
