#!/bin/bash
# GPU box: sample clocks / power while a command runs (evidence for the power-limited clock of the MFMA-bound layers)
#   bash tools/smi_sample.sh gpurun_out/r3/smi.log python bench.py --steps 200 --no-extra --no-cpu-baseline
OUT=$1; shift
"$@" > ${OUT%.log}_cmd.log 2>&1 &
PID=$!
: > $OUT
while kill -0 $PID 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ' ' >> $OUT
  echo >> $OUT
  sleep 0.3
done
wait $PID
grep '^{"metric"' ${OUT%.log}_cmd.log | cut -c1-160
echo "samples: $(wc -l < $OUT)"; sort $OUT | uniq -c | sort -rn | head -8
