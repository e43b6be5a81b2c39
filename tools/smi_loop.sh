#!/bin/bash
# GPU box: power / sclk while ONE kernel runs in a loop:  bash tools/smi_loop.sh "chain pair2 pair3 big3x3 halo copy"
cd $GRAFT_REPO_ROOT
for k in $1; do
  python tools/loop_kernel.py $k 5 > /tmp/loop_$k.log 2>&1 &
  PID=$!
  sleep 2.5
  S=""
  for i in 1 2 3 4; do
    S="$S | $(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Power:" | grep -oE "\([0-9]+Mhz\)|[0-9]+\.[0-9]+ ?W?$" | tr -d '()' | tr '\n' ' ')"
    sleep 0.4
  done
  wait $PID
  echo "$(grep 'per launch' /tmp/loop_$k.log) $S"
done
