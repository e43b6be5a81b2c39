python bench.py --steps 400 --warmup 3 --no-cpu-baseline > gpurun_out/clk_bench.log 2>&1 &
BP=$!
sleep 25
for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.7; done
wait $BP
tail -1 gpurun_out/clk_bench.log | cut -c1-300
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power"
