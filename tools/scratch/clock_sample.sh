#!/bin/bash
# Sample the GPU's clocks / power while bench.py runs (is the chip clock-limited under the step?); via gpurun.
set -o pipefail
out=gpurun_out/clock_sample.txt; : > $out
echo "## idle" >> $out
timeout -k 5 30 rocm-smi --showclocks --showpower --showuse 2>&1 | grep -v "^$" >> $out
for cfg in "--steps 60000" "--dtype fp32 --steps 12000" "--keep_prob 0.5 --steps 60000"; do
  timeout -k 10 200 python bench.py $cfg --warmup 50 > gpurun_out/clock_bench.log 2>/dev/null &
  pid=$!
  sleep 5   # import + warm-up
  for i in 1 2 3; do
    echo "## bench.py $cfg: sample $i" >> $out
    timeout -k 5 30 rocm-smi --showclocks --showpower --showuse 2>&1 | grep -i "sclk\|mclk\|power\|busy\|fclk" >> $out
    sleep 1
  done
  wait $pid || exit 1
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/clock_bench.log >> $out
done
cat $out
