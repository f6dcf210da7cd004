#!/bin/bash
# Canonical bench lines of the current build (one device, back to back); run via gpurun.
set -o pipefail
out=gpurun_out/final_numbers.txt; : > $out
run() { echo "## bench.py $*" >> $out; timeout -k 10 240 python bench.py "$@" 2>/dev/null >> $out || { echo "FAILED: $*" >> $out; exit 1; }; }
run
run --steps 20 --warmup 5
run --dtype fp32 --steps 300
run --per_gpu_batch 100 --steps 2000
run --per_gpu_batch 100 --steps 2000 --dtype fp32
run --per_gpu_batch 128 --steps 2000
run --per_gpu_batch 256 --steps 2000
run --per_gpu_batch 512 --steps 2000
run --per_gpu_batch 4096 --steps 300
run --per_gpu_batch 4096 --steps 100 --dtype fp32
run --optimizer adam
run --keep_prob 0.5
run --force_dp
run --force_dp --optimizer adam
run --model wide --steps 200
cat $out
