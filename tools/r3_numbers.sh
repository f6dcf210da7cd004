#!/bin/bash
# Round-3 number collection on one MI355X (run through gpurun): every line lands in gpurun_out/r3_final_numbers.txt
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3_final_numbers.txt
: > $OUT
run() { echo "## $*" >> $OUT; timeout -k 10 200 "$@" 2>gpurun_out/r3_numbers_last_stderr.log | tail -1 >> $OUT || { echo "FAILED: $*" >> $OUT; tail -20 gpurun_out/r3_numbers_last_stderr.log; exit 1; }; }
run python bench.py --steps 1000 --warmup 50
run python bench.py --steps 20 --warmup 5
run python bench.py --force_dp --steps 1000 --warmup 50
run python bench.py --force_dp --no_collapse --steps 1000 --warmup 50 --no_fp32
DTF_DP_FAKE_WORLD=8 run python bench.py --force_dp --no_collapse --steps 1000 --warmup 50 --no_fp32
run python bench.py --optimizer adam --steps 500 --warmup 50 --no_fp32
run python bench.py --force_dp --optimizer adam --steps 500 --warmup 50 --no_fp32
run python bench.py --per_gpu_batch 100 --steps 1000 --warmup 50
run python bench.py --per_gpu_batch 4096 --steps 200 --warmup 20
run python bench.py --model wide --steps 100 --warmup 10 --no_fp32
run python bench.py --dtype fp32 --per_gpu_batch 100 --steps 1000 --warmup 50
run python bench.py --dtype fp32 --per_gpu_batch 4096 --steps 100 --warmup 10
run python bench.py --dtype fp32 --model wide --steps 30 --warmup 5
echo "## tools/scratch/dp_host.py" >> $OUT
timeout -k 10 200 python tools/scratch/dp_host.py 2>/dev/null | grep "host" >> $OUT
cat $OUT | cut -c1-330
