#!/bin/bash
# 1-GPU measurement sweep for BASELINE.md: batch sizes, optimizer / dropout / model variants,
# the PyTorch-ROCm eager reference on the same box.  Usage: bash tools/bench_sweep.sh <tag>
set -o pipefail
tag=${1:-sweep}
out=gpurun_out/${tag}.log
mkdir -p gpurun_out
: > $out
run() { echo "## $*" >> $out; timeout -k 10 150 "$@" >> $out 2>/dev/null || exit 1; }
for b in 100 128 256 512 1024 2048 4096 8192; do run python bench.py --per_gpu_batch $b --steps 1000 --warmup 100; done
run python bench.py --graph --steps 1000 --warmup 100
run python bench.py --multi_stream --steps 1000 --warmup 100
run python bench.py --optimizer adam --steps 1000 --warmup 100
run python bench.py --keep_prob 0.5 --steps 1000 --warmup 100
run python bench.py --model wide --per_gpu_batch 256 --steps 50 --warmup 10
run python bench.py --model wide --per_gpu_batch 1024 --steps 30 --warmup 5
run python bench/torch_reference.py --batch 1024 --bf16 --channels_last
run python bench/torch_reference.py --batch 1024
run python bench/torch_reference.py --batch 100
run env DTF_MS=0 python tools/dp_world1_timing.py
grep -E '^##|"value"|img/s|ms' $out | cut -c1-230
