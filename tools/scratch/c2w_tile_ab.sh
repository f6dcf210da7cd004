#!/bin/bash
# A/B of the conv2 wgrad wave tile (16 ci x 32 co vs 16 ci x 64 co) on one device; run via gpurun.
set -o pipefail
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --steps 300 --warmup 20 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" || exit 1; }
ARGS="--model wide --per_gpu_batch 1024"; run DTF_C2W_NT=2; run DTF_C2W_NT=4
ARGS="--model wide --per_gpu_batch 256";  run DTF_C2W_NT=2; run DTF_C2W_NT=4
ARGS="--per_gpu_batch 1024"; run DTF_C2W_NT=2; run DTF_C2W_NT=4 DTF_C2_G=128; run DTF_C2W_NT=4 DTF_C2_G=64
ARGS="--per_gpu_batch 4096 --steps 200"; run DTF_C2W_NT=2; run DTF_C2W_NT=4 DTF_C2_G=128; run DTF_C2W_NT=4 DTF_C2_G=256
