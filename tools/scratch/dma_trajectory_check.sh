#!/bin/bash
for cfg in "4000 1024 sgd" "2000 1024 adam" "3000 256 sgd" "1500 1000 sgd"; do
  for m in 0 5 7; do DTF_GEMM_DMA=$m timeout -k 10 200 python tools/scratch/dma_trajectory_check.py $cfg 2>/dev/null | grep "^mask" || exit 1; done
done
