#!/bin/bash
# Step time by LDS-DMA mask (DTF_GEMM_DMA: bit 0 NT dgrad, bit 1 NN forward, bit 2 TN wgrad) on one device, interleaved.
# usage: gemm_dma_mask_ab.sh "<masks>" <bench.py arguments>
MASKS=$1; shift
ARGS="$@"
for rep in 1 2 3; do
  for m in $MASKS; do
    ms=$(DTF_GEMM_DMA=$m timeout -k 10 150 python bench.py $ARGS 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | cut -d' ' -f2)
    echo "mask $m: $ms"
  done
done
