#!/bin/bash
# large per-GPU batches: stream layout x LDS-DMA mask, interleaved on one device
B=$1
for rep in 1 2; do
  for cfg in "default:" "ss0:--single_stream" "ss5:--single_stream" "ss7:--single_stream" "ms3:" "ss3:--single_stream"; do
    name=${cfg%%:*}; flag=${cfg#*:}
    case $name in default) unset DTF_GEMM_DMA;; ss0) export DTF_GEMM_DMA=0;; ss5) export DTF_GEMM_DMA=5;; ss7) export DTF_GEMM_DMA=7;; ms3|ss3) export DTF_GEMM_DMA=3;; esac
    ms=$(timeout -k 10 150 python bench.py --per_gpu_batch $B --steps 200 --warmup 20 $flag 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | cut -d' ' -f2)
    echo "B=$B $name: $ms"
  done
done
