#!/bin/bash
# rocprofv3 kernel table of the default step (1024 images, bf16) + old/new head A/B at the other configurations; via gpurun.
set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/head_prof -- python3 $REPO/bench.py --steps 50 --warmup 5 --prewarm_s 0 > $REPO/gpurun_out/head_prof.log 2>&1 || { tail -5 $REPO/gpurun_out/head_prof.log; exit 1; }
(cd $REPO && python3 tools/prof_summary.py gpurun_out/head_prof 55 > gpurun_out/head_prof.txt && cat gpurun_out/head_prof.txt) || exit 1
cd $REPO
OLD=diag_build/libdtf_kernels_headold.so
for cfg in "--per_gpu_batch 100 --steps 2000" "--per_gpu_batch 256 --steps 2000" "--per_gpu_batch 4096 --steps 300" "--optimizer adam" "--force_dp" "--keep_prob 0.5" "--model wide --steps 200"; do
  for i in 1 2; do
    for lib in $OLD ""; do
      r=$(DTF_KERNELS_LIB=$lib timeout -k 10 200 python bench.py $cfg --warmup 50 2>/dev/null | grep -o '"ms_per_step": [0-9.]*') || exit 1
      echo "[$cfg] lib=${lib:-new}: $r"
    done
  done
done
