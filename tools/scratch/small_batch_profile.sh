#!/bin/bash
# rocprofv3 kernel tables of the small-batch step (BASELINE config 2, batch 100), both precisions; run via gpurun.
set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for dt in fp32 bf16; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/sb_prof_$dt -- python3 $REPO/bench.py --dtype $dt --per_gpu_batch 100 --steps 50 --warmup 5 --prewarm_s 0 --no_graph --single_stream > $REPO/gpurun_out/sb_prof_$dt.log 2>&1 || { tail -5 $REPO/gpurun_out/sb_prof_$dt.log; exit 1; }
  (cd $REPO && python3 tools/prof_summary.py gpurun_out/sb_prof_$dt 55 > gpurun_out/sb_prof_$dt.txt && cat gpurun_out/sb_prof_$dt.txt)
done
