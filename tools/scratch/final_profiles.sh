#!/bin/bash
# rocprofv3 kernel tables of the final build: bf16 and fp32 at 1024 and 100 images; run via gpurun.
set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for cfg in "bf16 1024" "fp32 1024" "bf16 100" "fp32 100"; do
  set -- $cfg; dt=$1; b=$2
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/fin_${dt}_$b -- python3 $REPO/bench.py --dtype $dt --per_gpu_batch $b --steps 50 --warmup 5 --prewarm_s 0 > $REPO/gpurun_out/fin_${dt}_$b.log 2>&1 || { tail -5 $REPO/gpurun_out/fin_${dt}_$b.log; exit 1; }
  (cd $REPO && python3 tools/prof_summary.py gpurun_out/fin_${dt}_$b 55 > gpurun_out/fin_${dt}_$b.txt && echo "== $dt batch $b" && cat gpurun_out/fin_${dt}_$b.txt) || exit 1
done
