#!/bin/bash
# rocprofv3 kernel table of the 4096-image step (BASELINE config 2) in its default mode (graph + fork/join) and single-stream eager; via gpurun.
set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for mode in "default" "--no_graph --single_stream"; do
  tag=$(echo $mode | tr -d ' -'); m=$mode; [ "$mode" = default ] && m=""
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/b4096_$tag -- python3 $REPO/bench.py --per_gpu_batch 4096 --steps 50 --warmup 5 --prewarm_s 0 $m > $REPO/gpurun_out/b4096_$tag.log 2>&1 || { tail -5 $REPO/gpurun_out/b4096_$tag.log; exit 1; }
  (cd $REPO && echo "== $mode" && python3 tools/prof_summary.py gpurun_out/b4096_$tag 55 && grep -o '"ms_per_step": [0-9.]*' gpurun_out/b4096_$tag.log)
done
