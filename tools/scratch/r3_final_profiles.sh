#!/bin/bash
# rocprofv3 kernel tables of the round-3 build: the fused step (bf16, fp32) and the data-parallel step on one rank
# (collapsed, with the N > 1 kernels kept, and with the dense shard of an 8-rank job).  Per-call averages are the
# numbers to read: the data-parallel runs include bench.py's path trials.  Run via gpurun.
set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
run() {  # tag, env assignment or "-", bench args...
  tag=$1; envs=$2; shift 2
  if [ "$envs" != "-" ]; then export $envs; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r3fin_$tag -- python3 $REPO/bench.py "$@" --steps 200 --warmup 20 --prewarm_s 0 > $REPO/gpurun_out/r3fin_$tag.log 2>&1 || { tail -5 $REPO/gpurun_out/r3fin_$tag.log; exit 1; }
  if [ "$envs" != "-" ]; then unset ${envs%%=*}; fi
  (cd $REPO && { echo "# bench.py $* --steps 200 --warmup 20 --prewarm_s 0 ${envs/#-/}"; grep -h "^{\"metric" gpurun_out/r3fin_$tag.log; python3 tools/prof_summary.py gpurun_out/r3fin_$tag 220; } > gpurun_out/r3_final_kernel_table_$tag.txt && echo "== $tag" && cat gpurun_out/r3_final_kernel_table_$tag.txt) || exit 1
}
run bf16_1024 - --no_fp32 --no_graph --single_stream &&
run fp32_1024 - --dtype fp32 --no_graph --single_stream &&
run fp32_100 - --dtype fp32 --no_graph --single_stream --per_gpu_batch 100 &&
run wide_bf16_1024 - --model wide --no_fp32 --no_graph --single_stream &&
run dp1_collapsed - --force_dp --no_fp32 &&
run dp1_full_kernels - --force_dp --no_collapse --no_fp32 &&
run dp1_shard_of_8 DTF_DP_FAKE_WORLD=8 --force_dp --no_collapse --no_fp32
