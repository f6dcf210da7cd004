set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/wide_prof -- python3 $REPO/bench.py --model wide --per_gpu_batch 1024 --steps 20 --warmup 3 --no_graph --single_stream > $REPO/gpurun_out/wide_prof.log 2>&1 || { tail -5 $REPO/gpurun_out/wide_prof.log; exit 1; }
cd $REPO; python3 tools/prof_summary.py gpurun_out/wide_prof 23
