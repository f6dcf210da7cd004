set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
DTF_MS=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/dpprof -- python3 $REPO/tools/dp_world1_timing.py > $REPO/gpurun_out/dpprof.log 2>&1 || { tail -5 $REPO/gpurun_out/dpprof.log; exit 1; }
cd $REPO; python3 tools/prof_summary.py gpurun_out/dpprof 220
