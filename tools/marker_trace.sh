# roctx marker trace of a short CLI run (utils/trace.py): host ranges next to the kernels they enqueue.
# usage (on the GPU box): bash tools/marker_trace.sh  -> gpurun_out/marker/
set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $REPO/gpurun_out/marker -- \
  python3 $REPO/mnistdist.py --synthetic --steps 50 --batch_size 1024 --trace --display_step 25 --logdir $REPO/gpurun_out/marker_logs \
  > $REPO/gpurun_out/marker.log 2>&1 || { tail -5 $REPO/gpurun_out/marker.log; exit 1; }
cd $REPO
f=$(ls gpurun_out/marker/*/*marker_api_trace.csv | head -1)
echo "marker rows: $(wc -l < $f)"; head -8 $f
