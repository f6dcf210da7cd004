# one SQ pass with issue counters over the eager fp32 step: tools/scratch/pmc_valu.sh <tag> [bench args]   (run via gpurun)
set -o pipefail
TAG=${1:-valu}; shift || true
REPO=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --output-format csv -d $REPO/gpurun_out/${TAG}_sq -- python3 $REPO/bench.py --steps 20 --warmup 3 --prewarm_s 0 --no_graph --single_stream "$@" > $REPO/gpurun_out/${TAG}_sq.log 2>&1 || { tail -5 $REPO/gpurun_out/${TAG}_sq.log; exit 1; }
cd $REPO && python3 tools/pmc_summary.py gpurun_out/${TAG}_sq | tee gpurun_out/${TAG}_sq.txt
