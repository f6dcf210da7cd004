#!/bin/bash
# Two PMC passes over the eager single-stream step (one rocprofv3 run per counter set; never mixed with trace flags).
# Usage (on the GPU box, via gpurun): bash tools/pmc_collect.sh <tag> [extra bench.py arguments, e.g. --dtype fp32]
set -o pipefail
TAG=${1:-pmc}
shift || true
EXTRA="$@"
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM \
  --output-format csv -d $REPO/gpurun_out/${TAG}_sq -- python3 $REPO/bench.py --steps 20 --warmup 3 --prewarm_s 0 --no_graph --single_stream $EXTRA > $REPO/gpurun_out/${TAG}_sq.log 2>&1 || { tail -5 $REPO/gpurun_out/${TAG}_sq.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum \
  --output-format csv -d $REPO/gpurun_out/${TAG}_tcc -- python3 $REPO/bench.py --steps 20 --warmup 3 --prewarm_s 0 --no_graph --single_stream $EXTRA > $REPO/gpurun_out/${TAG}_tcc.log 2>&1 || { tail -5 $REPO/gpurun_out/${TAG}_tcc.log; exit 1; }
cd $REPO
python3 tools/pmc_summary.py gpurun_out/${TAG}_sq > gpurun_out/${TAG}_sq.txt
python3 tools/pmc_summary.py gpurun_out/${TAG}_tcc > gpurun_out/${TAG}_tcc.txt
cat gpurun_out/${TAG}_sq.txt gpurun_out/${TAG}_tcc.txt
