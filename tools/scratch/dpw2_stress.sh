#!/bin/bash
# repeat the two-rank data-parallel replay worker with per-step diagnostics (DPW2_DIAG); prints only failing runs
MODE=${1:-adam}; TAIL=${2:-p2p}; N=${3:-12}
for i in $(seq 1 $N); do
  port=$((29700 + i))
  DPW2_DIAG=1 timeout -k 5 100 python -u tests/dp_world2_worker.py 0 $port $MODE $TAIL 2 bf16 > gpurun_out/dpw2_r0_$i.log 2>&1 &
  p0=$!
  DPW2_DIAG=1 timeout -k 5 100 python -u tests/dp_world2_worker.py 1 $port $MODE $TAIL 2 bf16 > gpurun_out/dpw2_r1_$i.log 2>&1
  rc1=$?; wait $p0; rc0=$?
  echo "run $i: rc $rc0 $rc1"
  if [ $rc0 -ne 0 ] || [ $rc1 -ne 0 ]; then grep -h "DIAG\|Assertion" gpurun_out/dpw2_r0_$i.log gpurun_out/dpw2_r1_$i.log | cut -c1-260; fi
done
