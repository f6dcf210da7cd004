#!/bin/bash
# Soak of the push parameter server: 1 ps + 2 workers on one GPU, staleness 1, 4000 global steps.
set -o pipefail
P0=23411; P1=23412; P2=23413
COMMON="--ps_hosts=127.0.0.1:$P0 --worker_hosts=127.0.0.1:$P1,127.0.0.1:$P2 --backend=hip --comm_backend=gloo --synthetic --mode=async --async_impl=push --staleness=${STALENESS:-1} --batch_size=128 --steps=4000 --display_step=1000 --learning_rate=0.01 --logdir=/tmp/soak_logs_$$ --noeval"
timeout -k 10 280 python -u mnistdist.py --job_name=ps --task_index=0 $COMMON > gpurun_out/soak_ps.log 2>&1 &
A=$!
timeout -k 10 280 python -u mnistdist.py --job_name=worker --task_index=0 $COMMON > gpurun_out/soak_w0.log 2>&1 &
B=$!
timeout -k 10 280 python -u mnistdist.py --job_name=worker --task_index=1 $COMMON > gpurun_out/soak_w1.log 2>&1 &
C=$!
wait $A; ra=$?; wait $B; rb=$?; wait $C; rc=$?
echo "exit codes ps=$ra w0=$rb w1=$rc"
grep -h "applied\|Optimization Finished\|step:" gpurun_out/soak_ps.log gpurun_out/soak_w0.log gpurun_out/soak_w1.log | tail -12
[ $ra -eq 0 ] && [ $rb -eq 0 ] && [ $rc -eq 0 ]
