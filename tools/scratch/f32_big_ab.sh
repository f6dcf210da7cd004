# exact-fp32 step at 4096 images: eager single stream (default) vs HIP graph with the backward fork/join (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
run() { python bench.py --dtype fp32 --per_gpu_batch $1 --steps 100 --warmup 10 "${@:3}" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d['config'].get('graph'), d['config'].get('streams'))"; }
for b in 2048 4096; do
timeout -k 10 200 bash -c "$(declare -f run); run $b 'fp32 $b default          '" || exit 1
timeout -k 10 200 bash -c "$(declare -f run); run $b 'fp32 $b graph            ' --graph" || exit 1
timeout -k 10 200 bash -c "$(declare -f run); run $b 'fp32 $b graph + streams  ' --graph --multi_stream" || exit 1
timeout -k 10 200 bash -c "$(declare -f run); run $b 'fp32 $b eager + streams  ' --multi_stream" || exit 1
done
