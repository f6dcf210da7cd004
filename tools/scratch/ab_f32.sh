# fp32 kernel set: tests, then the step of the in-tree library against diag_build/libdtf_kernels_prev.so; run via gpurun
# usage: ab_f32.sh [pytest -k expression] ; BATCHES="1024 100" by default
set -o pipefail
cd $GRAFT_REPO_ROOT
K=${1:-"f32"}
timeout -k 10 400 python -u -m pytest tests/test_kernels_f32_gpu.py tests/test_engine_f32_gpu.py -q -x --timeout 120 --timeout-method thread -k "$K" 2>&1 | tail -3 || exit 1
one() { python bench.py --dtype fp32 --per_gpu_batch $2 --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $2 $1', d['ms_per_step'], d['final_loss'])"; }
for b in ${BATCHES:-1024 100}; do for i in 1 2; do
DTF_KERNELS_LIB=$GRAFT_REPO_ROOT/diag_build/libdtf_kernels_prev.so timeout -k 10 120 bash -c "$(declare -f one); one prev $b" || exit 1
timeout -k 10 120 bash -c "$(declare -f one); one new $b" || exit 1
done; done
