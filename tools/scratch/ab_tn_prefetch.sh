# fc1 wgrad + SGD: master weights requested at the kernel's start (in-tree library) vs behind the K loop (DTF_KERNELS_LIB = the previous build); run via gpurun
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -x --timeout 120 --timeout-method thread -k "tn or sgd or engine or step" 2>&1 | tail -3 || exit 1
for b in 100 256 1024; do for i in 1 2; do
DTF_KERNELS_LIB=$GRAFT_REPO_ROOT/diag_build/libdtf_kernels_prev.so timeout -k 10 120 python bench.py --per_gpu_batch $b --steps 1000 --warmup 50 --no_fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b prev', d['ms_per_step'], d['final_loss'])" || exit 1
timeout -k 10 120 python bench.py --per_gpu_batch $b --steps 1000 --warmup 50 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b new ', d['ms_per_step'], d['final_loss'])" || exit 1
done; done
