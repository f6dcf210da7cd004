set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -q -x --timeout 120 --timeout-method thread -k "conv2_backward" 2>&1 | tail -5 || exit 1
for i in 1 2; do
DTF_C2W_PC=0 timeout -k 10 120 python bench.py --steps 1000 --warmup 50 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old ', d['ms_per_step'], d['final_loss'])" || exit 1
timeout -k 10 120 python bench.py --steps 1000 --warmup 50 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pc  ', d['ms_per_step'], d['final_loss'])" || exit 1
done
