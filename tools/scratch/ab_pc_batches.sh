# A/B of the two conv2 wgrad kernels over batch sizes (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
for b in 100 256 512 1000 2048; do
  for pc in 0 1; do
    DTF_C2W_PC=$pc timeout -k 10 120 python bench.py --per_gpu_batch $b --steps 1000 --warmup 50 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b pc=$pc', d['ms_per_step'], d['final_loss'])" || exit 1
  done
done
