# conv2 wgrad at 4096 / 8192 images: the default (16 ci x 64 co self-staging tile, 128 slabs) vs the producer/consumer kernel (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
for b in 4096 8192; do
  timeout -k 10 120 python bench.py --per_gpu_batch $b --steps 300 --warmup 30 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b default      ', d['ms_per_step'])" || exit 1
  for g in 64 128; do
    DTF_C2W_NT=0 DTF_C2_G=$g DTF_FORCE_C2W_PC=1 timeout -k 10 120 python bench.py --per_gpu_batch $b --steps 300 --warmup 30 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b pc kernel G=$g', d['ms_per_step'])" || exit 1
  done
done
