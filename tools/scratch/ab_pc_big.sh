# conv2 wgrad at 8192 / 16384 images: the self-staging 16 ci x 64 co tile with 128 slabs (the default until round 3) vs the producer/consumer kernel (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
for b in 8192 16384; do
  DTF_C2W_NT=4 DTF_C2_G=128 timeout -k 10 120 python bench.py --per_gpu_batch $b --steps 300 --warmup 30 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b self-staging nt 4, G 128', d['ms_per_step'])" || exit 1
  for g in 64 128; do
    DTF_C2_G=$g timeout -k 10 120 python bench.py --per_gpu_batch $b --steps 300 --warmup 30 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b pc kernel G=$g', d['ms_per_step'])" || exit 1
  done
done
