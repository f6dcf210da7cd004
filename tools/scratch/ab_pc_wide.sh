# widened model (conv 128/256, fc4096): conv2 wgrad default (16 ci x 64 co self-staging tile, 8 slabs) vs the producer/consumer kernel (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_kernels_gpu.py -q -x --timeout 120 --timeout-method thread -k "conv2_backward" 2>&1 | tail -2 || exit 1
run() { python bench.py --model wide --steps 100 --warmup 10 --no_fp32 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['final_loss'])"; }
timeout -k 10 200 bash -c "$(declare -f run); DTF_C2W_NT=4 DTF_C2_G=8 run 'wide nt 4, G 8 (the former default)'" || exit 1
for g in 4 8 16; do
  DTF_C2_G=$g timeout -k 10 200 bash -c "$(declare -f run); run 'wide pc kernel G=$g'" || exit 1
done
