# fp32 rows of the tables: batch 100 / 1024 / 4096 and the widened model   (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
one() { timeout -k 10 200 python bench.py --dtype fp32 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], 'ms', round(d['value']), 'img/s', d.get('model_tflops'), 'TF/s loss', d['final_loss'])" || exit 1; }
one --per_gpu_batch 100 --steps 1000 --warmup 50
one --per_gpu_batch 1024 --steps 500 --warmup 50
one --per_gpu_batch 4096 --steps 100 --warmup 10
one --model wide --per_gpu_batch 1024 --steps 30 --warmup 5
one --optimizer adam --per_gpu_batch 1024 --steps 300 --warmup 30
