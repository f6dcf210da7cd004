set -o pipefail
cd $GRAFT_REPO_ROOT
one() { python bench.py --dtype fp32 --per_gpu_batch 1024 --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $1', d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do for g in 0 768 512 256; do
DTF_C1F_GRID=$g timeout -k 10 120 bash -c "$(declare -f one); one $g" || exit 1
done; done
