# widened model, one library, an environment switch: ab_wide_env.sh VAR v1 v2 ...   (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
VAR=$1; shift
one() { python bench.py --model wide --steps 60 --warmup 10 --no_fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do
timeout -k 10 120 bash -c "$(declare -f one); one default" || exit 1
for v in "$@"; do
env $VAR=$v timeout -k 10 120 bash -c "$(declare -f one); one $VAR=$v" || exit 1
done; done
