# fp32 step, one library, an environment switch off / on: ab_f32_env.sh VAR OFF ON [batches...]   (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
VAR=$1; OFF=$2; ON=$3; shift 3
one() { python bench.py --dtype fp32 --per_gpu_batch $2 --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $2 $1', d['ms_per_step'], d['final_loss'])"; }
for b in ${@:-1024}; do for i in 1 2; do
env $VAR=$OFF timeout -k 10 120 bash -c "$(declare -f one); one $VAR=$OFF $b" || exit 1
env $VAR=$ON timeout -k 10 120 bash -c "$(declare -f one); one $VAR=$ON $b" || exit 1
done; done
