# local cost of the N > 1 step on one rank: fused step vs --force_dp (collapsed / N > 1 kernels kept / shard of an 8-rank job); run via gpurun
set -o pipefail
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; shift; "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('step_ms_by_path'))"; }
for i in 1 2; do
run "fused               " timeout -k 10 120 python bench.py --steps 1000 --warmup 50 --no_fp32 || exit 1
run "force_dp            " timeout -k 10 120 python bench.py --force_dp --steps 1000 --warmup 50 --no_fp32 || exit 1
run "force_dp no_collapse" timeout -k 10 120 python bench.py --force_dp --no_collapse --steps 1000 --warmup 50 --no_fp32 || exit 1
DTF_DP_FAKE_WORLD=8 run "  + shard of 8      " timeout -k 10 120 python bench.py --force_dp --no_collapse --steps 1000 --warmup 50 --no_fp32 || exit 1
done
