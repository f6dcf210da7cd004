# order of the two conv backward kernels in the data-parallel step, by shard size (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; shift; "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run "fused                      " timeout -k 10 120 python bench.py --steps 1000 --warmup 50 --no_fp32 || exit 1
for w in 2 4 8; do for o in 0 1; do
DTF_DGRAD_FIRST=$o DTF_DP_FAKE_WORLD=$w run "shard of $w, dgrad first = $o " timeout -k 10 120 python bench.py --force_dp --no_collapse --steps 1000 --warmup 50 --no_fp32 || exit 1
done; done
