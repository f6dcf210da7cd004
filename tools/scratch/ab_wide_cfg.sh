# widened model: step time with each tile-configuration variant library (run via gpurun)
set -o pipefail
cd $GRAFT_REPO_ROOT
one() { python bench.py --model wide --steps 60 --warmup 10 --no_fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do
timeout -k 10 120 bash -c "$(declare -f one); one base" || exit 1
for v in ${VARIANTS:-d_nw4 d_kc64 d_kc64nw4 f_kc64 f_nw2}; do
DTF_KERNELS_LIB=$GRAFT_REPO_ROOT/diag_build/libdtf_wide_$v.so timeout -k 10 120 bash -c "$(declare -f one); one $v" || echo "$v FAILED"
done; done
