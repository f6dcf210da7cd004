# control of the library A/B method: the same build through DTF_KERNELS_LIB and in-tree, against the previous build; run via gpurun
set -o pipefail
cd $GRAFT_REPO_ROOT
one() { python bench.py --per_gpu_batch ${B:-1024} --steps 1000 --warmup 50 --no_fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['final_loss'])"; }
for B in 1024 100; do export B; for i in 1 2; do
DTF_KERNELS_LIB=$GRAFT_REPO_ROOT/diag_build/libdtf_kernels_prev.so timeout -k 10 120 bash -c "$(declare -f one); one prev_via_env" || exit 1
DTF_KERNELS_LIB=$GRAFT_REPO_ROOT/diag_build/libdtf_kernels_new.so timeout -k 10 120 bash -c "$(declare -f one); one new_via_env" || exit 1
timeout -k 10 120 bash -c "$(declare -f one); one new_in_tree" || exit 1
done; done
