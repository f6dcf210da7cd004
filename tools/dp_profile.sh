# rocprofv3 kernel stats of the data-parallel step path on one GPU (bench.py --force_dp: process group of one rank,
# tail message through the one-shot peer-to-peer kernel).  usage (GPU box): bash tools/dp_profile.sh
set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/dpprof -- python3 $REPO/bench.py --force_dp --steps 200 --warmup 20 > $REPO/gpurun_out/dpprof.log 2>&1 || { tail -5 $REPO/gpurun_out/dpprof.log; exit 1; }
cd $REPO; python3 tools/prof_summary.py gpurun_out/dpprof 220
