# rocprofv3 kernel table of one bench.py configuration: tools/scratch/prof_one.sh <tag> [bench args...]   (run via gpurun)
set -o pipefail
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
tag=$1; shift
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/p1_$tag -- python3 $REPO/bench.py "$@" --steps 200 --warmup 20 --prewarm_s 0 --no_fp32 > $REPO/gpurun_out/p1_$tag.log 2>&1 || { tail -5 $REPO/gpurun_out/p1_$tag.log; exit 1; }
cd $REPO && python3 tools/prof_summary.py gpurun_out/p1_$tag 220
