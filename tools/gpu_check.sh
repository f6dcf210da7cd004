#!/bin/bash
# Run on the GPU box via gpurun: GPU tests, bench, rocprofv3 kernel stats.  Usage: tools/gpu_check.sh <tag> [pytest-k-expr]
set -o pipefail
TAG=${1:-run}
KEXPR=${2:-}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then KARG=(-k "$KEXPR"); else KARG=(); fi
timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread -x "${KARG[@]}" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -15 gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench.log 2>&1 || { tail -20 gpurun_out/${TAG}_bench.log; exit 1; }
tail -1 gpurun_out/${TAG}_bench.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 50 --warmup 5 --no_graph --single_stream > gpurun_out/${TAG}_prof.log 2>&1 || { tail -20 gpurun_out/${TAG}_prof.log; exit 1; }
python3 tools/prof_summary.py gpurun_out/${TAG}_prof 55
