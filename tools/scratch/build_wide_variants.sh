# variants of the widened layer's implicit-GEMM tile configurations: full libraries in diag_build/ (CPU box)
# usage: build_wide_variants.sh name "cfg-define" [name "cfg-define" ...]
set -e
cd /root/repo
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I csrc/kernels"
OBJS=$(ls build/*.o | grep -v conv2.o)
build() { # name, extra defines
  /opt/rocm/bin/hipcc $FL "${@:2}" -c csrc/kernels/conv2.hip -o /tmp/conv2_$1.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diag_build/libdtf_wide_$1.so $OBJS /tmp/conv2_$1.o && echo built $1
}
n=0
while [ $# -ge 2 ]; do
  build "$1" "$2" &
  shift 2; n=$((n+1))
  if [ $((n % 3)) -eq 0 ]; then wait; fi
done
wait
