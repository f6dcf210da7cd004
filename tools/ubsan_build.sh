#!/bin/bash
# Host-side UndefinedBehaviorSanitizer build of the kernel library (launch wrappers + csrc/runtime/engine.hip's executor;
# device code is compiled as usual: the GPU sanitizers are not available on this pool).  CPU box:
#   bash tools/ubsan_build.sh            -> diag_build/libdtf_kernels_ubsan.so
# GPU box:
#   DTF_KERNELS_LIB=diag_build/libdtf_kernels_ubsan.so python -m pytest tests -m gpu ...
# Any report aborts the process (-fno-sanitize-recover), so a green run means no report.
# -fno-sanitize=function: with the function-type check on, kernels launched through a function pointer
# (`auto kern = conv5x5_igemm<...>; hipLaunchKernelGGL(kern, ...)`) return success and run NOTHING on this toolchain
# (tools/scratch/ubsan_probe.py: outputs untouched, hipGetLastError 0) -- an artefact of the check's function prefix and
# HIP's stub registration, not a finding.
set -e -o pipefail
cd "$(dirname "$0")/.."
RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
mkdir -p diag_build/ubsan
for f in csrc/kernels/*.hip csrc/runtime/*.hip; do
  echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -g -std=c++17 -fPIC -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize=function -Xarch_host -fno-sanitize-recover=undefined -I csrc/kernels -c $f -o diag_build/ubsan/$(basename $f .hip).o"
done | xargs -P 8 -I{} sh -c "{}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diag_build/libdtf_kernels_ubsan.so diag_build/ubsan/*.o \
  -Wl,--whole-archive $RT/libclang_rt.ubsan_standalone-x86_64.a $RT/libclang_rt.ubsan_standalone_cxx-x86_64.a -Wl,--no-whole-archive -lpthread -ldl
echo "built diag_build/libdtf_kernels_ubsan.so"
