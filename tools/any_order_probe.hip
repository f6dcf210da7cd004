// Does hipExtAnyOrderLaunch (no AQL barrier bit) let kernels of ONE stream overlap on this stack?
// Build: hipcc --offload-arch=gfx950 -O2 tools/any_order_probe.hip -o diag_build/any_order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) *sink = 1;
}

static double run(hipStream_t s, int n, unsigned flags, int* sink) {
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) {
    if (flags == 0xffffffffu)
      hipLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, 2000LL, sink);   // 2000 ticks of 100 MHz = 20 us
    else
      hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, nullptr, nullptr, flags, 2000LL, sink);
  }
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  int* sink;
  hipMalloc(&sink, 4);
  for (int rep = 0; rep < 3; ++rep) {
    double a = run(s, 8, 0xffffffffu, sink);
    double b = run(s, 8, 0, sink);
    double c = run(s, 8, hipExtAnyOrderLaunch, sink);
    printf("8 x 20 us kernels, one stream: plain %.1f us | ext flags=0 %.1f us | ext any-order %.1f us\n", a, b, c);
  }
  // ordering check: a normal launch after any-order launches must wait for all of them
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
