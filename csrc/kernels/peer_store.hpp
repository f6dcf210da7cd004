// Peer-to-peer store / poll primitives shared by the exchange fabric (dp.hip) and the parameter-server service
// kernel (ps.hip).  Exchange buffers are UNCACHED pages (hipDeviceMallocUncached, mapped into the peers through hipIpc).
#pragma once
#include "common.hpp"

namespace dtf {

// Stores into a PEER's exchange buffer: write-through past every cache (sc0 sc1), and the "release" that publishes
// them is a wait for their completion -- NOT a system-scope fence.  A fence (buffer_wbl2 sc0 sc1) writes back every
// dirty line of this XCD's L2, i.e. whatever the conv backward kernels running next to the dense exchange have just
// produced: 64 of them per launch made the overlapped conv2 wgrad 13 us slower and the exchange itself 29 us long for
// 5 MB of traffic (measured, rocprofv3 timeline).  The exchange buffers are uncached pages, nothing of them ever sits
// in a cache, so once the write-through stores are acknowledged (vmcnt 0) there is nothing left for a write-back to do.
__device__ __forceinline__ void dp_put16(void* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void dp_put8(void* p, const u32x2& v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void dp_puts_done() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
__device__ __forceinline__ void dp_raise(unsigned* flag, unsigned v) {     // after dp_puts_done() + __syncthreads()
  __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// relaxed system-scope accesses to protocol words in uncached memory
__device__ __forceinline__ unsigned uc_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void uc_store(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// 16-byte poll of uncached protocol words, past every cache (sc0 sc1)
__device__ __forceinline__ u32x4 uc_load4(const unsigned* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

}  // namespace dtf
