// LDS-DMA (global_load_lds) helpers and the XCD-aware block remap shared by the conv kernels.
#pragma once
#include <hip/hip_runtime.h>

// XCD-aware logical block index: hardware places block b on XCD b % 8; give each
// XCD one contiguous range of logical tiles so neighbours share L2 (bijective).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA (global_load_lds) issued as inline asm.  Through the builtin, hipcc's wait-count
// pass puts "s_waitcnt vmcnt(0)" in front of EVERY later ds_read (it cannot tell which LDS
// bytes an in-flight DMA writes), which serialises the next chunk's staging with this chunk's
// MFMAs.  Hand-issued, the DMA stays in flight across the compute and is drained once, by
// dma_wait() in front of the buffer-swap barrier.  LDS destination = M0 + lane * size.
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(__UINTPTR_TYPE__)(lptr_t)p;
}
// l = LDS BYTE address of lane 0's destination (wave-uniform; lds_addr(array) + offsets)
__device__ __forceinline__ void dma4(const float* g, unsigned l) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
               :: "v"(g), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ void dma16(const float* g, unsigned l) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"(g), "s"(l) : "memory", "m0");
}
// the builtin form (compiler-managed waits): measured faster in the wgrad kernel, whose single
// resident wave per SIMD is bound by instruction issue, not by DMA latency
__device__ __forceinline__ void dma4_auto(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 4, 0, 0);
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

