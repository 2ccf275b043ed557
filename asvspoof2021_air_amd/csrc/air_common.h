// Shared helpers for the gfx950 kernels.  wave = 64 lanes, always.
#pragma once
#include <hip/hip_runtime.h>

#include "air_hip.h"

#define AIR_WAVE 64

#define AIR_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return AIR_ELAUNCH;     \
  } while (0)

static inline hipStream_t air_stream(air_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Compute units a launch on `st` can occupy: the stream's CU mask (hipExtStreamCreateWithCUMask; for a stream
// without one HIP reports the process-wide ROC_GLOBAL_CU_MASK or all CUs) capped by the device's CU count -
// what the persistent kernels size their grids by instead of assuming the 256 CUs of an unpartitioned MI355X
// (CPX / partitioned modes, masked streams).  api.hip; one query per (device, stream), cached.
int air_stream_cus(hipStream_t st);

__device__ __forceinline__ float air_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double air_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// s[q] = sum of p[k * stride + q], k = 0 .. n - 1, in index order - the bits a serial loop returns - by one wave:
// lane k loads the NQ values of element k (one round trip for the wave instead of n dependent ones in a thread),
// the fold walks the lanes with v_readlane.  n and p are wave-uniform; every lane holds the totals.  The BatchNorm
// finalize kernels are a single workgroup between two full-chip passes: their duration is this latency chain.
template <int NQ>
__device__ __forceinline__ void air_wave_ordered_sums_d(const double* __restrict__ p, int n, int stride, double (&s)[NQ]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = 0.0;
  for (int k0 = 0; k0 < n; k0 += 64) {
    int lo[NQ], hi[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const double v = (k0 + lane < n) ? p[(size_t)(k0 + lane) * stride + q] : 0.0;
      lo[q] = __double2loint(v);
      hi[q] = __double2hiint(v);
    }
    const int m = n - k0 < 64 ? n - k0 : 64;
    for (int k = 0; k < m; ++k) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        s[q] += __hiloint2double(__builtin_amdgcn_readlane(hi[q], k), __builtin_amdgcn_readlane(lo[q], k));
    }
  }
}
__device__ __forceinline__ float air_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Orders this wave's LDS traffic (s_waitcnt lgkmcnt(0)) and stops the compiler
// from moving LDS accesses across the point.  Used for wave-private LDS
// exchanges where a workgroup barrier would be wrong (waves run different trip counts).
__device__ __forceinline__ void air_wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
