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
