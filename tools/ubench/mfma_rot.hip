// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 with NACC rotating accumulators and NV
// independent VALU instructions issued between consecutive MFMAs.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NV, int NL, int NT = 256>
__global__ __launch_bounds__(NT) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[4096];
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = (f32x16){0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = a + q;
  lds[threadIdx.x] = a;
  __syncthreads();
  float ld = 0.f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        v[q & 7] = v[q & 7] + b;
        asm volatile("" : "+v"(v[q & 7]));
      }
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        float t = lds[(threadIdx.x + q * 64 + i) & 4095];
        asm volatile("" : "+v"(t));
        ld += t;
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float s = ld;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NV, int NL, int NT = 256>
void run(float* out, long long* cyc) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, NV, NL, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, NV, NL, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = 256.0 * (NT / 64) * (double)iters * NACC;
  printf("NT=%4d NACC=%2d NV=%2d NL=%d : wave0 %.1f ticks/MFMA | chip %.1f TFLOP/s | %.1f ns per MFMA per SIMD | tick = %.3f ns\n",
         NT, NACC, NV, NL, (double)c / ((double)iters * NACC), nm * 4096 / (ms * 1e-3) / 1e12,
         ms * 1e6 / (nm / 1024.0), ms * 1e6 / (double)c);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  run<1, 0, 0>(out, cyc); run<2, 0, 0>(out, cyc); run<4, 0, 0>(out, cyc); run<8, 0, 0>(out, cyc); run<16, 0, 0>(out, cyc);
  run<16, 2, 0>(out, cyc); run<16, 4, 0>(out, cyc); run<16, 6, 0>(out, cyc); run<16, 8, 0>(out, cyc); run<16, 12, 0>(out, cyc);
  run<16, 0, 1>(out, cyc); run<16, 4, 1>(out, cyc); run<16, 4, 2>(out, cyc);
  run<2, 4, 0>(out, cyc); run<2, 8, 0>(out, cyc);
  run<8, 0, 0, 512>(out, cyc); run<8, 4, 0, 512>(out, cyc); run<8, 8, 0, 512>(out, cyc); run<8, 12, 0, 512>(out, cyc); run<8, 16, 0, 512>(out, cyc);
  run<4, 0, 0, 1024>(out, cyc); run<4, 8, 0, 1024>(out, cyc); run<4, 16, 0, 1024>(out, cyc); run<4, 24, 0, 1024>(out, cyc);
  return 0;
}
