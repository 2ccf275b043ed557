// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 with NACC rotating accumulators, NV independent
// VALU instructions and NL LDS reads (ds_read_b128) issued between consecutive MFMAs.
// One wave per SIMD (256 threads per workgroup, one workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NV, int NL, int EVERY, int NT = 256>
__global__ __launch_bounds__(NT) void k(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  f32x4 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = a + q;
  for (int i = threadIdx.x; i < 8192; i += NT) lds[i] = a;
  __syncthreads();
  f32x4 ld = (f32x4){0};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      if (j % EVERY == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          v[q & 7] = v[q & 7] + b;
          asm volatile("" : "+v"(v[q & 7]));
        }
#pragma unroll
        for (int q = 0; q < NL; ++q) {
          f32x4 t = *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x * 4 + q * 1024 + i * 4) & 8191)]);
          asm volatile("" : "+v"(t));
          ld += t;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float s = ld[0] + ld[1] + ld[2] + ld[3];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 4; ++r) s += acc[j][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NV, int NL, int EVERY = 1, int NT = 256>
void run(float* out, long long* cyc) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, NV, NL, EVERY, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, NV, NL, EVERY, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = 256.0 * (NT / 64) * (double)iters * NACC;
  printf("NT=%4d NACC=%2d NV=%2d NL=%d every %d : wave0 %.1f cycles/MFMA | chip %.1f TFLOP/s | tick = %.3f ns\n",
         NT, NACC, NV, NL, EVERY, (double)c / ((double)iters * NACC), nm * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / (double)c);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  run<1, 0, 0>(out, cyc); run<2, 0, 0>(out, cyc); run<4, 0, 0>(out, cyc); run<36, 0, 0>(out, cyc); run<72, 0, 0>(out, cyc);
  run<72, 1, 0>(out, cyc); run<72, 2, 0>(out, cyc); run<72, 3, 0>(out, cyc); run<72, 4, 0>(out, cyc); run<72, 6, 0>(out, cyc);
  run<72, 8, 0, 4>(out, cyc); run<72, 16, 0, 8>(out, cyc);
  run<72, 0, 1, 4>(out, cyc); run<72, 0, 1, 2>(out, cyc); run<72, 2, 1, 4>(out, cyc); run<72, 8, 1, 4>(out, cyc);
  run<36, 0, 0, 1, 512>(out, cyc); run<36, 2, 0, 1, 512>(out, cyc); run<36, 4, 0, 1, 512>(out, cyc);
  return 0;
}
