// Does a second wave on the SIMD hide what the f32 MFMA (v_mfma_f32_16x16x4_f32) cannot overlap within ONE wave?
// wino4_conv_kernel runs one wave per SIMD (288 accumulator registers): its VALU transform, LDS reads and DMA issue all
// sit in the MFMA's issue stream (k-step 2950 clocks for 1920 of MFMA).  Cases, 256 workgroups:
//   A  1 wave / SIMD : per iteration [60 MFMA] [114 VALU] [31 ds_read_b128]            (today's k-step, MH = 3)
//   B  2 waves / SIMD: per iteration [30 MFMA] [114 VALU] [24 ds_read_b128]  each      (16 co x 16 tiles per wave, both
//                      waves transform the same patch: twice the VALU / patch reads per SIMD, same MFMA work)
//   C  2 waves / SIMD: waves 0-3 [60 MFMA] only, waves 4-7 [114 VALU + 31 reads] only  (pure overlap test)
//   D  1 wave / SIMD : [60 MFMA] only                                                   (floor)
// Reported: clocks per iteration of wave 0 (s_memtime) and of the whole launch (events x clock).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NL, int ROLE, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  f32x4 acc[NM > 0 ? NM : 1];
#pragma unroll
  for (int j = 0; j < (NM > 0 ? NM : 1); ++j) acc[j] = (f32x4){0};
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float v[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) v[q] = a + q;
  for (int i = threadIdx.x; i < 8192; i += NT) lds[i] = a;
  __syncthreads();
  f32x4 ld = (f32x4){0};
  const bool do_m = ROLE == 0 || (ROLE == 1 && wave < 4), do_v = ROLE == 0 || (ROLE == 1 && wave >= 4);
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (do_m) {
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (do_v) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        v[q % 12] = __builtin_fmaf(v[q % 12], b, v[(q + 5) % 12]);
        asm volatile("" : "+v"(v[q % 12]));
      }
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        f32x4 t = *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x * 4 + q * 1024 + i * 4) & 8191)]);
        asm volatile("" : "+v"(t));
        ld += t;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = ld[0] + ld[1] + ld[2] + ld[3];
#pragma unroll
  for (int j = 0; j < (NM > 0 ? NM : 1); ++j)
    for (int r = 0; r < 4; ++r) s += acc[j][r];
  for (int q = 0; q < 12; ++q) s += v[q];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[3 * wave] = t1 - t0; cyc[3 * wave + 1] = w1 - w0; }
}

template <int NM, int NV, int NL, int ROLE, int NT>
void run(const char* name, float* out, unsigned long long* cyc, double mfma_per_simd_iter) {
  const int iters = 4000;
  float best = 1e9f;
  unsigned long long h[24];
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NM, NV, NL, ROLE, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 1 && ms < best) { best = ms; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost); }
  }
  const double mhz = (double)h[0] / (double)h[1] * 100.0;
  const double clk_iter = best * 1e-3 * mhz * 1e6 / iters;
  printf("%-70s wave0 %7.0f", name, (double)h[0] / iters);
  if (NT == 512) printf("  wave4 %7.0f", (double)h[12] / iters);
  else printf("               ");
  printf("  launch %7.0f clocks/iter  (%4.0f MHz)  MFMA pipe %4.1f %%\n", clk_iter, mhz, 100.0 * mfma_per_simd_iter * 32.0 / clk_iter);
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 24 * 8);
  run<60, 0, 0, 0, 256>("D  1 wave/SIMD: 60 MFMA", out, cyc, 60);
  run<60, 114, 31, 0, 256>("A  1 wave/SIMD: 60 MFMA + 114 VALU + 31 ds_read_b128 (today)", out, cyc, 60);
  run<60, 114, 0, 0, 256>("   1 wave/SIMD: 60 MFMA + 114 VALU", out, cyc, 60);
  run<60, 0, 31, 0, 256>("   1 wave/SIMD: 60 MFMA + 31 ds_read_b128", out, cyc, 60);
  run<30, 0, 0, 0, 512>("   2 waves/SIMD: 30 MFMA each", out, cyc, 60);
  run<30, 114, 24, 0, 512>("B  2 waves/SIMD: 30 MFMA + 114 VALU + 24 ds_read_b128 each", out, cyc, 60);
  run<30, 57, 24, 0, 512>("   2 waves/SIMD: 30 MFMA + 57 VALU + 24 ds_read_b128 each (shared V)", out, cyc, 60);
  run<30, 114, 0, 0, 512>("   2 waves/SIMD: 30 MFMA + 114 VALU each", out, cyc, 60);
  run<60, 114, 31, 1, 512>("C  2 waves/SIMD: waves 0-3 60 MFMA | waves 4-7 114 VALU + 31 reads", out, cyc, 60);
  run<60, 228, 31, 1, 512>("   2 waves/SIMD: waves 0-3 60 MFMA | waves 4-7 228 VALU + 31 reads", out, cyc, 60);
  run<60, 456, 0, 1, 512>("   2 waves/SIMD: waves 0-3 60 MFMA | waves 4-7 456 VALU", out, cyc, 60);
  return 0;
}
