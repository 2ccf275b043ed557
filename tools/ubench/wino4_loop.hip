// Skeleton of the F(4x4,3x3) Winograd k-step on v_mfma_f32_16x16x4_f32: 72 MFMAs (2 co-blocks x 36
// positions, 64 accumulators pinned to AGPRs + 8 to VGPRs through inline-asm constraints), the 6x6
// input transform as one VALU batch (144 ops), 18 ds_read_b128 of transformed weights and the
// patch reads (6 x [b32, b128, b32]).  Measures cycles per k-step for one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA_A(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define MFMA_V(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

__device__ __forceinline__ void bt6(const float d0, const float d1, const float d2, const float d3, const float d4,
                                    const float d5, float& t0, float& t1, float& t2, float& t3, float& t4, float& t5) {
  t0 = __builtin_fmaf(4.0f, d0, __builtin_fmaf(-5.0f, d2, d4));
  const float a = __builtin_fmaf(-4.0f, d2, d4), b = __builtin_fmaf(-4.0f, d1, d3);
  t1 = a + b;
  t2 = a - b;
  const float c = d4 - d2, e = d3 - d1;
  t3 = __builtin_fmaf(2.0f, e, c);
  t4 = __builtin_fmaf(-2.0f, e, c);
  t5 = __builtin_fmaf(4.0f, d1, __builtin_fmaf(-5.0f, d3, d5));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 12288; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 accA[64], accV[8];
#pragma unroll
  for (int j = 0; j < 64; ++j) accA[j] = (f32x4){0};
#pragma unroll
  for (int j = 0; j < 8; ++j) accV[j] = (f32x4){0};
  float V[36];
#pragma unroll
  for (int j = 0; j < 36; ++j) V[j] = 0.01f * j + lane;
  const float* ub = lds + (lane & 15) * 36 + (lane >> 4) * 1152;         // [ci][co 32][36]
  const float* pb = lds + 5120 + (lane >> 4) * 448 + (lane & 15) * 4;    // plane 448, row 72
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    float raw[36];
    if (MODE & 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const float* p = pb + r * 72 + (it & 1);
        raw[6 * r] = p[3];
        const f32x4 m = *reinterpret_cast<const f32x4*>(p + 4 - (it & 1));
        raw[6 * r + 1] = m[0]; raw[6 * r + 2] = m[1]; raw[6 * r + 3] = m[2]; raw[6 * r + 4] = m[3];
        raw[6 * r + 5] = p[8];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 36; ++j) raw[j] = V[j] * 0.5f;
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    f32x4 u[18];
    constexpr int AHEAD = AHEAD_N;
    auto ldu = [&](int g) {
      if (MODE & 2) u[g] = *reinterpret_cast<const f32x4*>(ub + (g / 9) * 576 + (g % 9) * 4 + (it & 1) * 4);
      else u[g] = (f32x4){V[0], V[1], V[2], V[3]};
    };
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) ldu(g);
#pragma unroll
    for (int g = 0; g < 18; ++g) {  // group g: co-block g / 9, positions 4 (g % 9) .. +3
      if (g + AHEAD < 18) ldu(g + AHEAD);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = 4 * (g % 9) + q, a = (g / 9) * 36 + p;
        if (a < 64) MFMA_A(accA[a], u[g][q], V[p]);
        else MFMA_V(accV[a - 64], u[g][q], V[p]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 4) {
      float t[36];
#pragma unroll
      for (int r = 0; r < 6; ++r)
        bt6(raw[6 * r], raw[6 * r + 1], raw[6 * r + 2], raw[6 * r + 3], raw[6 * r + 4], raw[6 * r + 5],
            t[6 * r], t[6 * r + 1], t[6 * r + 2], t[6 * r + 3], t[6 * r + 4], t[6 * r + 5]);
#pragma unroll
      for (int c = 0; c < 6; ++c)
        bt6(t[c], t[6 + c], t[12 + c], t[18 + c], t[24 + c], t[30 + c],
            V[c], V[6 + c], V[12 + c], V[18 + c], V[24 + c], V[30 + c]);
    } else {
#pragma unroll
      for (int j = 0; j < 36; ++j) V[j] = raw[j];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = clock64();
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) s += accA[j][0] + accA[j][1] + accA[j][2] + accA[j][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += accV[j][0] + accV[j][1] + accV[j][2] + accV[j][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(float* out, long long* cyc) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 65536, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 65536, 0, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("MODE %d (1 patch reads, 2 weight reads, 4 transform): %.0f cycles per k-step (72 MFMAs = 2304 min) | tick %.3f ns | err %s\n",
         MODE, (double)c / iters, ms * 1e6 / (double)c, hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 8);
  run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<6>(out, cyc); run<7>(out, cyc);
  return 0;
}
