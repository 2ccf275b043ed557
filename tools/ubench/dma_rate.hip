// Microbenchmark: throughput of LDS-DMA (buffer_load ... lds) by width and alignment, and of plain
// global_load_dwordx4, from an L2-resident source.  4 waves per CU, 256 CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 rsrc(const void* b, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)b >> 32));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

// MODE 0: dword DMA, 1: dwordx4 DMA aligned, 2: dwordx4 DMA misaligned by 4 B, 3: global_load_dwordx4 to VGPR
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* src, float* out, int iters, unsigned span) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  i32x4 r = rsrc(src, span);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
  unsigned voff = (blockIdx.x * 4096u + tid * (MODE == 0 ? 4u : 16u) + (MODE == 2 ? 4u : 0u)) % (span - 65536u);
  f32x4 accv = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    unsigned so = __builtin_amdgcn_readfirstlane((i & 7) * 4096);
    unsigned m0 = __builtin_amdgcn_readfirstlane(lds0 + (i & 3) * 8192 + wave * (MODE == 0 ? 256 : 1024));
    if (MODE == 0)
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %3, %0, %2 offen lds\n\t"
                   "buffer_load_dword %3, %0, %2 offen offset:1024 lds\n\t"
                   "buffer_load_dword %3, %0, %2 offen offset:2048 lds\n\t"
                   "buffer_load_dword %3, %0, %2 offen offset:3072 lds" :: "s"(r), "s"(m0), "s"(so), "v"(voff) : "memory", "m0");
    else if (MODE == 1 || MODE == 2)
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %0, %2 offen lds\n\t"
                   "s_add_u32 m0, m0, 4096\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %3, %0, %2 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %3, %0, %2 offen offset:2048 lds\n\t"
                   "buffer_load_dwordx4 %3, %0, %2 offen offset:3072 lds" :: "s"(r), "s"(m0), "s"(so), "v"(voff) : "memory", "m0");
    else {
      f32x4 a, b, c, d;
      asm volatile("buffer_load_dwordx4 %0, %5, %4, %6 offen\n\t"
                   "buffer_load_dwordx4 %1, %5, %4, %6 offen offset:1024\n\t"
                   "buffer_load_dwordx4 %2, %5, %4, %6 offen offset:2048\n\t"
                   "buffer_load_dwordx4 %3, %5, %4, %6 offen offset:3072\n\t"
                   "s_waitcnt vmcnt(8)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "s"(r), "v"(voff), "s"(so) : "memory");
      accv += a + b + c + d;
    }
    if ((i & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * 256 + tid] = lds[tid] + accv[0] + accv[1] + accv[2] + accv[3];
}

template <int MODE>
void run(const float* src, float* out, unsigned span, const char* name) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 65536, 0, src, out, iters, span);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 65536, 0, src, out, iters, span);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ninstr = 4.0 * iters;             // per wave
  const double bytes = ninstr * 64 * (MODE == 0 ? 4 : 16) * 4 * 256;  // chip
  printf("%-34s: %.1f ns per wave-instruction per wave | %.2f TB/s chip | %.1f GB/s per CU\n", name,
         ms * 1e6 / ninstr, bytes / (ms * 1e-3) / 1e12, bytes / 256 / (ms * 1e-3) / 1e9);
}

int main() {
  const unsigned span = 64u << 20;  // 64 MB source: L2/MALL resident after the warm-up
  float *src, *out;
  hipMalloc(&src, span); hipMalloc(&out, 256 * 256 * 4);
  hipMemset(src, 0, span);
  run<0>(src, out, span, "LDS-DMA dword");
  run<1>(src, out, span, "LDS-DMA dwordx4 aligned");
  run<2>(src, out, span, "LDS-DMA dwordx4 4-byte aligned");
  run<3>(src, out, span, "global_load_dwordx4 to VGPR");
  return 0;
}
