// Semantics check of the inline-asm building blocks used by conv_wino.hip:
// buffer_load_dword ... lds with an out-of-range offset, and v_pk_add_f32 modifiers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* y, int nbytes) {
  __shared__ float lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -7.0f;
  __syncthreads();
  i32x4 r;
  const float* xb = x - 1024;  // biased base
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)xb);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)xb >> 32));
  r[2] = __builtin_amdgcn_readfirstlane(nbytes + 4096);
  r[3] = 0x00020000;
  unsigned voff = (threadIdx.x & 1) ? 0x80000000u - 1024u : 4096u + threadIdx.x * 4 - 1024u;  // odd lanes out of range
  unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
  int soff = __builtin_amdgcn_readfirstlane(16);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "buffer_load_dword %0, %1, %3 offen offset:1024 lds"
               :: "v"(voff), "s"(r), "s"(m0v), "s"(soff) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  y[threadIdx.x] = lds[256 + threadIdx.x];   // inst offset applies to the LDS address too
  f32x2 lo = {1.0f, 2.0f}, hi = {10.0f, 20.0f}, c, d, e;
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(c) : "v"(lo), "v"(hi));
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(lo), "v"(hi));
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e) : "v"(lo), "v"(hi));
  if (threadIdx.x == 0) { y[64] = c[0]; y[65] = c[1]; y[66] = d[0]; y[67] = d[1]; y[68] = e[0]; y[69] = e[1]; }
}
int main() {
  float h[1024], *x, *y, o[128];
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMalloc(&x, 4096 * 4); hipMalloc(&y, 512);
  hipMemcpy(x + 2048, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x + 2048, y, 4096);
  hipMemcpy(o, y, 512, hipMemcpyDeviceToHost);
  printf("lds[256..]: "); for (int i = 0; i < 8; ++i) printf("%g ", o[i]); printf("\n(expect 5 0 7 0 9 0 ...: x[(i*4+16)/4] for even lanes, 0 for out-of-range odd lanes)\n");
  printf("pk: %g %g | %g %g | %g %g (expect -9 12 | 8 -18 | -9 -18)\n", o[64], o[65], o[66], o[67], o[68], o[69]);
  return 0;
}
