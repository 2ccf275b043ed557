// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds t[e] = e; lane l supplies the address of elements 4 l .. 4 l + 3;
// prints what every lane receives.  hipcc --offload-arch=gfx950 -O2 -o tr_read tr_read.hip && ./tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) unsigned short t[1024];
  for (int e = threadIdx.x; e < 1024; e += 64) t[e] = (unsigned short)e;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(t + threadIdx.x * 4));
  out[threadIdx.x] = __builtin_bit_cast(unsigned long long, v);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned long long h[64]; hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" %4u", (unsigned)((h[l] >> (16 * e)) & 0xffff));
    printf("\n");
  }
  return 0;
}
