// Optimiser steps for gfx950 over flat fp32 arenas: torch.optim.Adam as
// configured at main_train.py:175-176 (coupled L2 weight decay, no amsgrad) and
// torch.optim.SGD(lr) (main_train.py:272).  HBM-bound: 4 reads + 3 writes per
// parameter, float4 wide.
#include "air_common.h"

namespace {

struct AdamK {
  float lr_bc1;   // lr / (1 - beta1^t)
  float isq_bc2;  // 1 / sqrt(1 - beta2^t)
  float beta1, beta2, eps, wd, gscale;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamK& k) {
  g = g * k.gscale + k.wd * p;
  m = k.beta1 * m + (1.0f - k.beta1) * g;
  v = k.beta2 * v + (1.0f - k.beta2) * g * g;
  const float denom = sqrtf(v) * k.isq_bc2 + k.eps;
  p = p - k.lr_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p,
                                                   const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   size_t n, AdamK k) {
  const size_t n4 = n / 4;
  const bool aligned = (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (aligned) {
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (size_t i = tid; i < n4; i += stride) {
      float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
      adam1(pp.x, gg.x, mm.x, vv.x, k);
      adam1(pp.y, gg.y, mm.y, vv.y, k);
      adam1(pp.z, gg.z, mm.z, vv.z, k);
      adam1(pp.w, gg.w, mm.w, vv.w, k);
      p4[i] = pp;
      m4[i] = mm;
      v4[i] = vv;
    }
    for (size_t i = n4 * 4 + tid; i < n; i += stride) adam1(p[i], g[i], m[i], v[i], k);
  } else {
    for (size_t i = tid; i < n; i += stride) adam1(p[i], g[i], m[i], v[i], k);
  }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, size_t n, float lr,
                           float gscale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = p[i] - lr * (g[i] * gscale);
}

}  // namespace

extern "C" {

int air_adam_step(float* p, const float* g, float* m, float* v, size_t n, int step, float lr,
                  float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  air_stream_t stream) {
  if (!p || !g || !m || !v || step < 1) return AIR_EINVAL;
  if (n == 0) return AIR_OK;
  AdamK k;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  k.lr_bc1 = (float)((double)lr / bc1);
  k.isq_bc2 = (float)(1.0 / sqrt(bc2));
  k.beta1 = beta1;
  k.beta2 = beta2;
  k.eps = eps;
  k.wd = weight_decay;
  k.gscale = grad_scale;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, air_stream(stream), p, g, m,
                     v, n, k);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_sgd_step(float* p, const float* g, size_t n, float lr, float grad_scale,
                 air_stream_t stream) {
  if (!p || !g) return AIR_EINVAL;
  if (n == 0) return AIR_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, air_stream(stream), p, g, n,
                     lr, grad_scale);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

}  // extern "C"
