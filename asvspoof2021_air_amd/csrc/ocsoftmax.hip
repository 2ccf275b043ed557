// OC-Softmax (``ang_iso``) head for gfx950: AngularIsoLoss.forward ==
// OCSoftmax.forward (loss.py:73-97, :187-206) and its backward, one launch each.
//   w = c/|c|, xh = x/|x| (F.normalize, eps 1e-12), s = <xh, w>
//   m = r_real - s (label 0) | s - r_fake (label 1); loss = mean softplus(alpha m)
//   returns (loss, -s)
#include "air_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXB = 4096;  // rows handled by the single-workgroup kernels

__device__ __forceinline__ float softplus20(float z) {  // nn.Softplus(beta=1, threshold=20)
  return z > 20.0f ? z : log1pf(expf(z));
}

// one workgroup; a wave per row
__global__ __launch_bounds__(NT) void ocs_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ center,
                                                     const int64_t* __restrict__ labels, int B,
                                                     int D, float r_real, float r_fake, float alpha,
                                                     float* __restrict__ loss,
                                                     float* __restrict__ neg_scores) {
  __shared__ float sh[NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float cn = 0.0f;
  for (int d = lane; d < D; d += 64) cn = fmaf(center[d], center[d], cn);
  cn = fmaxf(sqrtf(air_wave_sum(cn)), 1e-12f);
  float acc = 0.0f;
  for (int b = wave; b < B; b += NT / 64) {
    const float* __restrict__ xr = x + (size_t)b * D;
    float xx = 0.0f, xc = 0.0f;
    for (int d = lane; d < D; d += 64) {
      const float v = xr[d];
      xx = fmaf(v, v, xx);
      xc = fmaf(v, center[d], xc);
    }
    xx = air_wave_sum(xx);
    xc = air_wave_sum(xc);
    const float xn = fmaxf(sqrtf(xx), 1e-12f);
    const float s = xc / (xn * cn);
    const float m = labels[b] == 0 ? r_real - s : s - r_fake;
    if (lane == 0) {
      neg_scores[b] = -s;
      acc += softplus20(alpha * m);
    }
  }
  if (lane == 0) sh[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int w = 0; w < NT / 64; ++w) t += sh[w];
    loss[0] = t / (float)B;
  }
}

// one workgroup: phase 1 per-row scalars into LDS, phase 2 thread-per-dimension
__global__ __launch_bounds__(NT) void ocs_bwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ center,
                                                     const int64_t* __restrict__ labels, int B,
                                                     int D, float r_real, float r_fake, float alpha,
                                                     const float* __restrict__ gscale,
                                                     float* __restrict__ dx,
                                                     float* __restrict__ dcenter) {
  __shared__ float s_s[MAXB], s_coef[MAXB], s_inv[MAXB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float cn = 0.0f;
  for (int d = lane; d < D; d += 64) cn = fmaf(center[d], center[d], cn);
  cn = fmaxf(sqrtf(air_wave_sum(cn)), 1e-12f);
  const float g0 = gscale ? gscale[0] : 1.0f;
  for (int b = wave; b < B; b += NT / 64) {
    const float* __restrict__ xr = x + (size_t)b * D;
    float xx = 0.0f, xc = 0.0f;
    for (int d = lane; d < D; d += 64) {
      const float v = xr[d];
      xx = fmaf(v, v, xx);
      xc = fmaf(v, center[d], xc);
    }
    xx = air_wave_sum(xx);
    xc = air_wave_sum(xc);
    const float xn = fmaxf(sqrtf(xx), 1e-12f);
    const float s = xc / (xn * cn);
    const bool real = labels[b] == 0;
    const float z = alpha * (real ? r_real - s : s - r_fake);
    const float sig = z > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-z));
    if (lane == 0) {
      s_s[b] = s;
      s_coef[b] = g0 * sig * alpha * (real ? -1.0f : 1.0f) / (float)B;  // dL/ds
      s_inv[b] = 1.0f / xn;
    }
  }
  __syncthreads();
  const float icn = 1.0f / cn;
  for (int d = threadIdx.x; d < D; d += NT) {
    const float wh = center[d] * icn;
    float gc = 0.0f;
    for (int b = 0; b < B; ++b) {
      const float xh = x[(size_t)b * D + d] * s_inv[b];
      const float k = s_coef[b], s = s_s[b];
      dx[(size_t)b * D + d] = k * (wh - s * xh) * s_inv[b];
      gc = fmaf(k, xh - s * wh, gc);
    }
    dcenter[d] = gc * icn;
  }
}

}  // namespace

extern "C" {

int air_ocsoftmax_fwd(const float* x, const float* center, const int64_t* labels, int B, int D,
                      float r_real, float r_fake, float alpha, float* loss, float* neg_scores,
                      air_stream_t stream) {
  if (!x || !center || !labels || !loss || !neg_scores || B <= 0 || D <= 0) return AIR_EINVAL;
  hipLaunchKernelGGL(ocs_fwd_kernel, dim3(1), dim3(NT), 0, air_stream(stream), x, center, labels,
                     B, D, r_real, r_fake, alpha, loss, neg_scores);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_ocsoftmax_bwd(const float* x, const float* center, const int64_t* labels, int B, int D,
                      float r_real, float r_fake, float alpha, const float* gscale_dev, float* dx,
                      float* dcenter, air_stream_t stream) {
  if (!x || !center || !labels || !dx || !dcenter || B <= 0 || D <= 0) return AIR_EINVAL;
  if (B > MAXB) return AIR_EUNSUPPORTED;
  hipLaunchKernelGGL(ocs_bwd_kernel, dim3(1), dim3(NT), 0, air_stream(stream), x, center, labels,
                     B, D, r_real, r_fake, alpha, gscale_dev, dx, dcenter);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

}  // extern "C"
