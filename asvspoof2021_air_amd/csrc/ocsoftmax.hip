// OC-Softmax (``ang_iso``) head for gfx950: AngularIsoLoss.forward ==
// OCSoftmax.forward (loss.py:73-97, :187-206) and its backward, one launch each.
//   w = c/|c|, xh = x/|x| (F.normalize, eps 1e-12), s = <xh, w>
//   m = r_real - s (label 0) | s - r_fake (label 1); loss = mean softplus(alpha m)
//   returns (loss, -s)
#include "air_common.h"

namespace {

constexpr int NT = 1024;    // ONE workgroup (the loss is summed in a fixed order), 16 waves
constexpr int NW = NT / 64;
constexpr int MAXB = 4096;  // rows handled by the single-workgroup kernels
constexpr int RU = 4;       // rows a wave has in flight (each row: one exposed load round trip otherwise; B = 128 rows
                            // over 4 waves, one at a time, was 61 us forward / 44 us backward - 1.4 % of the ECAPA step)

__device__ __forceinline__ float softplus20(float z) {  // nn.Softplus(beta=1, threshold=20)
  return z > 20.0f ? z : log1pf(expf(z));
}

// |c| (every wave computes it: no barrier needed before the rows)
__device__ __forceinline__ float center_norm(const float* __restrict__ center, int D, int lane) {
  float cn = 0.0f;
  for (int d = lane; d < D; d += 64) cn = fmaf(center[d], center[d], cn);
  return fmaxf(sqrtf(air_wave_sum(cn)), 1e-12f);
}

// <x_b, x_b> and <x_b, c> of RU rows at once (lane-strided, the loads of all RU rows issued before the first reduce)
__device__ __forceinline__ void row_dots(const float* __restrict__ x, const float* __restrict__ center, int B, int D,
                                         int b0, int lane, float (&xx)[RU], float (&xc)[RU]) {
#pragma unroll
  for (int u = 0; u < RU; ++u) xx[u] = xc[u] = 0.0f;
  for (int d = lane; d < D; d += 64) {
    const float c = center[d];
    float v[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) v[u] = b0 + u * NW < B ? x[(size_t)(b0 + u * NW) * D + d] : 0.0f;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      xx[u] = fmaf(v[u], v[u], xx[u]);
      xc[u] = fmaf(v[u], c, xc[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    xx[u] = air_wave_sum(xx[u]);
    xc[u] = air_wave_sum(xc[u]);
  }
}

// a wave per row, RU rows per wave in flight; per-row softplus terms through LDS, summed by wave 0 in row order
__global__ __launch_bounds__(NT) void ocs_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ center,
                                                     const int64_t* __restrict__ labels, int B,
                                                     int D, float r_real, float r_fake, float alpha,
                                                     float* __restrict__ loss,
                                                     float* __restrict__ neg_scores) {
  __shared__ float s_sp[MAXB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float cn = center_norm(center, D, lane);
  for (int b0 = wave; b0 < B; b0 += NW * RU) {
    float xx[RU], xc[RU];
    row_dots(x, center, B, D, b0, lane, xx, xc);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int b = b0 + u * NW;
      if (b < B && lane == 0) {
        const float xn = fmaxf(sqrtf(xx[u]), 1e-12f);
        const float s = xc[u] / (xn * cn);
        const float m = labels[b] == 0 ? r_real - s : s - r_fake;
        neg_scores[b] = -s;
        s_sp[b] = softplus20(alpha * m);
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    float t = 0.0f;
    for (int b = lane; b < B; b += 64) t += s_sp[b];
    t = air_wave_sum(t);
    if (lane == 0) loss[0] = t / (float)B;
  }
}

// phase 1 per-row scalars into LDS (as the forward), phase 2: dx elementwise over (row, dimension) and dcenter as
// NT / D' row groups per dimension reduced through LDS in a fixed order
__global__ __launch_bounds__(NT) void ocs_bwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ center,
                                                     const int64_t* __restrict__ labels, int B,
                                                     int D, float r_real, float r_fake, float alpha,
                                                     const float* __restrict__ gscale,
                                                     float* __restrict__ dx,
                                                     float* __restrict__ dcenter) {
  __shared__ float s_s[MAXB], s_coef[MAXB], s_inv[MAXB];
  __shared__ float s_part[NT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float cn = center_norm(center, D, lane);
  const float g0 = gscale ? gscale[0] : 1.0f;
  for (int b0 = wave; b0 < B; b0 += NW * RU) {
    float xx[RU], xc[RU];
    row_dots(x, center, B, D, b0, lane, xx, xc);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int b = b0 + u * NW;
      if (b < B && lane == 0) {
        const float xn = fmaxf(sqrtf(xx[u]), 1e-12f);
        const float s = xc[u] / (xn * cn);
        const bool real = labels[b] == 0;
        const float z = alpha * (real ? r_real - s : s - r_fake);
        const float sig = z > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-z));
        s_s[b] = s;
        s_coef[b] = g0 * sig * alpha * (real ? -1.0f : 1.0f) / (float)B;  // dL/ds
        s_inv[b] = 1.0f / xn;
      }
    }
  }
  __syncthreads();
  const float icn = 1.0f / cn;
  // dimension chunk of DC = min(D', NT) columns at a time (D' = D rounded up to a power of two <= NT); the NT / DC row
  // groups of a column each take rows g, g + G, ... and leave a partial dcenter sum
  int DC = 64;
  while (DC < D && DC < NT) DC <<= 1;
  const int G = NT / DC, dl = threadIdx.x % DC, g = threadIdx.x / DC;
  for (int d0 = 0; d0 < D; d0 += DC) {
    const int d = d0 + dl;
    float gc = 0.0f;
    if (d < D) {
      const float wh = center[d] * icn;
      for (int b = g; b < B; b += G * RU) {
        float v[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) v[u] = b + u * G < B ? x[(size_t)(b + u * G) * D + d] : 0.0f;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int bb = b + u * G;
          if (bb < B) {
            const float xh = v[u] * s_inv[bb];
            const float k = s_coef[bb], s = s_s[bb];
            dx[(size_t)bb * D + d] = k * (wh - s * xh) * s_inv[bb];
            gc = fmaf(k, xh - s * wh, gc);
          }
        }
      }
    }
    s_part[threadIdx.x] = gc;
    __syncthreads();
    if (g == 0 && d < D) {
      float t = 0.0f;
      for (int q = 0; q < G; ++q) t += s_part[q * DC + dl];
      dcenter[d] = t * icn;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int air_ocsoftmax_fwd(const float* x, const float* center, const int64_t* labels, int B, int D,
                      float r_real, float r_fake, float alpha, float* loss, float* neg_scores,
                      air_stream_t stream) {
  if (!x || !center || !labels || !loss || !neg_scores || B <= 0 || D <= 0) return AIR_EINVAL;
  if (B > MAXB) return AIR_EUNSUPPORTED;  // (the per-row loss terms go through LDS)
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
