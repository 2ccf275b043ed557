// Small HBM/latency-bound kernels of the ECAPA-TDNN path (ecapa_tdnn.py) for gfx950:
// strided copies/adds between channel groups (the Res2 split / concat as views),
// per-channel sums (bias gradients), per-row statistics over time (SE mean, context
// mean/std), the SE gate, and the attentive-statistics pooling with its softmax over time.
// Tensors are (B, C, T) fp32, time contiguous; one wave per (b, c) row where rows matter.
#include "air_common.h"

namespace {

constexpr int NT = 256;

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef float f2a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// float -> bf16 bits, round to nearest even (v_cvt_pk_bf16_f32: the rounding of conv1d_bf16.hip)
__device__ __forceinline__ unsigned short bf16_bits(float a) {
  const f32x2_t v = {a, 0.0f};
  return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)) & 0xffffu);
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = air_wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < NT / 64; ++w) r += sh[w];
  return r;
}

// out[b][c][s] = a[b][c][s] (+ b2[b][c][s]); each tensor has its own batch stride.  bf (optional): the result also
// as bf16 at bf[b * bf_bs + c * Tp + s] (the weight-gradient GEMM's operand layout, conv1d_bf16.hip).
__global__ __launch_bounds__(NT) void add_strided_kernel(float* __restrict__ out, size_t ob,
                                                         const float* __restrict__ a, size_t ab,
                                                         const float* __restrict__ b2, size_t bb,
                                                         int CS, int S, unsigned short* __restrict__ bf, size_t bf_bs,
                                                         int Tp) {
  const int b = blockIdx.y;
  const float* __restrict__ pa = a + (size_t)b * ab;
  const float* __restrict__ pb = b2 ? b2 + (size_t)b * bb : nullptr;
  float* __restrict__ po = out + (size_t)b * ob;
  unsigned short* __restrict__ pf = bf ? bf + (size_t)b * bf_bs : nullptr;
  for (int i = blockIdx.x * NT + threadIdx.x; i < CS; i += gridDim.x * NT) {
    const float v = pb ? pa[i] + pb[i] : pa[i];
    po[i] = v;
    if (pf) {
      const int c = i / S;
      pf[(size_t)c * Tp + (i - c * S)] = bf16_bits(v);
    }
  }
}

// Res2 chain step (ecapa_tdnn.py:78-83): v = x * scale[c] + shift[c] (BatchNorm apply of branch i);
// y1 = v goes straight into its channel slice of the concat tensor, y2 = v + add is the input of
// branch i + 1 ("sp + spx[i + 1]").  One pass instead of bn_apply + two strided copies.
__global__ __launch_bounds__(NT) void res2_bn_apply_kernel(const float* __restrict__ x, int C, int S,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           float* __restrict__ y1, size_t y1_bs,
                                                           const float* __restrict__ add, size_t add_bs,
                                                           float* __restrict__ y2, unsigned short* __restrict__ bf,
                                                           size_t bf_bs, int Tp) {
  const int plane = blockIdx.x;  // b * C + c
  const int b = plane / C, c = plane - b * C;
  const float sc = scale[c], sh = shift[c];
  const float* __restrict__ p = x + (size_t)plane * S;
  float* __restrict__ q1 = y1 + (size_t)b * y1_bs + (size_t)c * S;
  const float* __restrict__ pa = add ? add + (size_t)b * add_bs + (size_t)c * S : nullptr;
  float* __restrict__ q2 = y2 ? y2 + (size_t)plane * S : nullptr;
  // bf (optional): y1 also as bf16, [b][c][Tp] with its own batch stride (the concat's copy, conv3's weight gradient)
  unsigned short* __restrict__ qf = bf ? bf + (size_t)b * bf_bs + (size_t)c * Tp : nullptr;
  const bool vec = (S & 1) == 0 && ((((size_t)p) | ((size_t)q1) | ((size_t)pa) | ((size_t)q2)) & 7) == 0;
  if (vec) {  // 16 bytes per lane (planes start on 8-byte boundaries), a 2-float tail when S % 4 == 2
    for (int i = 4 * (blockIdx.y * NT + threadIdx.x); i < S; i += 4 * gridDim.y * NT) {
      if (i + 3 < S) {
        const f4a8 v = *reinterpret_cast<const f4a8*>(p + i) * sc + sh;
        *reinterpret_cast<f4a8*>(q1 + i) = v;
        if (q2) *reinterpret_cast<f4a8*>(q2 + i) = v + *reinterpret_cast<const f4a8*>(pa + i);
        if (qf) {
#pragma unroll
          for (int e = 0; e < 4; ++e) qf[i + e] = bf16_bits(v[e]);
        }
      } else {
        const f2a8 v = *reinterpret_cast<const f2a8*>(p + i) * sc + sh;
        *reinterpret_cast<f2a8*>(q1 + i) = v;
        if (q2) *reinterpret_cast<f2a8*>(q2 + i) = v + *reinterpret_cast<const f2a8*>(pa + i);
        if (qf) {
          qf[i] = bf16_bits(v[0]);
          qf[i + 1] = bf16_bits(v[1]);
        }
      }
    }
    return;
  }
  for (int i = blockIdx.y * NT + threadIdx.x; i < S; i += gridDim.y * NT) {
    const float v = p[i] * sc + sh;
    q1[i] = v;
    if (q2) q2[i] = v + pa[i];
    if (qf) qf[i] = bf16_bits(v);
  }
}

// out[c] = sum_{b,s} x[b][c][s]   (bias gradients).  grid (C, nsplit): a channel's utterances are
// split over nsplit workgroups so narrow layers (the 64-channel Res2 groups) still fill the chip;
// the fp64 partials are folded in a fixed order by channel_sum_final_kernel (deterministic).
__global__ __launch_bounds__(NT) void channel_sum_kernel(const float* __restrict__ x, int B, int S,
                                                         size_t bstride, int b_per_split,
                                                         float* __restrict__ out,
                                                         double* __restrict__ partial) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x;
  const int b_lo = blockIdx.y * b_per_split, b_hi = min(B, b_lo + b_per_split);
  double d = 0.0;
  for (int b = b_lo; b < b_hi; ++b) {
    const float* __restrict__ p = x + (size_t)b * bstride + (size_t)c * S;
    float s = 0.0f;
    for (int i = threadIdx.x; i < S; i += NT) s += p[i];
    d += (double)s;
  }
  d = block_sum_d(d, sh);
  if (threadIdx.x == 0) {
    if (gridDim.y == 1) out[c] = (float)d;
    else partial[(size_t)c * gridDim.y + blockIdx.y] = d;
  }
}

__global__ __launch_bounds__(NT) void channel_sum_final_kernel(const double* __restrict__ partial, int C,
                                                               int nsplit, float* __restrict__ out) {
  const int c = blockIdx.x * NT + threadIdx.x;
  if (c >= C) return;
  double d = 0.0;
  for (int k = 0; k < nsplit; ++k) d += partial[(size_t)c * nsplit + k];
  out[c] = (float)d;
}

// per (b,c) row over T: mean and std = sqrt(clamp(unbiased var, 1e-4)) (ecapa_tdnn.py:178);
// one wave per row.  std may be null (SE squeeze only needs the mean, ecapa_tdnn.py:19).  Rows of up to
// 64 * ROW_R frames (the reference's 750 fits) stay in registers between the mean and the variance pass: the
// (B, 1536, T) context tensor is read once, not twice (0.28 -> 0.13 ms).
constexpr int ROW_R = 16;
__global__ __launch_bounds__(NT) void row_stats_kernel(const float* __restrict__ x, size_t rows,
                                                       int T, float* __restrict__ mean,
                                                       float* __restrict__ std_, float clamp_min) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* __restrict__ p = x + row * T;
  if (std_ != nullptr && T <= 64 * ROW_R) {
    float v[ROW_R];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < ROW_R; ++k) {
      const int t = lane + 64 * k;
      v[k] = t < T ? p[t] : 0.0f;
    }
    // same order of additions per lane as the loop below: t = lane, lane + 64, ...
#pragma unroll
    for (int k = 0; k < ROW_R; ++k)
      if (lane + 64 * k < T) s += v[k];
    const float m = air_wave_sum(s) / (float)T;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < ROW_R; ++k)
      if (lane + 64 * k < T) {
        const float d = v[k] - m;
        q = fmaf(d, d, q);
      }
    q = air_wave_sum(q) / (float)(T - 1);
    if (lane == 0) {
      mean[row] = m;
      std_[row] = sqrtf(fmaxf(q, clamp_min));
    }
    return;
  }
  float s = 0.0f;
  for (int t = lane; t < T; t += 64) s += p[t];
  const float m = air_wave_sum(s) / (float)T;
  if (lane == 0) mean[row] = m;
  if (std_ != nullptr) {
    float q = 0.0f;
    for (int t = lane; t < T; t += 64) {
      const float d = p[t] - m;
      q = fmaf(d, d, q);
    }
    q = air_wave_sum(q) / (float)(T - 1);
    if (lane == 0) std_[row] = sqrtf(fmaxf(q, clamp_min));
  }
}

// dx[row][t] (+)= dmean/T + dstd * (x - mean) / ((T-1) std)   [dstd term only where var > clamp]
__global__ __launch_bounds__(NT) void row_stats_bwd_kernel(
    const float* __restrict__ x, size_t rows, int T, const float* __restrict__ mean,
    const float* __restrict__ std_, const float* __restrict__ dmean,
    const float* __restrict__ dstd, float clamp_min, float* __restrict__ dx, int accumulate, int relu_mask,
    float* __restrict__ rowsum, unsigned short* __restrict__ bf, int Tp) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float m = mean[row];
  const float k0 = dmean ? dmean[row] / (float)T : 0.0f;
  float k1 = 0.0f;
  if (dstd != nullptr) {
    const float sd = std_[row];
    if (sd * sd > clamp_min) k1 = dstd[row] / ((float)(T - 1) * sd);
  }
  // relu_mask: x is a ReLU output (ecapa_tdnn.py:173) and dx the gradient w.r.t. its pre-activation;
  // rowsum[row] = sum_t of the result (its sum over b is the conv bias gradient)
  float s = 0.0f;
  for (int t = lane; t < T; t += 64) {
    const float xv = x[row * T + t];
    float v = k0 + k1 * (xv - m);
    if (accumulate) v += dx[row * T + t];
    if (relu_mask && !(xv > 0.0f)) v = 0.0f;
    dx[row * T + t] = v;
    if (bf) bf[row * Tp + t] = bf16_bits(v);  // bf16 copy [row][Tp]: the weight-gradient GEMM's operand
    s += v;
  }
  if (rowsum != nullptr) {
    s = air_wave_sum(s);
    if (lane == 0) rowsum[row] = s;
  }
}

// SE gate + residual (ecapa_tdnn.py:27-29, :93): out = x * sigmoid(z[b][c]) + res; wave per row
__global__ __launch_bounds__(NT) void se_scale_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ z,
                                                      const float* __restrict__ res, size_t res_b,
                                                      int C, int T, float* __restrict__ out,
                                                      size_t out_b, size_t rows, unsigned short* __restrict__ bf,
                                                      size_t bf_b, int Tp) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float g = 1.0f / (1.0f + expf(-z[row]));
  const float* __restrict__ px = x + row * T;
  const float* __restrict__ pr = res + b * res_b + c * T;
  float* __restrict__ po = out + b * out_b + c * T;
  // bf (optional): the result also as bf16 at bf[b * bf_b + c * Tp + t] - the block output's slice of the concat's
  // bf16 copy, which layer4's GEMM and three weight gradients read
  unsigned short* __restrict__ pf = bf ? bf + b * bf_b + c * (size_t)Tp : nullptr;
  for (int t = lane; t < T; t += 64) {
    const float v = fmaf(px[t], g, pr[t]);
    po[t] = v;
    if (pf) pf[t] = bf16_bits(v);
  }
}

// backward: dx = dout * sigmoid(z); dz = s(1-s) * sum_t dout*x   (dres = dout, same tensor)
__global__ __launch_bounds__(NT) void se_scale_bwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ z,
                                                          const float* __restrict__ dout,
                                                          size_t dout_b, int C, int T,
                                                          float* __restrict__ dx,
                                                          float* __restrict__ dz, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float g = 1.0f / (1.0f + expf(-z[row]));
  const float* __restrict__ px = x + row * T;
  const float* __restrict__ pd = dout + b * dout_b + c * T;
  float acc = 0.0f;
  for (int t = lane; t < T; t += 64) {
    const float d = pd[t];
    acc = fmaf(d, px[t], acc);
    dx[row * T + t] = d * g;
  }
  acc = air_wave_sum(acc);
  if (lane == 0) dz[row] = acc * g * (1.0f - g);
}

// Attentive statistics pooling (ecapa_tdnn.py:143-185): per (b,c) row
//   w = softmax_T(a);  mu = sum x w;  sg = sqrt(clamp(sum x^2 w - mu^2, 1e-4))
// a is overwritten with w (saved for backward); out = [mu | sg] as (B, 2C).
// One wave per row.  Rows of up to 64 * ASP_R frames (the reference's 750 fits) are held in
// registers: each tensor is read from memory once and every exp is evaluated once.
constexpr int ASP_R = 16;
template <bool CACHED>
__global__ __launch_bounds__(NT) void asp_fwd_kernel(const float* __restrict__ x,
                                                     float* __restrict__ a, int C, int T,
                                                     float* __restrict__ out, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  float* __restrict__ pa = a + row * T;
  const float* __restrict__ px = x + row * T;
  float s1 = 0.0f, s2 = 0.0f;
  if (CACHED) {
    float e[ASP_R], xv[ASP_R];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < ASP_R; ++k) {
      const int t = lane + 64 * k;
      e[k] = t < T ? pa[t] : -INFINITY;
      xv[k] = t < T ? px[t] : 0.0f;
      m = fmaxf(m, e[k]);
    }
    m = air_wave_max(m);
    float se = 0.0f;
#pragma unroll
    for (int k = 0; k < ASP_R; ++k)
      if (lane + 64 * k < T) {
        e[k] = expf(e[k] - m);
        se += e[k];
      }
    se = air_wave_sum(se);
#pragma unroll
    for (int k = 0; k < ASP_R; ++k) {
      const int t = lane + 64 * k;
      if (t < T) {
        const float w = e[k] / se;
        pa[t] = w;
        s1 = fmaf(xv[k], w, s1);
        s2 = fmaf(xv[k] * xv[k], w, s2);
      }
    }
  } else {
    float m = -INFINITY;
    for (int t = lane; t < T; t += 64) m = fmaxf(m, pa[t]);
    m = air_wave_max(m);
    float se = 0.0f;
    for (int t = lane; t < T; t += 64) se += expf(pa[t] - m);
    se = air_wave_sum(se);
    for (int t = lane; t < T; t += 64) {
      const float w = expf(pa[t] - m) / se;
      const float xv = px[t];
      pa[t] = w;
      s1 = fmaf(xv, w, s1);
      s2 = fmaf(xv * xv, w, s2);
    }
  }
  s1 = air_wave_sum(s1);
  s2 = air_wave_sum(s2);
  if (lane == 0) {
    out[b * 2 * C + c] = s1;
    out[b * 2 * C + C + c] = sqrtf(fmaxf(s2 - s1 * s1, 1e-4f));
  }
}

// backward: given dout (B,2C): dx (accumulated into dx_acc) and da (written over w);
// rowsum[row] (optional) = sum_t da: summed over b it is the gradient of attention.3's bias
template <bool CACHED>
__global__ __launch_bounds__(NT) void asp_bwd_kernel(const float* __restrict__ x,
                                                     float* __restrict__ w, int C, int T,
                                                     const float* __restrict__ out,
                                                     const float* __restrict__ dout,
                                                     float* __restrict__ dx, int accumulate,
                                                     float* __restrict__ rowsum, size_t rows,
                                                     unsigned short* __restrict__ bf, int Tp) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float mu = out[b * 2 * C + c], sg = out[b * 2 * C + C + c];
  const float dmu = dout[b * 2 * C + c], dsg = dout[b * 2 * C + C + c];
  const float dq = (sg * sg > 1e-4f) ? dsg / (2.0f * sg) : 0.0f;  // clamp passes no gradient
  const float dm = dmu - 2.0f * mu * dq;
  float* __restrict__ pw = w + row * T;
  const float* __restrict__ px = x + row * T;
  float dot = 0.0f, rs = 0.0f;
  if (CACHED) {
    float wv[ASP_R], xv[ASP_R];
#pragma unroll
    for (int k = 0; k < ASP_R; ++k) {
      const int t = lane + 64 * k;
      wv[k] = t < T ? pw[t] : 0.0f;
      xv[k] = t < T ? px[t] : 0.0f;
      if (t < T) dot = fmaf(wv[k], dm * xv[k] + dq * xv[k] * xv[k], dot);
    }
    dot = air_wave_sum(dot);
#pragma unroll
    for (int k = 0; k < ASP_R; ++k) {
      const int t = lane + 64 * k;
      if (t < T) {
        const float dwv = dm * xv[k] + dq * xv[k] * xv[k];
        const float g = dm * wv[k] + 2.0f * dq * xv[k] * wv[k];
        dx[row * T + t] = accumulate ? dx[row * T + t] + g : g;
        const float da = wv[k] * (dwv - dot);  // softmax backward
        pw[t] = da;
        if (bf) bf[row * Tp + t] = bf16_bits(da);  // bf16 copy [row][Tp]: the weight-gradient GEMM's operand
        rs += da;
      }
    }
  } else {
    for (int t = lane; t < T; t += 64) {
      const float xv = px[t];
      dot = fmaf(pw[t], dm * xv + dq * xv * xv, dot);
    }
    dot = air_wave_sum(dot);
    for (int t = lane; t < T; t += 64) {
      const float xv = px[t], wv = pw[t];
      const float dwv = dm * xv + dq * xv * xv;
      const float g = dm * wv + 2.0f * dq * xv * wv;
      dx[row * T + t] = accumulate ? dx[row * T + t] + g : g;
      const float da = wv * (dwv - dot);  // softmax backward
      pw[t] = da;
      if (bf) bf[row * Tp + t] = bf16_bits(da);
      rs += da;
    }
  }
  if (rowsum != nullptr) {
    rs = air_wave_sum(rs);
    if (lane == 0) rowsum[row] = rs;
  }
}

// dx *= (y > 0): backward of a stand-alone ReLU (ecapa_tdnn.py:173)
__global__ void relu_mask_kernel(float* __restrict__ dx, const float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.0f ? dx[i] : 0.0f;
}

// out[row] = sum_t x[row][t]
__global__ __launch_bounds__(NT) void row_sum_kernel(const float* __restrict__ x, size_t rows, int T,
                                                     float* __restrict__ out) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.0f;
  for (int t = lane; t < T; t += 64) s += x[row * T + t];
  s = air_wave_sum(s);
  if (lane == 0) out[row] = s;
}

}  // namespace

extern "C" {

int air_relu_mask(float* dx, const float* y, size_t n, air_stream_t stream) {
  if (!dx || !y) return AIR_EINVAL;
  if (n == 0) return AIR_OK;
  size_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)g), dim3(256), 0, air_stream(stream), dx, y, n);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_row_sum(const float* x, int B, int C, int T, float* out, air_stream_t stream) {
  if (!x || !out || B <= 0 || C <= 0 || T <= 0) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                     air_stream(stream), x, rows, T, out);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_add_strided_ex(float* out, size_t out_bstride, const float* a, size_t a_bstride, const float* b,
                       size_t b_bstride, int B, int C, int S, unsigned short* out_bf16, size_t out_bf16_bstride,
                       int out_bf16_tp, air_stream_t stream) {
  if (!out || !a || B <= 0 || C <= 0 || S <= 0) return AIR_EINVAL;
  if (out_bf16 && out_bf16_tp < S) return AIR_EINVAL;
  const int CS = C * S;
  dim3 grid(min((CS + NT - 1) / NT, 1024), B);
  hipLaunchKernelGGL(add_strided_kernel, grid, dim3(NT), 0, air_stream(stream), out, out_bstride,
                     a, a_bstride, b, b_bstride, CS, S, out_bf16,
                     out_bf16_bstride ? out_bf16_bstride : (size_t)C * out_bf16_tp, out_bf16_tp);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_add_strided(float* out, size_t out_bstride, const float* a, size_t a_bstride,
                    const float* b, size_t b_bstride, int B, int C, int S, air_stream_t stream) {
  return air_add_strided_ex(out, out_bstride, a, a_bstride, b, b_bstride, B, C, S, nullptr, 0, 0, stream);
}

int air_res2_bn_apply_ex(const float* x, int B, int C, int S, const float* scale, const float* shift, float* y1,
                         size_t y1_bstride, const float* add, size_t add_bstride, float* y2, unsigned short* y1_bf16,
                         size_t y1_bf16_bstride, int y1_bf16_tp, air_stream_t stream) {
  if (!x || !scale || !shift || !y1 || B <= 0 || C <= 0 || S <= 0 || ((add == nullptr) != (y2 == nullptr)))
    return AIR_EINVAL;
  if (y1_bf16 && y1_bf16_tp < S) return AIR_EINVAL;
  hipLaunchKernelGGL(res2_bn_apply_kernel, dim3(B * C, (S + NT * 4 - 1) / (NT * 4)), dim3(NT), 0, air_stream(stream),
                     x, C, S, scale, shift, y1, y1_bstride ? y1_bstride : (size_t)C * S, add,
                     add_bstride ? add_bstride : (size_t)C * S, y2, y1_bf16,
                     y1_bf16_bstride ? y1_bf16_bstride : (size_t)C * y1_bf16_tp, y1_bf16_tp);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_res2_bn_apply(const float* x, int B, int C, int S, const float* scale, const float* shift, float* y1,
                      size_t y1_bstride, const float* add, size_t add_bstride, float* y2, air_stream_t stream) {
  return air_res2_bn_apply_ex(x, B, C, S, scale, shift, y1, y1_bstride, add, add_bstride, y2, nullptr, 0, 0, stream);
}

static int channel_sum_split(int B, int C, int* b_per_split) {
  int want = 2048 / C;
  if (want < 1) want = 1;
  if (want > B) want = B;
  *b_per_split = (B + want - 1) / want;
  return (B + *b_per_split - 1) / *b_per_split;
}

size_t air_channel_sum_ws_bytes(int B, int C) {
  if (B <= 0 || C <= 0) return 0;
  int per;
  return (size_t)channel_sum_split(B, C, &per) * C * sizeof(double) + 256;
}

int air_channel_sum(const float* x, int B, int C, int S, size_t bstride, float* out, void* ws,
                    size_t ws_bytes, air_stream_t stream) {
  if (!x || !out || B <= 0 || C <= 0 || S <= 0) return AIR_EINVAL;
  int per;
  const int nsplit = channel_sum_split(B, C, &per);
  if (nsplit > 1 && (!ws || ws_bytes < air_channel_sum_ws_bytes(B, C))) return AIR_EWORKSPACE;
  double* partial = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C, nsplit), dim3(NT), 0, air_stream(stream), x, B, S,
                     bstride ? bstride : (size_t)C * S, per, out, partial);
  AIR_CHECK_LAUNCH();
  if (nsplit > 1) {
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, air_stream(stream),
                       partial, C, nsplit, out);
    AIR_CHECK_LAUNCH();
  }
  return AIR_OK;
}

int air_row_stats(const float* x, int B, int C, int T, float* mean, float* std_or_null,
                  float clamp_min, air_stream_t stream) {
  if (!x || !mean || B <= 0 || C <= 0 || T <= 1) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                     air_stream(stream), x, rows, T, mean, std_or_null, clamp_min);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_row_stats_bwd_ex(const float* x, int B, int C, int T, const float* mean, const float* std_,
                         const float* dmean, const float* dstd, float clamp_min, float* dx,
                         int accumulate, int relu_mask, float* rowsum, unsigned short* dx_bf16, int dx_bf16_tp,
                         air_stream_t stream) {
  if (!x || !mean || !dx || B <= 0 || C <= 0 || T <= 1) return AIR_EINVAL;
  if (dx_bf16 && dx_bf16_tp < T) return AIR_EINVAL;
  if (dstd != nullptr && std_ == nullptr) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(row_stats_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                     air_stream(stream), x, rows, T, mean, std_, dmean, dstd, clamp_min, dx,
                     accumulate, relu_mask, rowsum, dx_bf16, dx_bf16_tp);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_row_stats_bwd(const float* x, int B, int C, int T, const float* mean, const float* std_,
                      const float* dmean, const float* dstd, float clamp_min, float* dx,
                      int accumulate, int relu_mask, float* rowsum, air_stream_t stream) {
  return air_row_stats_bwd_ex(x, B, C, T, mean, std_, dmean, dstd, clamp_min, dx, accumulate, relu_mask, rowsum,
                              nullptr, 0, stream);
}

int air_se_scale_fwd_ex(const float* x, const float* z, const float* res, size_t res_bstride, int B,
                        int C, int T, float* out, size_t out_bstride, unsigned short* out_bf16, size_t out_bf16_bstride,
                        int out_bf16_tp, air_stream_t stream) {
  if (!x || !z || !res || !out || B <= 0 || C <= 0 || T <= 0) return AIR_EINVAL;
  if (out_bf16 && out_bf16_tp < T) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                     air_stream(stream), x, z, res, res_bstride ? res_bstride : (size_t)C * T, C, T,
                     out, out_bstride ? out_bstride : (size_t)C * T, rows, out_bf16,
                     out_bf16_bstride ? out_bf16_bstride : (size_t)C * out_bf16_tp, out_bf16_tp);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_se_scale_fwd(const float* x, const float* z, const float* res, size_t res_bstride, int B,
                     int C, int T, float* out, size_t out_bstride, air_stream_t stream) {
  return air_se_scale_fwd_ex(x, z, res, res_bstride, B, C, T, out, out_bstride, nullptr, 0, 0, stream);
}

int air_se_scale_bwd(const float* x, const float* z, const float* dout, size_t dout_bstride, int B,
                     int C, int T, float* dx, float* dz, air_stream_t stream) {
  if (!x || !z || !dout || !dx || !dz || B <= 0 || C <= 0 || T <= 0) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(se_scale_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                     air_stream(stream), x, z, dout, dout_bstride ? dout_bstride : (size_t)C * T, C,
                     T, dx, dz, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_asp_fwd(const float* x, float* logits_to_w, int B, int C, int T, float* out,
                air_stream_t stream) {
  if (!x || !logits_to_w || !out || B <= 0 || C <= 0 || T <= 0) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  if (T <= 64 * ASP_R)
    hipLaunchKernelGGL(asp_fwd_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                       air_stream(stream), x, logits_to_w, C, T, out, rows);
  else
    hipLaunchKernelGGL(asp_fwd_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                       air_stream(stream), x, logits_to_w, C, T, out, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_asp_bwd_ex(const float* x, float* w_to_dlogits, int B, int C, int T, const float* out,
                   const float* dout, float* dx, int accumulate, float* rowsum, unsigned short* dlogits_bf16,
                   int dlogits_bf16_tp, air_stream_t stream) {
  if (!x || !w_to_dlogits || !out || !dout || !dx || B <= 0 || C <= 0 || T <= 0) return AIR_EINVAL;
  if (dlogits_bf16 && dlogits_bf16_tp < T) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  if (T <= 64 * ASP_R)
    hipLaunchKernelGGL(asp_bwd_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                       air_stream(stream), x, w_to_dlogits, C, T, out, dout, dx, accumulate, rowsum, rows, dlogits_bf16,
                       dlogits_bf16_tp);
  else
    hipLaunchKernelGGL(asp_bwd_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0,
                       air_stream(stream), x, w_to_dlogits, C, T, out, dout, dx, accumulate, rowsum, rows, dlogits_bf16,
                       dlogits_bf16_tp);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_asp_bwd(const float* x, float* w_to_dlogits, int B, int C, int T, const float* out,
                const float* dout, float* dx, int accumulate, float* rowsum, air_stream_t stream) {
  return air_asp_bwd_ex(x, w_to_dlogits, B, C, T, out, dout, dx, accumulate, rowsum, nullptr, 0, stream);
}

}  // extern "C"
