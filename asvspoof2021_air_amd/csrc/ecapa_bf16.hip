// bf16-RESIDENT activations for ECAPA-TDNN (BASELINE.json configs[2], round 3): every (B, C, T) tensor between
// the layers of ecapa_tdnn.py:64-95,152-187 lives in HBM as bf16 rows [b][c][Tp] - Tp = air_conv1d_bf16_tp(T)
// frames per row, a multiple of 256, frames t >= T ZERO (every kernel here keeps that invariant: the GEMMs of
// conv1d_bf16.hip read these rows as operands without masks).  Element (b, c, t) of a tensor or of a
// channel-slice view of a wider one sits at p[b * bs + c * Tp + t], bs = the batch stride in elements.
// These are the HBM-bound passes between the GEMMs: BatchNorm statistics / apply / backward, the Res2 chain
// step, the SE gate, the context statistics and the attentive-statistics pooling - the arithmetic of their fp32
// counterparts (norm_act.hip, ecapa_ops.hip) evaluated in fp32 on the bf16 values, every stored result rounded
// once to bf16 (nearest even, v_cvt_pk_bf16_f32).  What is rounded where is stated by oracle/ecapa.py
// (bf16="resident"); the tensors a torch.autocast(bfloat16) run of the reference would hold in bf16 are the same.
//
// One wave per row, 8 bytes (4 values) per lane and access: a 768-frame row is three 512-byte wave accesses on
// 16-byte-aligned rows (the fp32 kernels needed alignment peeling for T = 750).  Reductions are two-stage and
// deterministic (fp64 partials per (channel, split), folded in a fixed order), as in norm_act.hip.
#include "air_common.h"

namespace {

constexpr int NT = 256;
constexpr int HMAXV = 8;  // 8-byte vectors per lane and row: Tp <= 2048
typedef unsigned short u16;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void unpack4(uint2 u, float* v) {
  v[0] = lo_f(u.x); v[1] = hi_f(u.x); v[2] = lo_f(u.y); v[3] = hi_f(u.y);
}
// the value a bf16 store of `a` holds
__device__ __forceinline__ float rnd(float a) { return lo_f(pack2(a, 0.0f)); }
// four results -> one 8-byte store, zeros for frames t0 + e >= T
__device__ __forceinline__ uint2 pack4_masked(const float* v, int t0, int T) {
  const float a = t0 < T ? v[0] : 0.0f, b = t0 + 1 < T ? v[1] : 0.0f;
  const float c = t0 + 2 < T ? v[2] : 0.0f, d = t0 + 3 < T ? v[3] : 0.0f;
  return make_uint2(pack2(a, b), pack2(c, d));
}
__device__ __forceinline__ const uint2* row_ld(const u16* p) { return reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ uint2* row_st(u16* p) { return reinterpret_cast<uint2*>(p); }

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

// idx -> (idx / nv, idx % nv) for 0 <= idx < 2^22 without the ~35-instruction integer division sequence: the flat
// loops below do it once per 8-byte load (rnv = 1.0f / nv; the float quotient is off by at most one)
__device__ __forceinline__ void divmod_flat(int idx, int nv, float rnv, int& q, int& r) {
  q = (int)((float)idx * rnv);
  r = idx - q * nv;
  if (r < 0) { r += nv; --q; }
  else if (r >= nv) { r -= nv; ++q; }
}

int h_splits(int B, int C) {
  int s = 2048 / C;
  if (s < 1) s = 1;
  if (s > B) s = B;
  if (s < (B + 8190) / 8191) s = (B + 8190) / 8191;  // (utterances per split) x (Tp / 4 <= 512) < 2^22: divmod_flat
  return s;
}
bool h_shape_ok(int B, int C, int T, int Tp) {
  return B > 0 && C > 0 && T > 0 && Tp >= T && Tp % 8 == 0 && Tp <= 256 * HMAXV;
}
int h_row_grid(size_t rows) { return (int)((rows + NT / 64 - 1) / (NT / 64)); }

// ---- BatchNorm statistics: partial[(c * nsplit + split) * 2 + {0, 1}] = sum, sum of squares of (x - K)
__global__ __launch_bounds__(NT) void h_bn_partial_kernel(const u16* __restrict__ x, size_t bs, int B, int C, int T,
                                                          int Tp, int nsplit, double* __restrict__ partial) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x / nsplit, split = blockIdx.x - c * nsplit;
  const int per = (B + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(B, b0 + per);
  const float K = lo_f((unsigned)x[(size_t)c * Tp]);
  const int nv = (T + 3) >> 2;
  // one flat loop over the split's (utterance, 4-frame vector) pairs, four loads in flight per thread: the Res2
  // branches' tensors are 12 MB, a launch is a handful of vectors per thread and lives on its load latency
  const int total = (b1 - b0) * nv;
  const float rnv = 1.0f / (float)nv;
  float s1 = 0.0f, s2 = 0.0f;
  const u16* __restrict__ xc = x + (size_t)c * Tp;
#pragma unroll 4
  for (int idx = threadIdx.x; idx < total; idx += NT) {
    int bq, i;
    divmod_flat(idx, nv, rnv, bq, i);
    float v[4];
    unpack4(row_ld(xc + (size_t)(b0 + bq) * bs)[i], v);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * i + e < T) {
        const float d = v[e] - K;
        s1 += d;
        s2 = fmaf(d, d, s2);
      }
  }
  double d1 = block_sum_d((double)s1, sh);
  double d2 = block_sum_d((double)s2, sh);
  if (threadIdx.x == 0) {
    partial[(size_t)blockIdx.x * 2] = d1;
    partial[(size_t)blockIdx.x * 2 + 1] = d2;
  }
}

__global__ void h_bn_finalize_kernel(const u16* __restrict__ x, int Tp, const double* __restrict__ partial, int nsplit,
                                     int C, double N, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float eps, float momentum, float* __restrict__ running_mean,
                                     float* __restrict__ running_var, float* __restrict__ mean,
                                     float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per channel (air_wave_ordered_sum_d)
  if (c >= C) return;
  // everything the tail needs is requested in front of the sums: one memory round trip for the kernel
  const float K = lo_f((unsigned)x[(size_t)c * Tp]), ga = gamma[c], be = beta[c];
  const float rm = running_mean != nullptr ? running_mean[c] : 0.0f, rv = running_mean != nullptr ? running_var[c] : 0.0f;
  double s12[2];
  air_wave_ordered_sums_d<2>(partial + (size_t)c * nsplit * 2, nsplit, 2, s12);
  const double s1 = s12[0], s2 = s12[1];
  if ((threadIdx.x & 63) != 0) return;
  const double ms = s1 / N;
  double var = s2 / N - ms * ms;
  if (var < 0.0) var = 0.0;
  const double m = ms + (double)K;
  const float mf = (float)m;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = mf;
  invstd[c] = is;
  const float sc = ga * is;
  scale[c] = sc;
  shift[c] = be - mf * sc;
  if (running_mean != nullptr) {
    const double unbiased = N > 1.0 ? var * N / (N - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * rm + momentum * mf;
    running_var[c] = (1.0f - momentum) * rv + momentum * (float)unbiased;
  }
}

// The same finalisation from the records a convolution's epilogue wrote (air_h_conv1d_tap_ex, `stats`): per channel
// NR pairs {sum, sum of squares} of 32 stored values each (fp32, values O(1): exact to ~1e-7), merged in fp64 in a
// fixed order by one workgroup per channel.  Unshifted sums: the variance is formed in fp64.
__global__ __launch_bounds__(256) void h_bn_finalize_records_kernel(
    const float* __restrict__ rec, int NR, int C, double N, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
    float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
    float* __restrict__ shift) {
  __shared__ double sh[2][4];
  const int c = blockIdx.x, tid = threadIdx.x;
  const float2* __restrict__ r = reinterpret_cast<const float2*>(rec) + (size_t)c * NR;
  const float ga = gamma[c], be = beta[c];
  const float rm = running_mean != nullptr ? running_mean[c] : 0.0f, rv = running_mean != nullptr ? running_var[c] : 0.0f;
  double s1 = 0.0, s2 = 0.0;
  int g = tid;
  for (; g + 768 < NR; g += 1024) {
    const float2 v0 = r[g], v1 = r[g + 256], v2 = r[g + 512], v3 = r[g + 768];
    s1 += (double)v0.x; s2 += (double)v0.y;
    s1 += (double)v1.x; s2 += (double)v1.y;
    s1 += (double)v2.x; s2 += (double)v2.y;
    s1 += (double)v3.x; s2 += (double)v3.y;
  }
  for (; g < NR; g += 256) {
    const float2 v = r[g];
    s1 += (double)v.x; s2 += (double)v.y;
  }
  s1 = air_wave_sum_d(s1);
  s2 = air_wave_sum_d(s2);
  if ((tid & 63) == 0) { sh[0][tid >> 6] = s1; sh[1][tid >> 6] = s2; }
  __syncthreads();
  if (tid != 0) return;
  s1 = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
  s2 = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
  const double m = s1 / N;
  double var = s2 / N - m * m;
  if (var < 0.0) var = 0.0;
  const float mf = (float)m;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = mf;
  invstd[c] = is;
  const float sc = ga * is;
  scale[c] = sc;
  shift[c] = be - mf * sc;
  if (running_mean != nullptr) {
    const double unbiased = N > 1.0 ? var * N / (N - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * rm + momentum * mf;
    running_var[c] = (1.0f - momentum) * rv + momentum * (float)unbiased;
  }
}

// ---- y = bf16(x * scale[c] + shift[c]); rowmean[b * C + c] (optional) = mean over t of the STORED values
__global__ __launch_bounds__(NT) void h_bn_apply_kernel(const u16* __restrict__ x, size_t xbs, int C, int T, int Tp,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        u16* __restrict__ y, size_t ybs, float* __restrict__ rowmean,
                                                        size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float sc = scale[c], sh = shift[c];
  const uint2* __restrict__ p = row_ld(x + b * xbs + c * Tp);
  uint2* __restrict__ q = row_st(y + b * ybs + c * Tp);
  float s = 0.0f;
  for (int i = lane; i < Tp / 4; i += 64) {
    float v[4];
    unpack4(p[i], v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc, sh);
    const uint2 o = pack4_masked(v, 4 * i, T);
    q[i] = o;
    if (rowmean != nullptr) s += (lo_f(o.x) + hi_f(o.x)) + (lo_f(o.y) + hi_f(o.y));
  }
  if (rowmean != nullptr) {
    s = air_wave_sum(s);
    if (lane == 0) rowmean[row] = s / (float)T;
  }
}

// ---- Res2 chain step (ecapa_tdnn.py:78-83): y1 = bf16(x * scale + shift) -> its slice of the concat;
// y2 (optional) = bf16(y1 + add): the next branch's input "sp + spx[i + 1]" (both addends are bf16 tensors)
__global__ __launch_bounds__(NT) void h_res2_kernel(const u16* __restrict__ x, size_t xbs, int C, int T, int Tp,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    u16* __restrict__ y1, size_t y1bs, const u16* __restrict__ add,
                                                    size_t addbs, u16* __restrict__ y2, size_t y2bs, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float sc = scale[c], sh = shift[c];
  const uint2* __restrict__ p = row_ld(x + b * xbs + c * Tp);
  uint2* __restrict__ q1 = row_st(y1 + b * y1bs + c * Tp);
  const uint2* __restrict__ pa = add ? row_ld(add + b * addbs + c * Tp) : nullptr;
  uint2* __restrict__ q2 = y2 ? row_st(y2 + b * y2bs + c * Tp) : nullptr;
  for (int i = lane; i < Tp / 4; i += 64) {
    float v[4];
    unpack4(p[i], v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc, sh);
    const uint2 o = pack4_masked(v, 4 * i, T);
    q1[i] = o;
    if (q2) {
      float a[4], w[4];
      unpack4(pa[i], a);
      unpack4(o, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] += a[e];
      q2[i] = pack4_masked(w, 4 * i, T);
    }
  }
}

// ---- SE gate + block residual (ecapa_tdnn.py:27-29, :93): out = bf16(x * sigmoid(z[b][c]) + res)
__global__ __launch_bounds__(NT) void h_se_fwd_kernel(const u16* __restrict__ x, size_t xbs, const float* __restrict__ z,
                                                      const u16* __restrict__ res, size_t rbs, int C, int T, int Tp,
                                                      u16* __restrict__ out, size_t obs, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float g = 1.0f / (1.0f + expf(-z[row]));
  const uint2* __restrict__ px = row_ld(x + b * xbs + c * Tp);
  const uint2* __restrict__ pr = row_ld(res + b * rbs + c * Tp);
  uint2* __restrict__ po = row_st(out + b * obs + c * Tp);
  for (int i = lane; i < Tp / 4; i += 64) {
    float v[4], r[4];
    unpack4(px[i], v);
    unpack4(pr[i], r);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], g, r[e]);
    po[i] = pack4_masked(v, 4 * i, T);
  }
}

// backward: dx = bf16(dout * sigmoid(z)); dz = s (1 - s) sum_t dout * x   (d res = dout, the same tensor)
__global__ __launch_bounds__(NT) void h_se_bwd_kernel(const u16* __restrict__ x, size_t xbs, const float* __restrict__ z,
                                                      const u16* __restrict__ dout, size_t dbs, int C, int T, int Tp,
                                                      u16* __restrict__ dx, size_t dxbs, float* __restrict__ dz,
                                                      size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float g = 1.0f / (1.0f + expf(-z[row]));
  const uint2* __restrict__ px = row_ld(x + b * xbs + c * Tp);
  const uint2* __restrict__ pd = row_ld(dout + b * dbs + c * Tp);
  uint2* __restrict__ po = row_st(dx + b * dxbs + c * Tp);
  float acc = 0.0f;
  for (int i = lane; i < Tp / 4; i += 64) {
    float v[4], d[4];
    unpack4(px[i], v);
    unpack4(pd[i], d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc = fmaf(d[e], v[e], acc);  // frames t >= T hold zeros
      d[e] *= g;
    }
    po[i] = pack4_masked(d, 4 * i, T);
  }
  acc = air_wave_sum(acc);
  if (lane == 0) dz[row] = acc * g * (1.0f - g);
}

// ---- BatchNorm backward for conv -> ReLU -> BN (ecapa_tdnn.py:67-69): x = the BN input (a ReLU output).
// Incoming gradient g = dy + dy2 + rb_scale * rowbias[b][c]; sums per (channel, split):
//   [0] sum g  [1] sum g xhat  [2] sum_{x>0} g  [3] count_{x>0}  [4] sum_{x>0} xhat
constexpr int HNACC = 5;
__global__ __launch_bounds__(NT) void h_bn_bwd_partial_kernel(
    const u16* __restrict__ x, size_t xbs, const u16* __restrict__ dy, size_t dybs, const u16* __restrict__ dy2,
    size_t dy2bs, const float* __restrict__ rowbias, float rb_scale, int B, int C, int T, int Tp, int nsplit,
    const float* __restrict__ mean, const float* __restrict__ invstd, int want_bias, double* __restrict__ partial) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x / nsplit, split = blockIdx.x - c * nsplit;
  const int per = (B + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(B, b0 + per);
  const float mu = mean[c], is = invstd[c];
  const int nv = (T + 3) >> 2;
  // flat loop over the split's (utterance, 4-frame vector) pairs, as in h_bn_partial_kernel
  const int total = (b1 - b0) * nv;
  const float rnv = 1.0f / (float)nv;
  float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f, s5 = 0.0f;
#pragma unroll 2
  for (int idx = threadIdx.x; idx < total; idx += NT) {
    int bq, i;
    divmod_flat(idx, nv, rnv, bq, i);
    const int b = b0 + bq;
    float xv[4], g[4];
    unpack4(row_ld(x + (size_t)b * xbs + (size_t)c * Tp)[i], xv);
    unpack4(row_ld(dy + (size_t)b * dybs + (size_t)c * Tp)[i], g);
    if (dy2) {
      float g2[4];
      unpack4(row_ld(dy2 + (size_t)b * dy2bs + (size_t)c * Tp)[i], g2);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] += g2[e];
    }
    const float rb = rowbias ? rowbias[(size_t)b * C + c] * rb_scale : 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * i + e < T) {
        const float gg = g[e] + rb;
        const float xh = (xv[e] - mu) * is;
        s1 += gg;
        s2 = fmaf(gg, xh, s2);
        if (want_bias && xv[e] > 0.0f) {
          s3 += gg;
          s4 += 1.0f;
          s5 += xh;
        }
      }
  }
  double d1 = (double)s1, d2 = (double)s2, d3 = (double)s3, d4 = (double)s4, d5 = (double)s5;
  d1 = block_sum_d(d1, sh);
  d2 = block_sum_d(d2, sh);
  if (want_bias) {
    d3 = block_sum_d(d3, sh);
    d4 = block_sum_d(d4, sh);
    d5 = block_sum_d(d5, sh);
  }
  if (threadIdx.x == 0) {
    double* o = partial + (size_t)blockIdx.x * HNACC;
    o[0] = d1; o[1] = d2; o[2] = d3; o[3] = d4; o[4] = d5;
  }
}

__global__ void h_bn_bwd_finalize_kernel(const double* __restrict__ partial, int nsplit, int C, double invN,
                                         const float* __restrict__ gamma, const float* __restrict__ invstd,
                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                         float* __restrict__ dbias) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per channel (air_wave_ordered_sum_d)
  if (c >= C) return;
  const double* o = partial + (size_t)c * nsplit * HNACC;
  const float ga = gamma[c], is = invstd[c];  // requested together with the partial sums
  double s1, s2, s3 = 0.0, s4 = 0.0, s5 = 0.0;
  if (dbias) {
    double t[5];
    air_wave_ordered_sums_d<5>(o, nsplit, HNACC, t);
    s1 = t[0]; s2 = t[1]; s3 = t[2]; s4 = t[3]; s5 = t[4];
  } else {
    double t[2];
    air_wave_ordered_sums_d<2>(o, nsplit, HNACC, t);
    s1 = t[0]; s2 = t[1];
  }
  if ((threadIdx.x & 63) != 0) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  if (dbias) {
    const double k1 = (double)(float)s1 * invN, k2 = (double)(float)s2 * invN;
    dbias[c] = (float)((double)ga * (double)is * (s3 - k1 * s4 - k2 * s5));
  }
}

// The same finalisation from the records of a data-gradient epilogue (air_h_conv1d_tap_ex2, bn_sums): per channel NR
// records of 8 floats {sum g, sum g xhat, sum_{x>0} g, count_{x>0}, sum_{x>0} xhat, 0, 0, 0} over 32 frames each,
// merged in fp64 in a fixed order by one workgroup per channel.
__global__ __launch_bounds__(256) void h_bn_bwd_finalize_records_kernel(
    const float* __restrict__ rec, int NR, int C, double invN, const float* __restrict__ gamma,
    const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
  __shared__ double sh[5][4];
  const int c = blockIdx.x, tid = threadIdx.x;
  const float4* __restrict__ r = reinterpret_cast<const float4*>(rec) + (size_t)c * NR * 2;
  const float ga = gamma[c], is = invstd[c];
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
  for (int g = tid; g < NR; g += 256) {
    const float4 a = r[2 * g], b = r[2 * g + 1];
    s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w; s[4] += (double)b.x;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    s[k] = air_wave_sum_d(s[k]);
    if ((tid & 63) == 0) sh[k][tid >> 6] = s[k];
  }
  __syncthreads();
  if (tid != 0) return;
#pragma unroll
  for (int k = 0; k < 5; ++k) s[k] = ((sh[k][0] + sh[k][1]) + sh[k][2]) + sh[k][3];
  dbeta[c] = (float)s[0];
  dgamma[c] = (float)s[1];
  if (dbias) {
    const double k1 = (double)(float)s[0] * invN, k2 = (double)(float)s[1] * invN;
    dbias[c] = (float)((double)ga * (double)is * (s[2] - k1 * s[3] - k2 * s[4]));
  }
}

// dx = bf16(gamma invstd (g - dbeta / N - xhat dgamma / N)), zero where the ReLU in front of the BN clipped
__global__ __launch_bounds__(NT) void h_bn_bwd_apply_kernel(
    const u16* __restrict__ x, size_t xbs, const u16* dy, size_t dybs, const u16* dy2, size_t dy2bs,
    const float* __restrict__ rowbias, float rb_scale, int C, int T, int Tp, float invN,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta, int relu_in, u16* dx, size_t dxbs, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float mu = mean[c], is = invstd[c];
  const float sc = gamma[c] * is;
  const float k1 = dbeta[c] * invN, k2 = dgamma[c] * invN;
  const float rb = rowbias ? rowbias[row] * rb_scale : 0.0f;
  const uint2* __restrict__ px = row_ld(x + b * xbs + c * Tp);
  const uint2* pg = row_ld(dy + b * dybs + c * Tp);  // may alias dx (in-place gradient joins)
  const uint2* pg2 = dy2 ? row_ld(dy2 + b * dy2bs + c * Tp) : nullptr;
  uint2* po = row_st(dx + b * dxbs + c * Tp);
  for (int i = lane; i < Tp / 4; i += 64) {
    float xv[4], g[4];
    unpack4(px[i], xv);
    unpack4(pg[i], g);
    if (pg2) {
      float g2[4];
      unpack4(pg2[i], g2);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] += g2[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - mu) * is;
      float r = sc * ((g[e] + rb) - k1 - xh * k2);
      if (relu_in && !(xv[e] > 0.0f)) r = 0.0f;
      g[e] = r;
    }
    po[i] = pack4_masked(g, 4 * i, T);
  }
}

// ---- context statistics (ecapa_tdnn.py:178): mean_T and sqrt(clamp(var_T unbiased, clamp_min)) per row
__global__ __launch_bounds__(NT) void h_row_stats_kernel(const u16* __restrict__ x, int T, int Tp, float* __restrict__ mean,
                                                         float* __restrict__ std_, float clamp_min, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const uint2* __restrict__ p = row_ld(x + row * Tp);
  float v[HMAXV][4];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < HMAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < Tp / 4) {
      unpack4(p[i], v[k]);
      s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);  // zeros behind T
    }
  }
  const float m = air_wave_sum(s) / (float)T;
  if (lane == 0) mean[row] = m;
  if (std_ != nullptr) {
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < HMAXV; ++k) {
      const int i = lane + 64 * k;
      if (i < Tp / 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * i + e < T) {
            const float d = v[k][e] - m;
            q = fmaf(d, d, q);
          }
      }
    }
    q = air_wave_sum(q) / (float)(T - 1);
    if (lane == 0) std_[row] = sqrtf(fmaxf(q, clamp_min));
  }
}

// dx = bf16(dx + dmean / T + dstd (x - mean) / ((T - 1) std)), zero where x == 0 (the ReLU after layer4,
// ecapa_tdnn.py:173); rowsum[row] = sum_t of the STORED values (over b: the conv bias gradient)
__global__ __launch_bounds__(NT) void h_row_stats_bwd_kernel(const u16* __restrict__ x, int T, int Tp,
                                                             const float* __restrict__ mean, const float* __restrict__ std_,
                                                             const float* __restrict__ dmean, const float* __restrict__ dstd,
                                                             float clamp_min, u16* dx, int accumulate, int relu_mask,
                                                             float* __restrict__ rowsum, size_t rows) {
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
  const uint2* __restrict__ px = row_ld(x + row * Tp);
  uint2* pd = row_st(dx + row * Tp);
  float s = 0.0f;
  for (int i = lane; i < Tp / 4; i += 64) {
    float xv[4], d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    unpack4(px[i], xv);
    if (accumulate) unpack4(pd[i], d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = k0 + k1 * (xv[e] - m) + d[e];
      if (relu_mask && !(xv[e] > 0.0f)) v = 0.0f;
      d[e] = v;
    }
    const uint2 o = pack4_masked(d, 4 * i, T);
    pd[i] = o;
    s += (lo_f(o.x) + hi_f(o.x)) + (lo_f(o.y) + hi_f(o.y));
  }
  if (rowsum != nullptr) {
    s = air_wave_sum(s);
    if (lane == 0) rowsum[row] = s;
  }
}

// ---- attentive statistics pooling (ecapa_tdnn.py:143-185).  a (logits, bf16) is overwritten with
// w = bf16(softmax_T(a)); mu = sum x w, sg = sqrt(clamp(sum x^2 w - mu^2, 1e-4)) with the STORED w.
__global__ __launch_bounds__(NT) void h_asp_fwd_kernel(const u16* __restrict__ x, u16* __restrict__ a, int C, int T, int Tp,
                                                       float* __restrict__ out, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  uint2* __restrict__ pa = row_st(a + row * Tp);
  const uint2* __restrict__ px = row_ld(x + row * Tp);
  float e[HMAXV][4], xv[HMAXV][4];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < HMAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < Tp / 4) {
      unpack4(pa[i], e[k]);
      unpack4(px[i], xv[k]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (4 * i + q >= T) e[k][q] = -INFINITY;
        m = fmaxf(m, e[k][q]);
      }
    }
  }
  m = air_wave_max(m);
  float se = 0.0f;
#pragma unroll
  for (int k = 0; k < HMAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < Tp / 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        e[k][q] = 4 * i + q < T ? expf(e[k][q] - m) : 0.0f;
        se += e[k][q];
      }
    }
  }
  se = air_wave_sum(se);
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int k = 0; k < HMAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < Tp / 4) {
      float w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = e[k][q] / se;
      const uint2 o = pack4_masked(w, 4 * i, T);
      pa[i] = o;
      unpack4(o, w);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s1 = fmaf(xv[k][q], w[q], s1);
        s2 = fmaf(xv[k][q] * xv[k][q], w[q], s2);
      }
    }
  }
  s1 = air_wave_sum(s1);
  s2 = air_wave_sum(s2);
  if (lane == 0) {
    out[b * 2 * C + c] = s1;
    out[b * 2 * C + C + c] = sqrtf(fmaxf(s2 - s1 * s1, 1e-4f));
  }
}

// backward: dx = bf16(d(mu, sg) / dx) (written), w overwritten with bf16(d logits); rowsum = sum_t of the stored d logits
__global__ __launch_bounds__(NT) void h_asp_bwd_kernel(const u16* __restrict__ x, u16* __restrict__ w, int C, int T, int Tp,
                                                       const float* __restrict__ out, const float* __restrict__ dout,
                                                       u16* __restrict__ dx, float* __restrict__ rowsum, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const float mu = out[b * 2 * C + c], sg = out[b * 2 * C + C + c];
  const float dmu = dout[b * 2 * C + c], dsg = dout[b * 2 * C + C + c];
  const float dq = (sg * sg > 1e-4f) ? dsg / (2.0f * sg) : 0.0f;  // the clamp passes no gradient
  const float dm = dmu - 2.0f * mu * dq;
  uint2* __restrict__ pw = row_st(w + row * Tp);
  const uint2* __restrict__ px = row_ld(x + row * Tp);
  uint2* __restrict__ pd = row_st(dx + row * Tp);
  float wv[HMAXV][4], xv[HMAXV][4];
  float dot = 0.0f;
#pragma unroll
  for (int k = 0; k < HMAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < Tp / 4) {
      unpack4(pw[i], wv[k]);
      unpack4(px[i], xv[k]);
#pragma unroll
      for (int q = 0; q < 4; ++q) dot = fmaf(wv[k][q], dm * xv[k][q] + dq * xv[k][q] * xv[k][q], dot);  // zeros behind T
    }
  }
  dot = air_wave_sum(dot);
  float rs = 0.0f;
#pragma unroll
  for (int k = 0; k < HMAXV; ++k) {
    const int i = lane + 64 * k;
    if (i < Tp / 4) {
      float g[4], da[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dwv = dm * xv[k][q] + dq * xv[k][q] * xv[k][q];
        g[q] = dm * wv[k][q] + 2.0f * dq * xv[k][q] * wv[k][q];
        da[q] = wv[k][q] * (dwv - dot);  // softmax backward
      }
      pd[i] = pack4_masked(g, 4 * i, T);
      const uint2 o = pack4_masked(da, 4 * i, T);
      pw[i] = o;
      rs += (lo_f(o.x) + hi_f(o.x)) + (lo_f(o.y) + hi_f(o.y));
    }
  }
  if (rowsum != nullptr) {
    rs = air_wave_sum(rs);
    if (lane == 0) rowsum[row] = rs;
  }
}

// ---- layout changes at the edges of the bf16 region
// y (B, C, T) fp32 dense <- x rows
__global__ __launch_bounds__(NT) void h_to_f32_kernel(const u16* __restrict__ x, size_t xbs, int C, int T, int Tp,
                                                      float* __restrict__ y, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const uint2* __restrict__ p = row_ld(x + b * xbs + c * Tp);
  float* __restrict__ q = y + row * T;
  for (int i = lane; i < (T + 3) / 4; i += 64) {
    float v[4];
    unpack4(p[i], v);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * i + e < T) q[4 * i + e] = v[e];
  }
}
// y rows <- x rows (channel-slice copies: torch.split / torch.cat of ecapa_tdnn.py:71,85 where a view does not do)
__global__ __launch_bounds__(NT) void h_copy_kernel(const u16* __restrict__ x, size_t xbs, int C, int Tp,
                                                    u16* __restrict__ y, size_t ybs, size_t rows) {
  const size_t row = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const size_t b = row / C, c = row - b * C;
  const uint2* __restrict__ p = row_ld(x + b * xbs + c * Tp);
  uint2* __restrict__ q = row_st(y + b * ybs + c * Tp);
  for (int i = lane; i < Tp / 4; i += 64) q[i] = p[i];
}

// ---- first layer (ecapa_tdnn.py:111, K = 5 Conv1d on the fp32 features) as a pointwise GEMM: the fp32 input is
// unfolded once into bf16 rows, row ci K + k (the order of the weight's own (Cin, K) axes, so the (Cout, Cin, K)
// weight IS the GEMM's (Cout, Cin K) matrix) holding x[ci][t + k dil - pad], zeros outside [0, T), in the frames
// T .. Tp - 1 and in the rows Cin K .. R - 1 that pad the contraction to the GEMM's 64-row steps.  One thread =
// 8 frames of one row = one 16-byte store.
__global__ __launch_bounds__(NT) void h_unfold_kernel(const float* __restrict__ x, size_t xbs, int Cin, int T, int Tp,
                                                      int K, int dil, int pad, int R, u16* __restrict__ y, size_t ybs,
                                                      size_t total) {
  const size_t idx = (size_t)blockIdx.x * NT + threadIdx.x;
  if (idx >= total) return;
  const int nch = Tp >> 3;
  const int chunk = (int)(idx % nch);
  const size_t rowi = idx / nch;
  const int row = (int)(rowi % R);
  const size_t b = rowi / R;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.0f;
  if (row < Cin * K) {
    const int ci = row / K, k = row - ci * K;
    const float* __restrict__ src = x + b * xbs + (size_t)ci * T;
    const int t0 = chunk * 8, sh = k * dil - pad;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = t0 + e, ts = t + sh;
      if (t < T && ts >= 0 && ts < T) v[e] = src[ts];
    }
  }
  *reinterpret_cast<uint4*>(y + b * ybs + (size_t)row * Tp + chunk * 8) =
      make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

inline size_t bs_or(size_t bs, int C, int Tp) { return bs ? bs : (size_t)C * Tp; }

}  // namespace

extern "C" {

size_t air_h_bn_ws_bytes(int B, int C) {
  if (B <= 0 || C <= 0) return 0;
  return (size_t)C * h_splits(B, C) * HNACC * sizeof(double);
}

int air_h_bn_stats(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                   float* mean, float* invstd, float* scale, float* shift, void* ws, size_t ws_bytes,
                   air_stream_t stream) {
  return air_h_bn_stats_ex(x, x_bs, B, C, T, Tp, nullptr, 0, gamma, beta, eps, momentum, running_mean, running_var, mean,
                           invstd, scale, shift, ws, ws_bytes, stream);
}

int air_h_bn_stats_ex(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, const void* stats_in,
                      size_t stats_bytes, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                      void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!x || !gamma || !beta || !mean || !invstd || !scale || !shift || !h_shape_ok(B, C, T, Tp)) return AIR_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return AIR_EINVAL;
  if (stats_in != nullptr) {  // the records of air_h_conv1d_tap_ex / air_h_conv1d_pointwise_ex for THIS tensor: no pass over x
    if (Tp % 128 != 0 || stats_bytes % ((size_t)C * 2 * sizeof(float)) != 0) return AIR_EINVAL;
    const int NR = (int)(stats_bytes / ((size_t)C * 2 * sizeof(float)));
    // (32-frame segments of the tap kernel's 128-frame tiles, or 64-frame segments of the GEMM's 256-frame tiles)
    if (NR != B * (Tp / 128) * 4 && !(Tp % 256 == 0 && NR == B * (Tp / 256) * 4)) return AIR_EINVAL;
    hipLaunchKernelGGL(h_bn_finalize_records_kernel, dim3(C), dim3(256), 0, air_stream(stream),
                       reinterpret_cast<const float*>(stats_in), NR, C, (double)B * (double)T, gamma, beta, eps, momentum,
                       running_mean, running_var, mean, invstd, scale, shift);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  if (!ws || ws_bytes < air_h_bn_ws_bytes(B, C)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  const int ns = h_splits(B, C);
  double* partial = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(h_bn_partial_kernel, dim3(C * ns), dim3(NT), 0, st, x, bs_or(x_bs, C, Tp), B, C, T, Tp, ns, partial);
  AIR_CHECK_LAUNCH();
  hipLaunchKernelGGL(h_bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, x, Tp, partial, ns, C,
                     (double)B * (double)T, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                     scale, shift);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_bn_apply(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, const float* scale,
                   const float* shift, unsigned short* y, size_t y_bs, float* rowmean, air_stream_t stream) {
  if (!x || !scale || !shift || !y || !h_shape_ok(B, C, T, Tp)) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_bn_apply_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, bs_or(x_bs, C, Tp), C,
                     T, Tp, scale, shift, y, bs_or(y_bs, C, Tp), rowmean, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_res2_bn_apply(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, const float* scale,
                        const float* shift, unsigned short* y1, size_t y1_bs, const unsigned short* add, size_t add_bs,
                        unsigned short* y2, size_t y2_bs, air_stream_t stream) {
  if (!x || !scale || !shift || !y1 || !h_shape_ok(B, C, T, Tp) || ((add == nullptr) != (y2 == nullptr))) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_res2_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, bs_or(x_bs, C, Tp), C, T,
                     Tp, scale, shift, y1, bs_or(y1_bs, C, Tp), add, bs_or(add_bs, C, Tp), y2, bs_or(y2_bs, C, Tp), rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_se_scale_fwd(const unsigned short* x, size_t x_bs, const float* z, const unsigned short* res, size_t res_bs,
                       int B, int C, int T, int Tp, unsigned short* out, size_t out_bs, air_stream_t stream) {
  if (!x || !z || !res || !out || !h_shape_ok(B, C, T, Tp)) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_se_fwd_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, bs_or(x_bs, C, Tp), z,
                     res, bs_or(res_bs, C, Tp), C, T, Tp, out, bs_or(out_bs, C, Tp), rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_se_scale_bwd(const unsigned short* x, size_t x_bs, const float* z, const unsigned short* dout, size_t dout_bs,
                       int B, int C, int T, int Tp, unsigned short* dx, size_t dx_bs, float* dz, air_stream_t stream) {
  if (!x || !z || !dout || !dx || !dz || !h_shape_ok(B, C, T, Tp)) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_se_bwd_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, bs_or(x_bs, C, Tp), z,
                     dout, bs_or(dout_bs, C, Tp), C, T, Tp, dx, bs_or(dx_bs, C, Tp), dz, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_bn_bwd(const unsigned short* x, size_t x_bs, const unsigned short* dy, size_t dy_bs, const unsigned short* dy2,
                 size_t dy2_bs, const float* dy_rowbias, float rowbias_scale, int B, int C, int T, int Tp,
                 const float* mean, const float* invstd, const float* gamma, int relu_in, unsigned short* dx,
                 size_t dx_bs, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                 air_stream_t stream) {
  return air_h_bn_bwd_ex(x, x_bs, dy, dy_bs, dy2, dy2_bs, dy_rowbias, rowbias_scale, B, C, T, Tp, mean, invstd, gamma,
                         relu_in, dx, dx_bs, dgamma, dbeta, dbias, nullptr, 0, ws, ws_bytes, stream);
}

int air_h_bn_bwd_ex(const unsigned short* x, size_t x_bs, const unsigned short* dy, size_t dy_bs, const unsigned short* dy2,
                    size_t dy2_bs, const float* dy_rowbias, float rowbias_scale, int B, int C, int T, int Tp,
                    const float* mean, const float* invstd, const float* gamma, int relu_in, unsigned short* dx,
                    size_t dx_bs, float* dgamma, float* dbeta, float* dbias, const void* sums_in, size_t sums_bytes,
                    void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!x || !dy || !mean || !invstd || !gamma || !dgamma || !dbeta || !h_shape_ok(B, C, T, Tp)) return AIR_EINVAL;
  if (dbias && !relu_in) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_h_bn_ws_bytes(B, C)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  const int ns = h_splits(B, C);
  double* partial = reinterpret_cast<double*>(ws);
  const double invN = 1.0 / ((double)B * (double)T);
  if (sums_in != nullptr) {
    if (!relu_in || dy_rowbias != nullptr || dy2 == nullptr || Tp % 128 != 0) return AIR_EUNSUPPORTED;
    const int NR = B * (Tp / 128) * 4;
    if (sums_bytes != (size_t)C * NR * 8 * sizeof(float)) return AIR_EINVAL;
    hipLaunchKernelGGL(h_bn_bwd_finalize_records_kernel, dim3(C), dim3(256), 0, st, reinterpret_cast<const float*>(sums_in),
                       NR, C, invN, gamma, invstd, dgamma, dbeta, dbias);
    AIR_CHECK_LAUNCH();
  } else {
    hipLaunchKernelGGL(h_bn_bwd_partial_kernel, dim3(C * ns), dim3(NT), 0, st, x, bs_or(x_bs, C, Tp), dy, bs_or(dy_bs, C, Tp),
                       dy2, bs_or(dy2_bs, C, Tp), dy_rowbias, rowbias_scale, B, C, T, Tp, ns, mean, invstd, dbias ? 1 : 0,
                       partial);
    AIR_CHECK_LAUNCH();
    hipLaunchKernelGGL(h_bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, partial, ns, C, invN, gamma, invstd,
                       dgamma, dbeta, dbias);
    AIR_CHECK_LAUNCH();
  }
  if (dx == nullptr) return AIR_OK;  // (round 6: the apply is the consumer's prologue, air_h_conv1d_tap_pro)
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_bn_bwd_apply_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, st, x, bs_or(x_bs, C, Tp), dy,
                     bs_or(dy_bs, C, Tp), dy2, bs_or(dy2_bs, C, Tp), dy_rowbias, rowbias_scale, C, T, Tp, (float)invN, mean,
                     invstd, gamma, dgamma, dbeta, relu_in, dx, bs_or(dx_bs, C, Tp), rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_row_stats(const unsigned short* x, int B, int C, int T, int Tp, float* mean, float* std_or_null,
                    float clamp_min, air_stream_t stream) {
  if (!x || !mean || !h_shape_ok(B, C, T, Tp) || (std_or_null && T < 2)) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_row_stats_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, T, Tp, mean,
                     std_or_null, clamp_min, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_row_stats_bwd(const unsigned short* x, int B, int C, int T, int Tp, const float* mean, const float* std_,
                        const float* dmean, const float* dstd, float clamp_min, unsigned short* dx, int accumulate,
                        int relu_mask, float* rowsum_or_null, air_stream_t stream) {
  if (!x || !mean || !dx || !h_shape_ok(B, C, T, Tp) || (dstd && !std_)) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_row_stats_bwd_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, T, Tp, mean, std_,
                     dmean, dstd, clamp_min, dx, accumulate, relu_mask, rowsum_or_null, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_asp_fwd(const unsigned short* x, unsigned short* logits_to_w, int B, int C, int T, int Tp, float* out,
                  air_stream_t stream) {
  if (!x || !logits_to_w || !out || !h_shape_ok(B, C, T, Tp)) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_asp_fwd_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, logits_to_w, C, T, Tp,
                     out, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_asp_bwd(const unsigned short* x, unsigned short* w_to_dlogits, int B, int C, int T, int Tp, const float* out,
                  const float* dout, unsigned short* dx, float* rowsum_or_null, air_stream_t stream) {
  if (!x || !w_to_dlogits || !out || !dout || !dx || !h_shape_ok(B, C, T, Tp)) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_asp_bwd_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, w_to_dlogits, C, T, Tp,
                     out, dout, dx, rowsum_or_null, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_to_f32(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, float* y, air_stream_t stream) {
  if (!x || !y || !h_shape_ok(B, C, T, Tp)) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_to_f32_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, bs_or(x_bs, C, Tp), C, T,
                     Tp, y, rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_unfold(const float* x, size_t x_bs, int B, int Cin, int T, int Tp, int K, int dil, int pad, int rows,
                 unsigned short* y, size_t y_bs, air_stream_t stream) {
  if (!x || !y || B <= 0 || Cin <= 0 || T <= 0 || Tp < T || Tp % 8 != 0 || K <= 0 || dil <= 0 || pad < 0 ||
      rows < Cin * K)
    return AIR_EINVAL;
  const size_t total = (size_t)B * rows * (Tp / 8);
  hipLaunchKernelGGL(h_unfold_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, air_stream(stream), x,
                     x_bs ? x_bs : (size_t)Cin * T, Cin, T, Tp, K, dil, pad, rows, y, y_bs ? y_bs : (size_t)rows * Tp,
                     total);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_copy(const unsigned short* x, size_t x_bs, int B, int C, int Tp, unsigned short* y, size_t y_bs,
               air_stream_t stream) {
  if (!x || !y || B <= 0 || C <= 0 || Tp <= 0 || Tp % 4 != 0) return AIR_EINVAL;
  const size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(h_copy_kernel, dim3(h_row_grid(rows)), dim3(NT), 0, air_stream(stream), x, bs_or(x_bs, C, Tp), C, Tp, y,
                     bs_or(y_bs, C, Tp), rows);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

}  // extern "C"
