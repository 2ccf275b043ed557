// BatchNorm (+ReLU) forward/backward for gfx950: HBM-bound reductions and
// elementwise passes over (B, C, S) tensors, S = H*W (BatchNorm2d) or T
// (BatchNorm1d).  Replaces nn.BatchNorm2d + F.relu at resnet.py:55-67,132,142
// and nn.BatchNorm1d in ecapa_tdnn.py.
//
// Layout: a (b, c) plane is S contiguous floats, so every kernel walks whole
// planes with consecutive lanes on consecutive addresses, 16 bytes per lane from the plane's first
// 8-byte boundary on (planes start on 8-byte boundaries when S is even - ECAPA's T = 750 - and on
// alternating 4 / 8-byte boundaries when it is odd - the ResNet's 9 x 375 maps: one head element is
// peeled, kernels instantiated with PEEL); the floats behind the last full quad are one 8-byte access (even
// S) or single accesses (PEEL).
// Reductions are two-stage and deterministic: per-(channel, split) partials in
// fp64, then one finalize block per launch.  No atomics.
#include "air_common.h"

namespace {

constexpr int NT = 256;
constexpr int FLAT_S = 32;  // planes shorter than this: flat-indexed kernels (below)
typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef float f2a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32: the rounding of conv1d_bf16.hip)
__device__ __forceinline__ unsigned bf16_pack2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// floats in front of p's next 8-byte boundary (0 or 1); -1 when the pointers disagree
__device__ __forceinline__ int head8(const void* p) { return (int)((((size_t)p) >> 2) & 1); }
__device__ __forceinline__ int head8(const void* p, const void* q) {
  const int h = head8(p);
  return q == nullptr || head8(q) == h ? h : -1;
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

int splits_for(int B, int C) {
  int s = 2048 / C;
  if (s < 1) s = 1;
  if (s > B) s = B;
  return s;
}

// partial[(c*nsplit + split)*2 + {0,1}] = sum, sum of squares over images [b0,b1)
template <bool PEEL>
__global__ __launch_bounds__(NT) void bn_partial_stats_kernel(const float* __restrict__ x, int B,
                                                              int C, int S, int nsplit,
                                                              double* __restrict__ partial) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x / nsplit, split = blockIdx.x - c * nsplit;
  const int per = (B + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(B, b0 + per);
  // Shifted sums: accumulate (x - K) with K = the channel's first element, so that
  // var = E[(x-K)^2] - E[x-K]^2 does not cancel when |mean| >> std (or when N is tiny).
  const float K = x[(size_t)c * S];
  float s1 = 0.0f, s2 = 0.0f;
  double d1 = 0.0, d2 = 0.0;
  for (int b = b0; b < b1; ++b) {
    const float* __restrict__ p = x + ((size_t)b * C + c) * S;
    const int h = PEEL ? head8(p) : 0;
    if (PEEL || ((S & 1) == 0 && head8(p) == 0)) {
      const int nq = (S - h) >> 2;
      for (int i = threadIdx.x; i < nq; i += NT) {
        f4a8 v = *reinterpret_cast<const f4a8*>(p + h + 4 * i);
        v -= K;
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
      if (threadIdx.x == NT - 1) {
        if (PEEL) {  // head element and the 0 - 3 behind the last quad
          for (int i = 0; i < h; ++i) {
            const float v = p[i] - K;
            s1 += v;
            s2 += v * v;
          }
          for (int i = h + 4 * nq; i < S; ++i) {
            const float v = p[i] - K;
            s1 += v;
            s2 += v * v;
          }
        } else if (S & 2) {
          f2a8 v = *reinterpret_cast<const f2a8*>(p + S - 2);
          v -= K;
          s1 += v[0] + v[1];
          s2 += v[0] * v[0] + v[1] * v[1];
        }
      }
    } else {
      for (int i = threadIdx.x; i < S; i += NT) {
        const float v = p[i] - K;
        s1 += v;
        s2 += v * v;
      }
    }
    d1 += (double)s1;  // flush the fp32 running sums per image to bound their length
    d2 += (double)s2;
    s1 = 0.0f;
    s2 = 0.0f;
  }
  d1 = block_sum_d(d1, sh);
  d2 = block_sum_d(d2, sh);
  if (threadIdx.x == 0) {
    partial[(size_t)blockIdx.x * 2] = d1;
    partial[(size_t)blockIdx.x * 2 + 1] = d2;
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ x, int S,
                                   const double* __restrict__ partial, int nsplit, int C, double N,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift) {
  // one wave per channel (air_wave_ordered_sum_d: the sums of the serial loop over the splits, one round trip)
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  // everything the tail needs is requested in front of the sums: one memory round trip for the kernel
  const float K = x[(size_t)c * S], ga = gamma[c], be = beta[c];
  const float rm = running_mean != nullptr ? running_mean[c] : 0.0f, rv = running_mean != nullptr ? running_var[c] : 0.0f;
  double s12[2];
  air_wave_ordered_sums_d<2>(partial + (size_t)c * nsplit * 2, nsplit, 2, s12);
  const double s1 = s12[0], s2 = s12[1];
  if ((threadIdx.x & 63) != 0) return;
  const double ms = s1 / N;  // mean of the shifted data
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

// Batch statistics from the records a convolution epilogue wrote (air_conv2d_fwd, `stats`): header {G, C, 0, 0},
// then per channel G records {n, K, sum(y - K), sum((y - K)^2)} - the count, mean (K + s1 / n) and M2
// (s2 - s1^2 / n) of one tile group's outputs.  One workgroup per channel: thread t merges the records t, t + 256, ...
// in fp64 relative to the channel's first shift K0 (d = mean_i - K0: A += n d, Q += M2_i + n d^2), the sums meet in
// a fixed order: mean = K0 + A / N, var = Q / N - (A / N)^2 (Chan et al.; in fp64 the last subtraction
// has 29 bits to spare over the fp32 records).  The data itself is not read.  `bad` (device flag) is raised when the
// header does not match the caller's C or the counts do not add up to N - a buffer from another tensor.
__global__ __launch_bounds__(256) void bn_finalize_records_kernel(
    const float* __restrict__ rec, int C, double N, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
  // one workgroup per channel: thread t merges the records t, t + 256, ... (four loads in flight), the four waves'
  // sums meet in LDS in wave order (a wave per channel walked layer1's 4608 records in 72 dependent round trips)
  __shared__ double sh[3][4];
  const int c = blockIdx.x;
  const int tid = threadIdx.x;
  const int G = reinterpret_cast<const int*>(rec)[0], Ch = reinterpret_cast<const int*>(rec)[1];
  const float ga = gamma[c], be = beta[c];
  const float rm = running_mean != nullptr ? running_mean[c] : 0.0f, rv = running_mean != nullptr ? running_var[c] : 0.0f;
  const float4* __restrict__ r = reinterpret_cast<const float4*>(rec + 4) + (size_t)c * G;
  // K0: the shift of the channel's first non-empty record (uniform: every thread scans the same leading records)
  double K0 = 0.0;
  for (int g = 0; g < G; ++g) {
    const float4 v = r[g];
    if (v.x > 0.0f) { K0 = (double)v.y; break; }
  }
  double n = 0.0, A = 0.0, Q = 0.0;
  auto merge = [&](const float4 v) {
    if (v.x > 0.0f) {
      const double ni = (double)v.x, s1 = (double)v.z, s2 = (double)v.w;
      const double d = (double)v.y + s1 / ni - K0;
      n += ni;
      A += ni * d;
      Q += (s2 - s1 * s1 / ni) + ni * d * d;
    }
  };
  int g = tid;
  for (; g + 768 < G; g += 1024) {
    const float4 v0 = r[g], v1 = r[g + 256], v2 = r[g + 512], v3 = r[g + 768];
    merge(v0); merge(v1); merge(v2); merge(v3);
  }
  for (; g < G; g += 256) merge(r[g]);
  n = air_wave_sum_d(n);
  A = air_wave_sum_d(A);
  Q = air_wave_sum_d(Q);
  if ((tid & 63) == 0) { sh[0][tid >> 6] = n; sh[1][tid >> 6] = A; sh[2][tid >> 6] = Q; }
  __syncthreads();
  if (tid != 0) return;
  n = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
  A = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
  Q = ((sh[2][0] + sh[2][1]) + sh[2][2]) + sh[2][3];
  const bool ok = Ch == C && n == N;
  const double ms = A / N;
  double var = Q / N - ms * ms;
  if (var < 0.0) var = 0.0;
  const double m = ms + K0;
  // a mismatched buffer must not train silently on garbage: NaN statistics fail every downstream check loudly
  const float mf = ok ? (float)m : __builtin_nanf("");
  const float is = ok ? (float)(1.0 / sqrt(var + (double)eps)) : __builtin_nanf("");
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

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, int C, float* scale,
                                      float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// grid: (B*C planes, chunks of S).  rowmean (optional, single-chunk planes only): mean over the plane of the
// OUTPUT - the SE squeeze of ecapa_tdnn.py:19 taken while bn3's output is written, instead of a pass that re-reads it.
template <bool PEEL>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const float* __restrict__ x, int C, int S,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu,
                                                      float* __restrict__ y, float* __restrict__ rowmean,
                                                      unsigned short* __restrict__ bf, int Tp) {
  __shared__ float wsum[NT / 64];
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float sc = scale[c], sh = shift[c];
  const float* __restrict__ p = x + (size_t)plane * S;
  float* __restrict__ q = y + (size_t)plane * S;
  // thread qi owns the quad at h + 4 qi (h = floats in front of the first 8-byte boundary: 0 unless PEEL);
  // PEEL: thread 0 also owns the head, the thread of the last, incomplete quad walks it element by element
  const int qi = blockIdx.y * NT + threadIdx.x;
  const int h = PEEL ? head8(p, q) : (head8(p) | head8(q) ? -1 : 0);
  const int i0 = h < 0 ? 4 * qi : h + 4 * qi;
  float tsum = 0.0f;
  // bf (optional): y also as bf16, [plane][Tp] (a later weight gradient's operand, conv1d_bf16.hip)
  unsigned short* __restrict__ pbf = bf ? bf + (size_t)plane * Tp : nullptr;
  auto scalar = [&](int lo, int hi) {
    for (int i = lo; i < hi; ++i) {
      float v = p[i] * sc + sh;
      if (relu) v = fmaxf(v, 0.f);
      q[i] = v;
      if (pbf) pbf[i] = (unsigned short)(bf16_pack2(v, 0.0f) & 0xffffu);
      tsum += v;
    }
  };
  if (h >= 0 && i0 + 3 < S) {
    if (PEEL && qi == 0) scalar(0, h);
    f4a8 v = *reinterpret_cast<const f4a8*>(p + i0);
    v = v * sc + sh;
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *reinterpret_cast<f4a8*>(q + i0) = v;
    if (pbf) {
#pragma unroll
      for (int e = 0; e < 4; ++e) pbf[i0 + e] = (unsigned short)(bf16_pack2(v[e], 0.0f) & 0xffffu);
    }
    tsum += (v[0] + v[1]) + (v[2] + v[3]);
  } else if (!PEEL && h >= 0 && i0 + 2 == S) {
    f2a8 v = *reinterpret_cast<const f2a8*>(p + i0);
    v = v * sc + sh;
    if (relu) {
      v[0] = fmaxf(v[0], 0.f);
      v[1] = fmaxf(v[1], 0.f);
    }
    *reinterpret_cast<f2a8*>(q + i0) = v;
    if (pbf) *reinterpret_cast<unsigned*>(pbf + i0) = bf16_pack2(v[0], v[1]);
    tsum += v[0] + v[1];
  } else {
    scalar((PEEL && h > 0 && qi == 0) ? 0 : i0, min(S, i0 + 4));
  }
  if (rowmean != nullptr) {  // wave sums folded in wave order: a fixed summation order
    tsum = air_wave_sum(tsum);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = tsum;
    __syncthreads();
    if (threadIdx.x == 0) rowmean[plane] = (((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]) / (float)S;
  }
}

// backward stage 1: partial[(c*nsplit+split)*NACC + {0,1}] = sum g, sum g*xhat, g = (dy + rowbias)*[y>0].
// With want_bias (conv -> ReLU -> BN layers, ecapa_tdnn.py:67-69) three more sums over the positions
// where the BN input is positive: sum g, count, sum xhat - from them the finalize kernel gets the
// gradient of the conv BIAS, sum_{b,s} dx, in closed form, so no pass over dx is needed for it:
//   dx = [x > 0] * gamma*invstd * (g - dbeta/N - xhat * dgamma/N).
constexpr int NACC = 6;
template <bool PEEL>
__global__ __launch_bounds__(NT) void bn_bwd_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, size_t dy_bs, const float* __restrict__ dy2,
    size_t dy2_bs, const float* __restrict__ rowbias, float rb_scale, int B, int C, int S, int nsplit,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, int want_bias,
    double* __restrict__ partial) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x / nsplit, split = blockIdx.x - c * nsplit;
  const int per = (B + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(B, b0 + per);
  const float mu = mean[c], is = invstd[c];
  const float sc = gamma[c] * is;
  const float shf = beta[c] - mu * sc;
  double d1 = 0.0, d2 = 0.0, d3 = 0.0, d4 = 0.0, d5 = 0.0;
  for (int b = b0; b < b1; ++b) {
    const float* __restrict__ px = x + ((size_t)b * C + c) * S;
    const float* __restrict__ pg = dy + (size_t)b * dy_bs + (size_t)c * S;
    const float* __restrict__ pg2 = dy2 ? dy2 + (size_t)b * dy2_bs + (size_t)c * S : nullptr;
    const float rb = rowbias ? rowbias[(size_t)b * C + c] * rb_scale : 0.0f;
    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f, s5 = 0.0f;
    auto one = [&](float xv, float g) {
      g += rb;
      if (relu && !(xv * sc + shf > 0.0f)) g = 0.0f;
      const float xh = (xv - mu) * is;
      s1 += g;
      s2 += g * xh;
      if (want_bias && xv > 0.0f) {
        s3 += g;
        s4 += 1.0f;
        s5 += xh;
      }
    };
    int h = head8(px, pg) == head8(px, pg2) ? head8(px, pg) : -1;
    if (!PEEL && (h != 0 || (S & 1))) h = -1;
    if (h >= 0) {
      const int nq = (S - h) >> 2;
      for (int i = threadIdx.x; i < nq; i += NT) {
        const f4a8 xv = *reinterpret_cast<const f4a8*>(px + h + 4 * i);
        f4a8 g = *reinterpret_cast<const f4a8*>(pg + h + 4 * i);
        if (pg2) g += *reinterpret_cast<const f4a8*>(pg2 + h + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) one(xv[e], g[e]);
      }
      if (threadIdx.x == NT - 1) {
        if (PEEL) {  // head element and the 0 - 3 behind the last quad
          for (int i = 0; i < h; ++i) one(px[i], pg[i] + (pg2 ? pg2[i] : 0.0f));
          for (int i = h + 4 * nq; i < S; ++i) one(px[i], pg[i] + (pg2 ? pg2[i] : 0.0f));
        } else if (S & 2) {
          const f2a8 xv = *reinterpret_cast<const f2a8*>(px + S - 2);
          f2a8 g = *reinterpret_cast<const f2a8*>(pg + S - 2);
          if (pg2) g += *reinterpret_cast<const f2a8*>(pg2 + S - 2);
          one(xv[0], g[0]);
          one(xv[1], g[1]);
        }
      }
    } else {
      for (int i = threadIdx.x; i < S; i += NT) one(px[i], pg[i] + (pg2 ? pg2[i] : 0.0f));
    }
    d1 += (double)s1;
    d2 += (double)s2;
    d3 += (double)s3;
    d4 += (double)s4;
    d5 += (double)s5;
  }
  d1 = block_sum_d(d1, sh);
  d2 = block_sum_d(d2, sh);
  if (want_bias) {
    d3 = block_sum_d(d3, sh);
    d4 = block_sum_d(d4, sh);
    d5 = block_sum_d(d5, sh);
  }
  if (threadIdx.x == 0) {
    double* o = partial + (size_t)blockIdx.x * NACC;
    o[0] = d1; o[1] = d2; o[2] = d3; o[3] = d4; o[4] = d5;
  }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nsplit, int C, double invN,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ dbias) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per channel, as bn_finalize_kernel
  if (c >= C) return;
  const double* o = partial + (size_t)c * nsplit * NACC;
  const float ga = gamma[c], is = invstd[c];  // requested together with the partial sums
  double s1, s2, s3 = 0.0, s4 = 0.0, s5 = 0.0;
  if (dbias) {
    double t[5];
    air_wave_ordered_sums_d<5>(o, nsplit, NACC, t);
    s1 = t[0]; s2 = t[1]; s3 = t[2]; s4 = t[3]; s5 = t[4];
  } else {
    double t[2];
    air_wave_ordered_sums_d<2>(o, nsplit, NACC, t);
    s1 = t[0]; s2 = t[1];
  }
  if ((threadIdx.x & 63) != 0) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  if (dbias) {
    // the apply kernel uses the fp32 dbeta / dgamma just written: mirror its constants
    const double k1 = (double)(float)s1 * invN, k2 = (double)(float)s2 * invN;
    dbias[c] = (float)((double)ga * (double)is * (s3 - k1 * s4 - k2 * s5));
  }
}

// The two backward sums from the records a data-gradient epilogue wrote (air_conv2d_dgrad_bn): header {G, C, 0, 0},
// then per channel G records {sum g, sum g * xhat, n, 0}.  One workgroup per channel, fp64, fixed order (as
// bn_finalize_records_kernel).  A buffer whose header or element count does not match gives NaN gradients.
__global__ __launch_bounds__(256) void bn_bwd_finalize_records_kernel(const float* __restrict__ rec, int C, double N,
                                                                      float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta) {
  __shared__ double sh[3][4];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int G = reinterpret_cast<const int*>(rec)[0], Ch = reinterpret_cast<const int*>(rec)[1];
  const float4* __restrict__ r = reinterpret_cast<const float4*>(rec + 4) + (size_t)c * G;
  double s1 = 0.0, s2 = 0.0, n = 0.0;
  int g = tid;
  for (; g + 768 < G; g += 1024) {
    const float4 v0 = r[g], v1 = r[g + 256], v2 = r[g + 512], v3 = r[g + 768];
    s1 += (double)v0.x; s2 += (double)v0.y; n += (double)v0.z;
    s1 += (double)v1.x; s2 += (double)v1.y; n += (double)v1.z;
    s1 += (double)v2.x; s2 += (double)v2.y; n += (double)v2.z;
    s1 += (double)v3.x; s2 += (double)v3.y; n += (double)v3.z;
  }
  for (; g < G; g += 256) {
    const float4 v = r[g];
    s1 += (double)v.x; s2 += (double)v.y; n += (double)v.z;
  }
  s1 = air_wave_sum_d(s1);
  s2 = air_wave_sum_d(s2);
  n = air_wave_sum_d(n);
  if ((tid & 63) == 0) { sh[0][tid >> 6] = s1; sh[1][tid >> 6] = s2; sh[2][tid >> 6] = n; }
  __syncthreads();
  if (tid != 0) return;
  s1 = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
  s2 = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
  n = ((sh[2][0] + sh[2][1]) + sh[2][2]) + sh[2][3];
  const bool ok = Ch == C && n == N;
  dbeta[c] = ok ? (float)s1 : __builtin_nanf("");
  dgamma[c] = ok ? (float)s2 : __builtin_nanf("");
}

// backward stage 2: dx = gamma*invstd*(g - dbeta/N - xhat*dgamma/N)  (+= if accum)
// (dy and dx may alias: the in-place gradient joins of resnet.py / ecapa_tdnn.py)
template <bool PEEL>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(
    const float* __restrict__ x, const float* dy, size_t dy_bs, const float* dy2, size_t dy2_bs,
    const float* __restrict__ rowbias, float rb_scale, int C, int S, float invN,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta, int relu, int accum,
    int relu_in, float* dx, unsigned short* __restrict__ bf, int Tp) {
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float mu = mean[c], is = invstd[c];
  const float sc = gamma[c] * is;
  const float shf = beta[c] - mu * sc;
  const float k1 = dbeta[c] * invN, k2 = dgamma[c] * invN;
  const float rb = rowbias ? rowbias[plane] * rb_scale : 0.0f;
  const size_t base = (size_t)plane * S;
  const int bb = plane / C;
  const float* pdy = dy + (size_t)bb * dy_bs + (size_t)c * S;               // may alias dx (in-place joins)
  const float* pdy2 = dy2 ? dy2 + (size_t)bb * dy2_bs + (size_t)c * S : nullptr;
  auto one = [&](float xv, float g, float old) -> float {
    g += rb;
    if (relu && !(xv * sc + shf > 0.0f)) g = 0.0f;
    const float xh = (xv - mu) * is;
    float r = sc * (g - k1 - xh * k2);
    if (relu_in && !(xv > 0.0f)) r = 0.0f;  // x = relu(c): no gradient where the ReLU clipped
    return accum ? r + old : r;
  };
  // quads from the plane's first 8-byte boundary on, as in bn_apply_kernel
  const int qi = blockIdx.y * NT + threadIdx.x;
  int h = head8(x + base, pdy);
  if (h != head8(dx + base, pdy2)) h = -1;
  if (!PEEL && h != 0) h = -1;
  const int i0 = h < 0 ? 4 * qi : h + 4 * qi;
  // bf: optional bf16 copy of dx, [plane][Tp] (the weight-gradient GEMM's operand layout, conv1d_bf16.hip)
  unsigned short* __restrict__ pbf = bf ? bf + (size_t)plane * Tp : nullptr;
  auto scalar = [&](int lo, int hi) {
    for (int i = lo; i < hi; ++i) {
      const float r = one(x[base + i], pdy[i] + (pdy2 ? pdy2[i] : 0.0f), accum ? dx[base + i] : 0.0f);
      dx[base + i] = r;
      if (pbf) pbf[i] = (unsigned short)(bf16_pack2(r, 0.0f) & 0xffffu);
    }
  };
  if (h >= 0 && i0 + 3 < S) {
    if (PEEL && qi == 0) scalar(0, h);
    const f4a8 xv = *reinterpret_cast<const f4a8*>(x + base + i0);
    f4a8 gv = *reinterpret_cast<const f4a8*>(pdy + i0);
    if (pdy2) gv += *reinterpret_cast<const f4a8*>(pdy2 + i0);
    f4a8 ov = {0.f, 0.f, 0.f, 0.f};
    if (accum) ov = *reinterpret_cast<const f4a8*>(dx + base + i0);
    f4a8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = one(xv[e], gv[e], ov[e]);
    *reinterpret_cast<f4a8*>(dx + base + i0) = r;
    if (pbf) {
      if (!PEEL) {
        *reinterpret_cast<uint2*>(pbf + i0) = make_uint2(bf16_pack2(r[0], r[1]), bf16_pack2(r[2], r[3]));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) pbf[i0 + e] = (unsigned short)(bf16_pack2(r[e], 0.0f) & 0xffffu);
      }
    }
  } else if (!PEEL && h >= 0 && i0 + 2 == S) {
    const f2a8 xv = *reinterpret_cast<const f2a8*>(x + base + i0);
    f2a8 gv = *reinterpret_cast<const f2a8*>(pdy + i0);
    if (pdy2) gv += *reinterpret_cast<const f2a8*>(pdy2 + i0);
    f2a8 ov = {0.f, 0.f};
    if (accum) ov = *reinterpret_cast<const f2a8*>(dx + base + i0);
    f2a8 r;
    r[0] = one(xv[0], gv[0], ov[0]);
    r[1] = one(xv[1], gv[1], ov[1]);
    *reinterpret_cast<f2a8*>(dx + base + i0) = r;
    if (pbf) *reinterpret_cast<unsigned*>(pbf + i0) = bf16_pack2(r[0], r[1]);
  } else {
    scalar((PEEL && h > 0 && qi == 0) ? 0 : i0, min(S, i0 + 4));
  }
}

// ---------------------------------------------------------------------------------------------
// Short planes (S < 32: BatchNorm1d over (B, C) features - the SE bottleneck, bn5 on the pooled
// 3072 statistics, bn7 - where S = 1): a workgroup per plane would be a workgroup per ELEMENT
// (393 k of them for bn5).  These variants index the (b, s) positions of a channel, or all elements,
// flat; same sums, same order of the fp64 folds.
__global__ __launch_bounds__(NT) void bn_partial_stats_flat_kernel(const float* __restrict__ x, int B, int C, int S,
                                                                   int nsplit, double* __restrict__ partial) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x / nsplit, split = blockIdx.x - c * nsplit;
  const int per = (B + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(B, b0 + per);
  const float K = x[(size_t)c * S];
  double d1 = 0.0, d2 = 0.0;
  for (int i = threadIdx.x; i < (b1 - b0) * S; i += NT) {
    const int b = b0 + i / S, sidx = i % S;
    const float v = x[((size_t)b * C + c) * S + sidx] - K;
    d1 += (double)v;
    d2 += (double)(v * v);
  }
  d1 = block_sum_d(d1, sh);
  d2 = block_sum_d(d2, sh);
  if (threadIdx.x == 0) {
    partial[(size_t)blockIdx.x * 2] = d1;
    partial[(size_t)blockIdx.x * 2 + 1] = d2;
  }
}

__global__ __launch_bounds__(NT) void bn_apply_flat_kernel(const float* __restrict__ x, int C, int S, size_t n,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu,
                                                           float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  const int c = (int)((i / S) % C);
  float v = x[i] * scale[c] + shift[c];
  if (relu) v = fmaxf(v, 0.f);
  y[i] = v;
}

__global__ __launch_bounds__(NT) void bn_bwd_partial_flat_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, size_t dy_bs, const float* __restrict__ dy2,
    size_t dy2_bs, const float* __restrict__ rowbias, float rb_scale, int B, int C, int S, int nsplit,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, int want_bias,
    double* __restrict__ partial) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x / nsplit, split = blockIdx.x - c * nsplit;
  const int per = (B + nsplit - 1) / nsplit;
  const int b0 = split * per, b1 = min(B, b0 + per);
  const float mu = mean[c], is = invstd[c];
  const float sc = gamma[c] * is;
  const float shf = beta[c] - mu * sc;
  double d1 = 0.0, d2 = 0.0, d3 = 0.0, d4 = 0.0, d5 = 0.0;
  for (int i = threadIdx.x; i < (b1 - b0) * S; i += NT) {
    const int b = b0 + i / S, sidx = i % S;
    const float xv = x[((size_t)b * C + c) * S + sidx];
    float g = dy[(size_t)b * dy_bs + (size_t)c * S + sidx];
    if (dy2) g += dy2[(size_t)b * dy2_bs + (size_t)c * S + sidx];
    if (rowbias) g += rowbias[(size_t)b * C + c] * rb_scale;
    if (relu && !(xv * sc + shf > 0.0f)) g = 0.0f;
    const float xh = (xv - mu) * is;
    d1 += (double)g;
    d2 += (double)(g * xh);
    if (want_bias && xv > 0.0f) {
      d3 += (double)g;
      d4 += 1.0;
      d5 += (double)xh;
    }
  }
  d1 = block_sum_d(d1, sh);
  d2 = block_sum_d(d2, sh);
  if (want_bias) {
    d3 = block_sum_d(d3, sh);
    d4 = block_sum_d(d4, sh);
    d5 = block_sum_d(d5, sh);
  }
  if (threadIdx.x == 0) {
    double* o = partial + (size_t)blockIdx.x * NACC;
    o[0] = d1; o[1] = d2; o[2] = d3; o[3] = d4; o[4] = d5;
  }
}

__global__ __launch_bounds__(NT) void bn_bwd_apply_flat_kernel(
    const float* __restrict__ x, const float* dy, size_t dy_bs, const float* dy2, size_t dy2_bs,
    const float* __restrict__ rowbias, float rb_scale, int C, int S, size_t n, float invN,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta, int relu, int accum,
    int relu_in, float* dx) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  const size_t plane = i / S;
  const int sidx = (int)(i - plane * S), c = (int)(plane % C);
  const size_t b = plane / C;
  const float mu = mean[c], is = invstd[c];
  const float sc = gamma[c] * is;
  const float shf = beta[c] - mu * sc;
  const float xv = x[i];
  float g = dy[b * dy_bs + (size_t)c * S + sidx];
  if (dy2) g += dy2[b * dy2_bs + (size_t)c * S + sidx];
  if (rowbias) g += rowbias[plane] * rb_scale;
  if (relu && !(xv * sc + shf > 0.0f)) g = 0.0f;
  const float xh = (xv - mu) * is;
  float r = sc * (g - dbeta[c] * invN - xh * (dgamma[c] * invN));
  if (relu_in && !(xv > 0.0f)) r = 0.0f;
  dx[i] = accum ? r + dx[i] : r;
}

__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] += x[i];
}

}  // namespace

extern "C" {

size_t air_bn_ws_bytes(int B, int C, int S) {
  if (B <= 0 || C <= 0 || S <= 0) return 0;
  return (size_t)C * splits_for(B, C) * NACC * sizeof(double);
}

int air_bn_stats(const float* x, int B, int C, int S, const double* stats_in, const float* gamma,
                 const float* beta, float eps, float momentum, float* running_mean,
                 float* running_var, float* mean, float* invstd, float* scale, float* shift,
                 void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!x || !gamma || !beta || !mean || !invstd || !scale || !shift || B <= 0 || C <= 0 || S <= 0)
    return AIR_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return AIR_EINVAL;
  hipStream_t st = air_stream(stream);
  if (stats_in != nullptr) {  // records from the producing convolution's epilogue: no pass over x
    if (reinterpret_cast<size_t>(stats_in) & 15) return AIR_EINVAL;
    hipLaunchKernelGGL(bn_finalize_records_kernel, dim3(C), dim3(256), 0, st,
                       reinterpret_cast<const float*>(stats_in), C, (double)B * (double)S, gamma, beta, eps, momentum,
                       running_mean, running_var, mean, invstd, scale, shift);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  if (!ws || ws_bytes < air_bn_ws_bytes(B, C, S)) return AIR_EWORKSPACE;
  const int nsplit = splits_for(B, C);
  double* partial = reinterpret_cast<double*>(ws);
  if (S < FLAT_S)
    hipLaunchKernelGGL(bn_partial_stats_flat_kernel, dim3(C * nsplit), dim3(NT), 0, st, x, B, C, S, nsplit, partial);
  else if (S & 1)
    hipLaunchKernelGGL(bn_partial_stats_kernel<true>, dim3(C * nsplit), dim3(NT), 0, st, x, B, C, S, nsplit, partial);
  else
    hipLaunchKernelGGL(bn_partial_stats_kernel<false>, dim3(C * nsplit), dim3(NT), 0, st, x, B, C, S, nsplit, partial);
  AIR_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, x, S, partial, nsplit, C,
                     (double)B * (double)S, gamma, beta, eps, momentum, running_mean, running_var,
                     mean, invstd, scale, shift);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* scale, float* shift,
                       air_stream_t stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0)
    return AIR_EINVAL;
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, air_stream(stream),
                     gamma, beta, running_mean, running_var, eps, C, scale, shift);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_bn_apply_ex(const float* x, int B, int C, int S, const float* scale, const float* shift,
                    int relu, float* y, float* rowmean, unsigned short* y_bf16, int y_bf16_tp, air_stream_t stream) {
  if (!x || !scale || !shift || !y || B <= 0 || C <= 0 || S <= 0) return AIR_EINVAL;
  dim3 grid(B * C, (S + NT * 4 - 1) / (NT * 4));
  if (rowmean && (S < FLAT_S || grid.y != 1)) return AIR_EUNSUPPORTED;
  if (y_bf16 && (S < FLAT_S || y_bf16_tp < S || (y_bf16_tp & 1))) return AIR_EINVAL;
  const size_t n = (size_t)B * C * S;
  if (S < FLAT_S)
    hipLaunchKernelGGL(bn_apply_flat_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, air_stream(stream), x, C, S,
                       n, scale, shift, relu, y);
  else if (S & 1)
    hipLaunchKernelGGL(bn_apply_kernel<true>, grid, dim3(NT), 0, air_stream(stream), x, C, S, scale, shift, relu, y, rowmean,
                       y_bf16, y_bf16_tp);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, grid, dim3(NT), 0, air_stream(stream), x, C, S, scale, shift, relu, y, rowmean,
                       y_bf16, y_bf16_tp);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_bn_apply(const float* x, int B, int C, int S, const float* scale, const float* shift,
                 int relu, float* y, air_stream_t stream) {
  return air_bn_apply_ex(x, B, C, S, scale, shift, relu, y, nullptr, nullptr, 0, stream);
}

int air_bn_bwd_ex2(const float* x, const float* dy, size_t dy_bstride, const float* dy2, size_t dy2_bstride,
                   const float* dy_rowbias, float rowbias_scale, int B, int C, int S, const float* mean,
                   const float* invstd, const float* gamma, const float* beta, int relu, float* dx, int dx_accum,
                   float* dgamma, float* dbeta, float* dbias, unsigned short* dx_bf16, int dx_bf16_tp, void* ws,
                   size_t ws_bytes, air_stream_t stream) {
  return air_bn_bwd_ex3(x, dy, dy_bstride, dy2, dy2_bstride, dy_rowbias, rowbias_scale, B, C, S, mean, invstd, gamma, beta,
                        relu, dx, dx_accum, dgamma, dbeta, dbias, dx_bf16, dx_bf16_tp, nullptr, ws, ws_bytes, stream);
}

int air_bn_bwd_ex3(const float* x, const float* dy, size_t dy_bstride, const float* dy2, size_t dy2_bstride,
                   const float* dy_rowbias, float rowbias_scale, int B, int C, int S, const float* mean,
                   const float* invstd, const float* gamma, const float* beta, int relu, float* dx, int dx_accum,
                   float* dgamma, float* dbeta, float* dbias, unsigned short* dx_bf16, int dx_bf16_tp,
                   const void* sums_in, void* ws, size_t ws_bytes, air_stream_t stream) {
  // sums_in: the records of air_conv2d_dgrad_bn for THIS (x, dy): plain relu(batchnorm) only - one dense gradient,
  // no second gradient, no row bias, no conv-bias gradient (those change the sums the producer took)
  if (sums_in && (dy2 || dy_rowbias || dbias || !(relu & 1) || (relu & 2) || dy_bstride != 0 ||
                  (reinterpret_cast<size_t>(sums_in) & 15)))
    return AIR_EINVAL;
  if (dx_bf16 && (S < FLAT_S || dx_bf16_tp < S || (dx_bf16_tp & 3) || (reinterpret_cast<size_t>(dx_bf16) & 7)))
    return AIR_EINVAL;
  if (!x || !dy || !mean || !invstd || !gamma || !beta || !dx || !dgamma || !dbeta || B <= 0 ||
      C <= 0 || S <= 0)
    return AIR_EINVAL;
  if (dbias && !(relu & 2)) return AIR_EINVAL;  // the bias gradient is defined for conv -> ReLU -> BN layers
  if (!ws || ws_bytes < air_bn_ws_bytes(B, C, S)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  const int nsplit = splits_for(B, C);
  double* partial = reinterpret_cast<double*>(ws);
  const double invN = 1.0 / ((double)B * (double)S);
  const size_t dense = (size_t)C * S;
  const size_t dbs = dy_bstride ? dy_bstride : dense, d2bs = dy2_bstride ? dy2_bstride : dense;
  if (sums_in != nullptr) {
    hipLaunchKernelGGL(bn_bwd_finalize_records_kernel, dim3(C), dim3(256), 0, st, reinterpret_cast<const float*>(sums_in), C,
                       (double)B * (double)S, dgamma, dbeta);
    AIR_CHECK_LAUNCH();
  } else if (S < FLAT_S)
    hipLaunchKernelGGL(bn_bwd_partial_flat_kernel, dim3(C * nsplit), dim3(NT), 0, st, x, dy, dbs, dy2, d2bs, dy_rowbias,
                       rowbias_scale, B, C, S, nsplit, mean, invstd, gamma, beta, relu & 1, dbias ? 1 : 0, partial);
  else if (S & 1)
    hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, dim3(C * nsplit), dim3(NT), 0, st, x, dy, dbs, dy2, d2bs, dy_rowbias,
                       rowbias_scale, B, C, S, nsplit, mean, invstd, gamma, beta, relu & 1, dbias ? 1 : 0, partial);
  else
    hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(C * nsplit), dim3(NT), 0, st, x, dy, dbs, dy2, d2bs, dy_rowbias,
                       rowbias_scale, B, C, S, nsplit, mean, invstd, gamma, beta, relu & 1, dbias ? 1 : 0, partial);
  AIR_CHECK_LAUNCH();
  if (sums_in == nullptr) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, partial, nsplit, C, (double)(float)invN,
                       gamma, invstd, dgamma, dbeta, dbias);
    AIR_CHECK_LAUNCH();
  }
  dim3 grid(B * C, (S + NT * 4 - 1) / (NT * 4));
  const size_t n = (size_t)B * C * S;
  if (S < FLAT_S)
    hipLaunchKernelGGL(bn_bwd_apply_flat_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, x, dy, dbs, dy2, d2bs,
                       dy_rowbias, rowbias_scale, C, S, n, (float)invN, mean, invstd, gamma, beta, dgamma, dbeta, relu & 1,
                       dx_accum, (relu >> 1) & 1, dx);
  else if (S & 1)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(NT), 0, st, x, dy, dbs, dy2, d2bs, dy_rowbias, rowbias_scale,
                       C, S, (float)invN, mean, invstd, gamma, beta, dgamma, dbeta, relu & 1, dx_accum, (relu >> 1) & 1, dx, dx_bf16, dx_bf16_tp);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(NT), 0, st, x, dy, dbs, dy2, d2bs, dy_rowbias, rowbias_scale,
                       C, S, (float)invN, mean, invstd, gamma, beta, dgamma, dbeta, relu & 1, dx_accum, (relu >> 1) & 1, dx,
                       dx_bf16, dx_bf16_tp);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_bn_bwd_ex(const float* x, const float* dy, size_t dy_bstride, const float* dy2, size_t dy2_bstride,
                  const float* dy_rowbias, float rowbias_scale, int B, int C, int S, const float* mean,
                  const float* invstd, const float* gamma, const float* beta, int relu, float* dx, int dx_accum,
                  float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes, air_stream_t stream) {
  return air_bn_bwd_ex2(x, dy, dy_bstride, dy2, dy2_bstride, dy_rowbias, rowbias_scale, B, C, S, mean, invstd, gamma, beta,
                        relu, dx, dx_accum, dgamma, dbeta, dbias, nullptr, 0, ws, ws_bytes, stream);
}

int air_bn_bwd(const float* x, const float* dy, int B, int C, int S, const float* mean,
               const float* invstd, const float* gamma, const float* beta, int relu, float* dx,
               int dx_accum, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
               air_stream_t stream) {
  return air_bn_bwd_ex(x, dy, 0, nullptr, 0, nullptr, 0.0f, B, C, S, mean, invstd, gamma, beta, relu, dx, dx_accum,
                       dgamma, dbeta, nullptr, ws, ws_bytes, stream);
}

int air_add_inplace(float* y, const float* x, size_t n, air_stream_t stream) {
  if (!y || !x) return AIR_EINVAL;
  if (n == 0) return AIR_OK;
  size_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)g), dim3(256), 0, air_stream(stream), y, x,
                     n);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

}  // extern "C"
