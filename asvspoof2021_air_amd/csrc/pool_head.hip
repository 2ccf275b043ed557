// Pooling head of the ResNet for gfx950: SelfAttention pooling (resnet.py:23-46)
// and the two nn.Linear layers (resnet.py:143-144, :187-189; also fc6/fc7 of
// ecapa_tdnn.py:148-149).  Tiny tensors ((B,256,94), (B,512)): one workgroup per
// utterance, everything staged in LDS, no intermediate ever reaches HBM.
#include "air_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = air_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = 0.0f;
  for (int w = 0; w < NT / 64; ++w) r += sh[w];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = air_wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < NT / 64; ++w) r = fmaxf(r, sh[w]);
  return r;
}

// x: (B, C, T) channel-major (conv5 output after bn5+ReLU, squeezed); the
// reference works on its (B, T, C) permutation (resnet.py:185) - same numbers.
// dynamic LDS: xs[C][T+1] + w[T] + alpha[T]
// INLDS = false (round 3): maps longer than the LDS holds (T' > 148 at C = 256, i.e. more than ~1190 input
// frames; the reference pools any length, resnet.py:23-46) are read from global memory / L2 in place - the same
// arithmetic in the same order, only slower; the per-frame vectors stay in LDS.
template <bool INLDS>
__global__ __launch_bounds__(NT) void selfatt_fwd_kernel(const float* __restrict__ x, int C, int T,
                                                         const float* __restrict__ att,
                                                         const float* __restrict__ noise,
                                                         float* __restrict__ out,
                                                         float* __restrict__ alpha_save) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[NT / 64];
  const int TP = INLDS ? T + 1 : T;
  const int b = blockIdx.x;
  const float* __restrict__ xb = x + (size_t)b * C * T;
  const float* xs = INLDS ? smem : xb;
  float* ws = INLDS ? smem + (size_t)C * TP : smem;
  // staging with eight loads in flight per thread (a load-then-store loop pays one memory latency per element)
  for (int e0 = threadIdx.x; INLDS && e0 < C * T; e0 += 8 * NT) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = e0 + u * NT < C * T ? xb[e0 + u * NT] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NT;
      if (e < C * T) {
        const int c = e / T, t = e - c * T;
        smem[c * TP + t] = v[u];
      }
    }
  }
  __syncthreads();
  // w_t = <x[:, t], a>   (torch.bmm, resnet.py:26)
  for (int t = threadIdx.x; t < T; t += NT) {
    float s = 0.0f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) s = fmaf(xs[c * TP + t], att[c], s);
    ws[t] = tanhf(s);
  }
  __syncthreads();
  // softmax over T of tanh(w)  (resnet.py:28-33)
  float m = -INFINITY;
  for (int t = threadIdx.x; t < T; t += NT) m = fmaxf(m, ws[t]);
  m = block_max(m, red);
  float se = 0.0f;
  for (int t = threadIdx.x; t < T; t += NT) se += expf(ws[t] - m);
  se = block_sum(se, red);
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += NT) {
    const float al = expf(ws[t] - m) / se;
    ws[t] = al;
    alpha_save[(size_t)b * T + t] = al;
  }
  __syncthreads();
  // weighted mean and unbiased std over T (resnet.py:37-42)
  // (round 6: the noise values of a channel are requested 32 frames at a time - every load of a chunk in flight before
  // the first is consumed - instead of 8: the loop was one memory round trip per 8 frames, twice over; the sums keep
  // their order, so the results are the same bits)
  constexpr int NCH = 32;
  for (int c = threadIdx.x; c < C; c += NT) {
    float avg = 0.0f, zs = 0.0f;
    const float* __restrict__ nc = noise ? noise + (size_t)b * T * C + c : nullptr;
    for (int t0 = 0; t0 < T; t0 += NCH) {
      float nz[NCH];
#pragma unroll
      for (int u = 0; u < NCH; ++u) nz[u] = (nc && t0 + u < T) ? nc[(size_t)(t0 + u) * C] : 0.0f;
#pragma unroll
      for (int u = 0; u < NCH; ++u)
        if (t0 + u < T) {
          const float wv = xs[c * TP + t0 + u] * ws[t0 + u];
          avg += wv;
          zs += noise ? wv + nz[u] : wv;
        }
    }
    const float zm = zs / (float)T;
    float ss = 0.0f;
    for (int t0 = 0; t0 < T; t0 += NCH) {
      float nz[NCH];
#pragma unroll
      for (int u = 0; u < NCH; ++u) nz[u] = (nc && t0 + u < T) ? nc[(size_t)(t0 + u) * C] : 0.0f;
#pragma unroll
      for (int u = 0; u < NCH; ++u)
        if (t0 + u < T) {
          float z = xs[c * TP + t0 + u] * ws[t0 + u];
          if (noise) z += nz[u];
          const float d = z - zm;
          ss = fmaf(d, d, ss);
        }
    }
    out[(size_t)b * 2 * C + c] = avg;
    out[(size_t)b * 2 * C + C + c] = sqrtf(ss / (float)(T - 1));
  }
}

// backward; dynamic LDS: xs[C][T+1] + alpha[T] + dalpha[T] + dw[T] + zmean[C] + coef[C] + davg[C]
//   z[t][c] = x[c][t] alpha_t + noise;  G[t][c] = d_avg[c] + d_std[c] (z - mean_z)/((T-1) std)
//   dalpha_t = sum_c G x;  du = alpha (dalpha - <alpha, dalpha>);  dw = du (1 - tanh^2)
//   dx[c][t] = G alpha_t + dw_t a_c;  datt[c] = sum_t dw_t x[c][t]
template <bool INLDS>
__global__ __launch_bounds__(NT) void selfatt_bwd_kernel(
    const float* __restrict__ x, int C, int T, const float* __restrict__ att,
    const float* __restrict__ noise, const float* __restrict__ alpha,
    const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ dx,
    float* __restrict__ datt_partial) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[NT / 64];
  const int TP = INLDS ? T + 1 : T;
  float* al = INLDS ? smem + (size_t)C * TP : smem;
  float* dal = al + T;
  float* dw = dal + T;
  float* zm = dw + T;
  float* coef = zm + C;
  float* dav = coef + C;
  const int b = blockIdx.x;
  const float* __restrict__ xb = x + (size_t)b * C * T;
  const float* xs = INLDS ? smem : xb;
  float* __restrict__ dxb = dx + (size_t)b * C * T;
  float* dst = INLDS ? smem : dxb;  // dx rows: staged over the x rows, or straight to memory
  const float* __restrict__ nb = noise ? noise + (size_t)b * T * C : nullptr;
  // staging with eight loads in flight per thread (a load-then-store loop pays one memory latency per element)
  for (int e0 = threadIdx.x; INLDS && e0 < C * T; e0 += 8 * NT) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = e0 + u * NT < C * T ? xb[e0 + u * NT] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NT;
      if (e < C * T) {
        const int c = e / T, t = e - c * T;
        smem[c * TP + t] = v[u];
      }
    }
  }
  for (int t = threadIdx.x; t < T; t += NT) al[t] = alpha[(size_t)b * T + t];
  __syncthreads();
  constexpr int NCH = 32;  // noise values requested per chunk (see the forward kernel)
  for (int c = threadIdx.x; c < C; c += NT) {
    float zs = 0.0f;
    for (int t0 = 0; t0 < T; t0 += NCH) {
      float nz[NCH];
#pragma unroll
      for (int u = 0; u < NCH; ++u) nz[u] = (nb && t0 + u < T) ? nb[(size_t)(t0 + u) * C + c] : 0.0f;
#pragma unroll
      for (int u = 0; u < NCH; ++u)
        if (t0 + u < T) {
          float z = xs[c * TP + t0 + u] * al[t0 + u];
          if (nb) z += nz[u];
          zs += z;
        }
    }
    zm[c] = zs / (float)T;
    const float sd = out[(size_t)b * 2 * C + C + c];
    const float dstd = dout[(size_t)b * 2 * C + C + c];
    coef[c] = sd > 0.0f ? dstd / ((float)(T - 1) * sd) : 0.0f;
    dav[c] = dout[(size_t)b * 2 * C + c];
  }
  __syncthreads();
  // a wave per frame, lanes over the channels: the noise rows (B, T, C) are read coalesced
  // (round 6: FOUR frames of a wave per trip - their noise rows requested together; each frame's sums keep their order)
  constexpr int FPT = 4;
  for (int tb = (threadIdx.x >> 6) * FPT; tb < T; tb += (NT / 64) * FPT) {
    float acc[FPT], s[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) acc[f] = s[f] = 0.0f;
    for (int c = threadIdx.x & 63; c < C; c += 64) {
      float nzv[FPT];
#pragma unroll
      for (int f = 0; f < FPT; ++f) nzv[f] = (nb && tb + f < T) ? nb[(size_t)(tb + f) * C + c] : 0.0f;
      const float dvc = dav[c], cfc = coef[c], zmc = zm[c], ac = att[c];
#pragma unroll
      for (int f = 0; f < FPT; ++f)
        if (tb + f < T) {
          const float xv = xs[c * TP + tb + f];
          float z = xv * al[tb + f];
          if (nb) z += nzv[f];
          const float g = dvc + cfc * (z - zmc);
          acc[f] = fmaf(g, xv, acc[f]);
          s[f] = fmaf(xv, ac, s[f]);
        }
    }
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      const float a2 = air_wave_sum(acc[f]), s2 = air_wave_sum(s[f]);
      if ((threadIdx.x & 63) == 0 && tb + f < T) {
        dal[tb + f] = a2;
        const float u = tanhf(s2);
        dw[tb + f] = 1.0f - u * u;  // finished below once <alpha, dalpha> is known
      }
    }
  }
  __syncthreads();
  float dot = 0.0f;
  for (int t = threadIdx.x; t < T; t += NT) dot += al[t] * dal[t];
  dot = block_sum(dot, red);
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += NT) dw[t] = al[t] * (dal[t] - dot) * dw[t];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NT) {
    const float ac = att[c], cf = coef[c], dv = dav[c], zmc = zm[c];
    float da = 0.0f;
    for (int t0 = 0; t0 < T; t0 += NCH) {
      float nz[NCH];
#pragma unroll
      for (int u = 0; u < NCH; ++u) nz[u] = (nb && t0 + u < T) ? nb[(size_t)(t0 + u) * C + c] : 0.0f;
#pragma unroll
      for (int u = 0; u < NCH; ++u)
        if (t0 + u < T) {
          const int t = t0 + u;
          const float xv = xs[c * TP + t];
          float z = xv * al[t];
          if (nb) z += nz[u];
          const float g = dv + cf * (z - zmc);
          dst[c * TP + t] = g * al[t] + dw[t] * ac;  // dx (staged: this thread is the only reader of its x row)
          da = fmaf(dw[t], xv, da);
        }
    }
    datt_partial[(size_t)b * C + c] = da;
  }
  __syncthreads();
  // dx leaves with consecutive lanes on consecutive frames (a thread per channel row wrote 64 rows per store)
  for (int e = threadIdx.x; INLDS && e < C * T; e += NT) {
    const int c = e / T, t = e - c * T;
    dxb[e] = smem[c * TP + t];
  }
}

// y[m][n] = sum_k x[m][k] w[n][k] + b[n].  grid (ceil(N/4), ceil(M/8)): one wave per output n
// for 8 rows m at once, lanes striding k (coalesced weight rows, each read once per 8 rows instead
// of once per row: the SE / fc6 layers of ECAPA at batch 128 were re-streaming W 128 times).
// Per (m, n) the summation order is the same as a lane-strided dot + butterfly.
constexpr int LIN_MB = 8;
__global__ __launch_bounds__(NT) void linear_fwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias, int M, int K,
                                                        int N, int relu, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * (NT / 64) + wave;
  const int m0 = blockIdx.y * LIN_MB;
  if (n >= N) return;
  const float* __restrict__ wr = w + (size_t)n * K;
  float s[LIN_MB];
#pragma unroll
  for (int i = 0; i < LIN_MB; ++i) s[i] = 0.0f;
// (LIN_FWD_UNROLL: A/B builds.  Round 3 reported that `#pragma unroll` here "hung the GPU test run".  Round 4: plain
// `#pragma unroll` compiles to the identical ISA (the per-lane trip count is not unrollable); `unroll 4` / `16` give a
// per-lane remainder loop in front of the unrolled one whose exec-mask logic reads correct, and both builds pass
// test_linear (incl. K = 40 / 100 / 130 / 1100: idle lanes, ragged trip counts) and the ResNet / ECAPA model suites on
// hardware - not reproduced, not a miscompile of this loop.)
#ifdef LIN_FWD_UNROLL
#pragma unroll LIN_FWD_UNROLL
#endif
  for (int k = lane; k < K; k += 64) {
    const float wv = wr[k];
#pragma unroll
    for (int i = 0; i < LIN_MB; ++i) {
      const int m = min(m0 + i, M - 1);
      s[i] = fmaf(x[(size_t)m * K + k], wv, s[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < LIN_MB; ++i) s[i] = air_wave_sum(s[i]);
  if (lane == 0) {
    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < LIN_MB; ++i)
      if (m0 + i < M) {
        const float v = s[i] + bv;
        y[(size_t)(m0 + i) * N + n] = relu ? fmaxf(v, 0.0f) : v;
      }
  }
}

// dx[m][k] = sum_n dy[m][n] w[n][k]: a lane owns one k for 8 rows m, so every weight row is streamed once per
// 8 rows (coalesced along k); the 8 dy rows sit in LDS and are read as broadcasts.
constexpr int LIN_RB = 8;
// grid (ceil(K / 64), ceil(M / 8)): lane = one k of 64, the workgroup's four waves split the sum over n into four
// contiguous quarters (the SE layers have K = 128: a thread per k with the whole sum was 16 workgroups walking
// 512 dependent loads each) and fold them through LDS in quarter order - still a fixed summation order.
__global__ __launch_bounds__(NT) void linear_dx_kernel(const float* __restrict__ dy,
                                                       const float* __restrict__ w, int M, int K, int N,
                                                       float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float ds[];  // [LIN_RB][N], then [4][LIN_RB][64] partial sums
  __shared__ float part[4][LIN_RB][64];
  const int m0 = blockIdx.y * LIN_RB;
  for (int i = threadIdx.x; i < LIN_RB * N; i += NT) {
    const int r = i / N, n = i - r * N;
    ds[i] = m0 + r < M ? dy[(size_t)(m0 + r) * N + n] : 0.0f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int nq = (N + 3) / 4, n_lo = wave * nq, n_hi = min(N, n_lo + nq);
  float s[LIN_RB];
#pragma unroll
  for (int r = 0; r < LIN_RB; ++r) s[r] = 0.0f;
  if (k < K) {
#pragma unroll 16
    for (int n = n_lo; n < n_hi; ++n) {
      const float wv = w[(size_t)n * K + k];
#pragma unroll
      for (int r = 0; r < LIN_RB; ++r) s[r] = fmaf(ds[r * N + n], wv, s[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < LIN_RB; ++r) part[wave][r][lane] = s[r];
  __syncthreads();
  // wave w finishes rows 2 w, 2 w + 1
  if (k < K) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = wave * 2 + rr;
      if (m0 + r < M) dx[(size_t)(m0 + r) * K + k] = ((part[0][r][lane] + part[1][r][lane]) + part[2][r][lane]) + part[3][r][lane];
    }
  }
}

// dw[n][k] = sum_m dy[m][n] x[m][k]; db[n] = sum_m dy[m][n].  grid (ceil(K/64), ceil(N/8)): a lane
// owns one k for 8 outputs n, so x is streamed once per 8 outputs.
__global__ __launch_bounds__(NT) void linear_dw_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ dy, int M, int K,
                                                       int N, float* __restrict__ dw,
                                                       float* __restrict__ db) {
  extern __shared__ __attribute__((aligned(16))) float ds[];  // [M][LIN_RB]
  __shared__ float part[4][LIN_RB][64];
  const int n0 = blockIdx.y * LIN_RB;
  for (int i = threadIdx.x; i < M * LIN_RB; i += NT) {
    const int m = i / LIN_RB, r = i - m * LIN_RB;
    ds[i] = n0 + r < N ? dy[(size_t)m * N + n0 + r] : 0.0f;
  }
  __syncthreads();
  // lane = one k of 64; the four waves take contiguous quarters of the rows m and fold in quarter order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int mq = (M + 3) / 4, m_lo = wave * mq, m_hi = min(M, m_lo + mq);
  float s[LIN_RB];
#pragma unroll
  for (int r = 0; r < LIN_RB; ++r) s[r] = 0.0f;
  if (k < K) {
#pragma unroll 16
    for (int m = m_lo; m < m_hi; ++m) {
      const float xv = x[(size_t)m * K + k];
#pragma unroll
      for (int r = 0; r < LIN_RB; ++r) s[r] = fmaf(ds[m * LIN_RB + r], xv, s[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < LIN_RB; ++r) part[wave][r][lane] = s[r];
  __syncthreads();
  if (k < K) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = wave * 2 + rr;
      if (n0 + r < N)
        dw[(size_t)(n0 + r) * K + k] = ((part[0][r][lane] + part[1][r][lane]) + part[2][r][lane]) + part[3][r][lane];
    }
  }
  if (db != nullptr && blockIdx.x == 0 && threadIdx.x < LIN_RB && n0 + threadIdx.x < N) {
    float t = 0.0f;
    for (int m = 0; m < M; ++m) t += ds[m * LIN_RB + threadIdx.x];
    db[n0 + threadIdx.x] = t;
  }
}

// out[n] = sum_m x[m][n]  (e.g. per-utterance attention-vector partials -> gradient).  Lane = one n of 64; the
// workgroup's four waves take contiguous quarters of the rows (eight loads in flight each) and fold their sums
// in quarter order - a fixed summation order (a thread per column with the whole sum was M dependent loads).
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ x, int M, int N, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int mq = (M + 3) / 4, m_lo = wave * mq, m_hi = min(M, m_lo + mq);
  float s = 0.0f;
  if (n < N) {
#pragma unroll 8
    for (int m = m_lo; m < m_hi; ++m) s += x[(size_t)m * N + n];
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) out[n] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// Philox4x32-10 counter-based generator -> N(0,1) via Box-Muller, scaled.
// Replaces the host-side ``1e-5*torch.randn`` of resnet.py:38 (drawn on the CPU
// and copied every call in the reference) with an on-device draw.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u;
  k[1] += 0xBB67AE85u;
}

__device__ __forceinline__ void randn_body(float* __restrict__ out, size_t n, uint64_t seed, uint64_t offset,
                                           float scale) {
  const size_t quad = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // 4 outputs each
  if (quad * 4 >= n) return;
  const uint64_t ctr = offset + quad;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
  const float inv = 2.3283064365386963e-10f;  // 2^-32
  float z[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)c[2 * h] + 1.0f) * inv;  // (0, 1]
    const float u2 = (float)c[2 * h + 1] * inv;
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * h] = r * cs;
    z[2 * h + 1] = r * sn;
  }
  for (int j = 0; j < 4; ++j)
    if (quad * 4 + j < n) out[quad * 4 + j] = scale * z[j];
}
__global__ void randn_kernel(float* __restrict__ out, size_t n, uint64_t seed, uint64_t offset, float scale) {
  randn_body(out, n, seed, offset, scale);
}

// The same draw with the stream offset read from device memory, and the 1-thread kernel that advances it behind the
// draw: inside a captured hipGraph the offset cannot be a kernel argument (it would be frozen into the graph).
__global__ void randn_ctr_kernel(float* __restrict__ out, size_t n, uint64_t seed,
                                 const unsigned long long* __restrict__ counter, float scale) {
  randn_body(out, n, seed, (uint64_t)*counter, scale);
}
__global__ void counter_add_kernel(unsigned long long* counter, unsigned long long inc) { *counter += inc; }

}  // namespace

extern "C" {

int air_sum_rows(const float* x, int M, int N, float* out, air_stream_t stream) {
  if (!x || !out || M <= 0 || N <= 0) return AIR_EINVAL;
  hipLaunchKernelGGL(sum_rows_kernel, dim3((N + 63) / 64), dim3(256), 0, air_stream(stream), x, M, N, out);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_randn(float* out, size_t n, uint64_t seed, uint64_t offset, float scale,
              air_stream_t stream) {
  if (!out) return AIR_EINVAL;
  if (n == 0) return AIR_OK;
  const size_t quads = (n + 3) / 4;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0,
                     air_stream(stream), out, n, seed, offset, scale);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_randn_ctr(float* out, size_t n, uint64_t seed, uint64_t* counter, float scale, air_stream_t stream) {
  if (!out || !counter || (reinterpret_cast<size_t>(counter) & 7)) return AIR_EINVAL;
  if (n == 0) return AIR_OK;
  const size_t quads = (n + 3) / 4;
  hipLaunchKernelGGL(randn_ctr_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, air_stream(stream), out, n,
                     seed, reinterpret_cast<const unsigned long long*>(counter), scale);
  AIR_CHECK_LAUNCH();
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, air_stream(stream),
                     reinterpret_cast<unsigned long long*>(counter), (unsigned long long)quads);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_selfatt_pool_fwd(const float* x, int B, int C, int T, const float* att_w,
                         const float* noise, float* out, float* alpha_save, air_stream_t stream) {
  if (!x || !att_w || !out || !alpha_save || B <= 0 || C <= 0 || T <= 1) return AIR_EINVAL;
  size_t lds = ((size_t)C * (T + 1) + 2 * (size_t)T) * sizeof(float);
  const bool inlds = lds <= 150 * 1024;
  if (!inlds) lds = 2 * (size_t)T * sizeof(float);
  if (lds > 150 * 1024) return AIR_EUNSUPPORTED;  // (T' > 19200)
  const void* k = inlds ? reinterpret_cast<const void*>(selfatt_fwd_kernel<true>)
                        : reinterpret_cast<const void*>(selfatt_fwd_kernel<false>);
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AIR_ELAUNCH;
  if (inlds)
    hipLaunchKernelGGL(selfatt_fwd_kernel<true>, dim3(B), dim3(NT), lds, air_stream(stream), x, C, T, att_w, noise, out,
                       alpha_save);
  else
    hipLaunchKernelGGL(selfatt_fwd_kernel<false>, dim3(B), dim3(NT), lds, air_stream(stream), x, C, T, att_w, noise, out,
                       alpha_save);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_selfatt_pool_bwd(const float* x, int B, int C, int T, const float* att_w,
                         const float* noise, const float* alpha, const float* out,
                         const float* dout, float* dx, float* datt_partial, air_stream_t stream) {
  if (!x || !att_w || !alpha || !out || !dout || !dx || !datt_partial || B <= 0 || C <= 0 ||
      T <= 1)
    return AIR_EINVAL;
  size_t lds = ((size_t)C * (T + 1) + 3 * (size_t)T + 3 * (size_t)C) * sizeof(float);
  const bool inlds = lds <= 150 * 1024;
  if (!inlds) lds = (3 * (size_t)T + 3 * (size_t)C) * sizeof(float);
  if (lds > 150 * 1024 || (!inlds && x == dx)) return AIR_EUNSUPPORTED;
  const void* k = inlds ? reinterpret_cast<const void*>(selfatt_bwd_kernel<true>)
                        : reinterpret_cast<const void*>(selfatt_bwd_kernel<false>);
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AIR_ELAUNCH;
  if (inlds)
    hipLaunchKernelGGL(selfatt_bwd_kernel<true>, dim3(B), dim3(NT), lds, air_stream(stream), x, C, T, att_w, noise, alpha,
                       out, dout, dx, datt_partial);
  else
    hipLaunchKernelGGL(selfatt_bwd_kernel<false>, dim3(B), dim3(NT), lds, air_stream(stream), x, C, T, att_w, noise,
                       alpha, out, dout, dx, datt_partial);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

static int linear_fwd_impl(const float* x, const float* w, const float* b, int M, int K, int N,
                           int relu, float* y, air_stream_t stream) {
  if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return AIR_EINVAL;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((N + NT / 64 - 1) / (NT / 64), (M + LIN_MB - 1) / LIN_MB), dim3(NT), 0,
                     air_stream(stream), x, w, b, M, K, N, relu, y);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_linear_relu_fwd(const float* x, const float* w, const float* b, int M, int K, int N,
                        float* y, air_stream_t stream) {
  return linear_fwd_impl(x, w, b, M, K, N, 1, y, stream);
}

int air_linear_fwd(const float* x, const float* w, const float* b, int M, int K, int N, float* y,
                   air_stream_t stream) {
  return linear_fwd_impl(x, w, b, M, K, N, 0, y, stream);
}

int air_linear_bwd(const float* x, const float* w, const float* dy, int M, int K, int N, float* dx,
                   float* dw, float* db, air_stream_t stream) {
  if (!x || !w || !dy || M <= 0 || K <= 0 || N <= 0) return AIR_EINVAL;
  if ((size_t)N * LIN_RB * sizeof(float) > 64 * 1024 || (size_t)M * LIN_RB * sizeof(float) > 64 * 1024)
    return AIR_EUNSUPPORTED;
  if (dx != nullptr) {
    hipLaunchKernelGGL(linear_dx_kernel, dim3((K + 63) / 64, (M + LIN_RB - 1) / LIN_RB), dim3(NT),
                       (size_t)N * LIN_RB * sizeof(float), air_stream(stream), dy, w, M, K, N, dx);
    AIR_CHECK_LAUNCH();
  }
  if (dw != nullptr) {
    hipLaunchKernelGGL(linear_dw_kernel, dim3((K + 63) / 64, (N + LIN_RB - 1) / LIN_RB), dim3(NT),
                       (size_t)M * LIN_RB * sizeof(float), air_stream(stream), x, dy, M, K, N, dw, db);
    AIR_CHECK_LAUNCH();
  }
  return AIR_OK;
}

}  // extern "C"
