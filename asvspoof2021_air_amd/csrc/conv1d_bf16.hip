// bf16-compute pointwise (K = 1) Conv1d of ECAPA-TDNN on the gfx950 matrix cores
// (BASELINE.json configs[2]: "ECAPA-TDNN-512 + OCSoftmax bf16 train").
//
// ecapa_tdnn.py:39 (Bottle2neck.conv1), :55 (conv3), :118 (layer4), :140/:143 (attention) are
// plain GEMMs per utterance and hold 97 % of the model's FLOPs:
//     forward   Y_b (Cout x T) = W (Cout x Cin)   . X_b  (Cin x T)
//     dgrad     dX_b (Cin x T) = W^T (Cin x Cout) . dY_b (Cout x T)
//     wgrad     dW (Cout x Cin) = sum_b dY_b (Cout x T) . X_b^T (T x Cin)
// Tensors stay fp32 in HBM, (B, C, T) with T contiguous and a batch stride (channel-slice
// views are addressed in place).  Arithmetic: both operands are rounded to bf16 (round to
// nearest even, v_cvt_pk_bf16_f32) while they are staged into LDS, products are exact,
// accumulation is fp32 inside v_mfma_f32_32x32x16_bf16 - what torch autocast(bf16) computes
// for these layers.  oracle/ecapa.py restates exactly this (operands .bfloat16().float(), fp32
// conv), so parity differs by summation order only.
//
// Tiling (both kernels): 256 threads = 4 waves own a 128 x 128 output tile, each wave a
// 64 x 64 quadrant = 2 x 2 MFMA tiles (64 accumulator registers); K advances 32 per stage
// through a double-buffered LDS pair.  An MFMA lane needs 8 CONSECUTIVE k of one row/column:
//   * forward/dgrad: k = input channel, but memory is contiguous along t.  The staging thread
//     reads 8 channel rows at ITS t (each load 256 B coalesced per wave), packs the 8 values and
//     writes one 16-byte [t][k0..k0+7] LDS row segment: the transpose happens in registers.
//   * wgrad: k = t, contiguous in memory for both operands: 16-byte loads, pack, 8-byte LDS writes.
// LDS rows are padded to 80 bytes: the 16 lanes ds_read_b128 serves per cycle hit 64 distinct banks.
// Workgroups are numbered so the tiles sharing an X_b time tile (all Cout tiles) run
// back-to-back on ONE XCD and re-read it from that XCD's L2.
#include "air_common.h"
#include "air_prof.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 8;  // bf16 elements per LDS row: 80 bytes
constexpr int NXCD = 8;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, r);
}

// blockIdx -> work item: consecutive work items live on one XCD (hardware deals workgroups to
// XCDs round-robin), so neighbours in the work order share that XCD's L2.
__device__ __forceinline__ int xcd_chunked(int bid, int per_xcd) { return (bid % NXCD) * per_xcd + bid / NXCD; }

// fp32 weights -> bf16 A operand [M][K] (transpose = 1: A[m][k] = w[k][m], the dgrad operand)
__global__ __launch_bounds__(256) void c1b_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ a,
                                                       int M, int K, int transpose) {
  const size_t n = (size_t)M * K / 2;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const size_t m = (2 * e) / K, k = (2 * e) % K;
    float v0, v1;
    if (transpose) {
      v0 = w[k * M + m];
      v1 = w[(k + 1) * M + m];
    } else {
      v0 = w[m * K + k];
      v1 = w[m * K + k + 1];
    }
    reinterpret_cast<unsigned*>(a)[e] = pack2(v0, v1);
  }
}

struct C1bFwd {
  const float* x;
  const unsigned short* a;
  float* y;
  const float* bias;
  const float* bias_bc;
  const float* acc;
  size_t x_bs, y_bs;
  int B, M, K, T, relu, tiles_m, tiles_t, total, per_xcd;
};

// One K = 32 stage of a wave's (32 TM) x (32 TN) quadrant from the staged LDS tiles.
template <int TM, int TN>
__device__ __forceinline__ void mma_stage(const unsigned short* __restrict__ sa, const unsigned short* __restrict__ sb,
                                          int wm, int wn, int lane, f32x16 (&acc)[TM][TN]) {
  const int r = lane & 31, kg = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk) {
    bf16x8 a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
      a[i] = *reinterpret_cast<const bf16x8*>(sa + (wm * 32 * TM + i * 32 + r) * LDK + kk * 16 + kg * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      b[j] = *reinterpret_cast<const bf16x8*>(sb + (wn * 32 * TN + j * 32 + r) * LDK + kk * 16 + kg * 8);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// Y_b[m][t] = sum_k A[m][k] X_b[k][t]  (+ bias[m] + bias_bc[b][m] + acc_b[m][t], optional ReLU)
// Workgroup tile (64 TM) x 128: TM = 4 for the wide layers (every X tile is then re-read by half
// as many workgroups), TM = 2 otherwise.
template <int TM>
__global__ __launch_bounds__(256, 2) void c1b_fwd_kernel(const C1bFwd p) {
  constexpr int TBM = 64 * TM;
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][TBM * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][BN * LDK];
  const int work = xcd_chunked(blockIdx.x, p.per_xcd);
  if (work >= p.total) return;
  const int mt = work % p.tiles_m;
  const int rest = work / p.tiles_m;
  const int tt = rest % p.tiles_t, b = rest / p.tiles_t;
  const int m0 = mt * TBM, t0 = tt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // staging roles.  A: TBM rows x 4 16-byte chunks (TM slots of 64 rows); B: 128 t x 4 groups of 8 channels.
  const int a_row = tid >> 2, a_ch = tid & 3;
  const int b_t = tid & 127, b_kg = tid >> 7;  // + 2 groups for the second slot
  const bool t_ok = t0 + b_t < p.T;
  const unsigned short* __restrict__ ga = p.a + (size_t)(m0 + a_row) * p.K + a_ch * 8;
  const float* __restrict__ gx = p.x + (size_t)b * p.x_bs + (size_t)(b_kg * 8) * p.T + t0 + b_t;

  uint4 ra[TM];
  float rb[2][8];
#define C1B_FETCH(k0)                                                                                   \
  do {                                                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < TM; ++s_)                                                   \
        ra[s_] = *reinterpret_cast<const uint4*>(ga + (size_t)s_ * 64 * p.K + (k0));                    \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                                \
            rb[s_][j_] = t_ok ? gx[(size_t)((k0) + s_ * 16 + j_) * p.T] : 0.0f;                        \
  } while (0)
#define C1B_STASH(buf)                                                                                  \
  do {                                                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < TM; ++s_)                                                   \
        *reinterpret_cast<uint4*>(&sA[buf][(a_row + s_ * 64) * LDK + a_ch * 8]) = ra[s_];               \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                  \
      uint4 v_;                                                                                         \
      v_.x = pack2(rb[s_][0], rb[s_][1]);                                                               \
      v_.y = pack2(rb[s_][2], rb[s_][3]);                                                               \
      v_.z = pack2(rb[s_][4], rb[s_][5]);                                                               \
      v_.w = pack2(rb[s_][6], rb[s_][7]);                                                               \
      *reinterpret_cast<uint4*>(&sB[buf][b_t * LDK + (b_kg + s_ * 2) * 8]) = v_;                        \
    }                                                                                                   \
  } while (0)

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = p.K / BK;
  C1B_FETCH(0);
  C1B_STASH(0);
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int cur = s & 1;
    if (s + 1 < nk) C1B_FETCH((s + 1) * BK);
    mma_stage<TM, 2>(sA[cur], sB[cur], wm, wn, lane, acc);
    if (s + 1 < nk) C1B_STASH(cur ^ 1);
    __syncthreads();
  }
#undef C1B_FETCH
#undef C1B_STASH

  const int col = lane & 31, half = lane >> 5;
  float* __restrict__ yb = p.y + (size_t)b * p.y_bs;
  const float* __restrict__ ab = p.acc ? p.acc + (size_t)b * p.y_bs : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float add = 0.0f;
      if (p.bias) add += p.bias[m];
      if (p.bias_bc) add += p.bias_bc[(size_t)b * p.M + m];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = t0 + wn * 64 + j * 32 + col;
        if (t < p.T) {
          float v = acc[i][j][r] + add;
          if (ab) v += ab[(size_t)m * p.T + t];
          if (p.relu) v = fmaxf(v, 0.0f);
          yb[(size_t)m * p.T + t] = v;
        }
      }
    }
  }
}

struct C1bWgrad {
  const float* dy;  // (B, M, T)
  const float* x;   // (B, N, T)
  float* out;       // nsplit > 1: partial[split][M][N]; else dW[M][N]
  size_t dy_bs, x_bs;
  int B, M, N, T, tiles_m, tiles_n, nsplit, b_per_split, total, per_xcd, vec_ok;
};

// 4 consecutive t of one row, zero beyond T.  vec: the row start is 8-byte aligned.
__device__ __forceinline__ void load4(const float* __restrict__ row, int t, int T, bool vec, float (&v)[4]) {
  if (vec && t + 3 < T) {
    const f32x4a8 q = *reinterpret_cast<const f32x4a8*>(row + t);
    v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t + e < T ? row[t + e] : 0.0f;
  }
}

// dW[m][n] = sum_{b in split} sum_t dY_b[m][t] X_b[n][t].  Workgroup tile (64 TM) x (64 TN):
// 256 x 256 where the layer allows it - every staged fp32 byte then feeds twice the MFMA work of a
// 128 x 128 tile, which is what bounds this kernel (L2 -> LDS traffic, not HBM and not the MFMA).
template <int TM, int TN>
__global__ __launch_bounds__(256) void c1b_wgrad_kernel(const C1bWgrad p) {
  constexpr int TBM = 64 * TM, TBN = 64 * TN;
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][TBM * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][TBN * LDK];
  const int work = xcd_chunked(blockIdx.x, p.per_xcd);
  if (work >= p.total) return;
  const int tiles = p.tiles_m * p.tiles_n;
  const int tile = work % tiles, split = work / tiles;
  const int mt = tile % p.tiles_m, nt = tile / p.tiles_m;
  const int m0 = mt * TBM, n0 = nt * TBN;
  const int b_lo = split * p.b_per_split;
  const int b_hi = min(p.B, b_lo + p.b_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const bool vec = p.vec_ok != 0;

  // staging: rows x 8 float4 chunks per operand, 32 rows per slot
  const int s_row = tid >> 3, s_c4 = tid & 7;
  float ra[2 * TM][4], rb[2 * TN][4];
  auto fetch = [&](int b, int t0) {
    const float* __restrict__ gy = p.dy + (size_t)b * p.dy_bs + (size_t)(m0 + s_row) * p.T;
    const float* __restrict__ gx = p.x + (size_t)b * p.x_bs + (size_t)(n0 + s_row) * p.T;
    const int t = t0 + s_c4 * 4;
#pragma unroll
    for (int s = 0; s < 2 * TM; ++s) load4(gy + (size_t)s * 32 * p.T, t, p.T, vec, ra[s]);
#pragma unroll
    for (int s = 0; s < 2 * TN; ++s) load4(gx + (size_t)s * 32 * p.T, t, p.T, vec, rb[s]);
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2 * TM; ++s) {
      uint2 v;
      v.x = pack2(ra[s][0], ra[s][1]);
      v.y = pack2(ra[s][2], ra[s][3]);
      *reinterpret_cast<uint2*>(&sA[buf][(s_row + s * 32) * LDK + s_c4 * 4]) = v;
    }
#pragma unroll
    for (int s = 0; s < 2 * TN; ++s) {
      uint2 v;
      v.x = pack2(rb[s][0], rb[s][1]);
      v.y = pack2(rb[s][2], rb[s][3]);
      *reinterpret_cast<uint2*>(&sB[buf][(s_row + s * 32) * LDK + s_c4 * 4]) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int steps_per_b = (p.T + BK - 1) / BK;
  const int nsteps = (b_hi - b_lo) * steps_per_b;
  int nb = b_lo, nt0 = 0;  // coordinates of the NEXT stage to fetch
  auto advance = [&]() {
    nt0 += BK;
    if (nt0 >= p.T) { nt0 = 0; ++nb; }
  };
  if (nsteps > 0) {
    fetch(nb, nt0);
    advance();
    stash(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) { fetch(nb, nt0); advance(); }
    mma_stage<TM, TN>(sA[cur], sB[cur], wm, wn, lane, acc);
    if (s + 1 < nsteps) stash(cur ^ 1);
    __syncthreads();
  }

  const int col = lane & 31, half = lane >> 5;
  float* __restrict__ out = p.out + (p.nsplit > 1 ? (size_t)split * p.M * p.N : 0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
      for (int j = 0; j < TN; ++j) out[(size_t)m * p.N + n0 + wn * 32 * TN + j * 32 + col] = acc[i][j][r];
    }
}

// dw[e] = sum_s partial[s][e], fixed order (deterministic)
__global__ __launch_bounds__(256) void c1b_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                         size_t n, int nsplit) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n / 4; e += (size_t)gridDim.x * 256) {
    float4 s = reinterpret_cast<const float4*>(partial)[e];
    for (int k = 1; k < nsplit; ++k) {
      const float4 v = reinterpret_cast<const float4*>(partial + (size_t)k * n)[e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(dw)[e] = s;
  }
}

bool shape_ok(const AirConv1d* p) {
  return p && p->B > 0 && p->T > 0 && p->Cin > 0 && p->Cout > 0 && p->K == 1 && p->pad == 0;
}
size_t xbs(const AirConv1d* p) { return p->x_bstride ? p->x_bstride : (size_t)p->Cin * p->T; }
size_t ybs(const AirConv1d* p) { return p->y_bstride ? p->y_bstride : (size_t)p->Cout * p->T; }

// weight-gradient tile: 256 where the channel count divides, else 128
int wg_tm(const AirConv1d* p) { return p->Cout % 256 == 0 ? 4 : 2; }
int wg_tn(const AirConv1d* p) { return p->Cin % 256 == 0 ? 4 : 2; }

int wgrad_nsplit(const AirConv1d* p, int* b_per_split) {
  const int tiles = (p->Cout / (64 * wg_tm(p))) * (p->Cin / (64 * wg_tn(p)));
  int want = 512 / tiles;
  if (want < 1) want = 1;
  if (want > p->B) want = p->B;
  const int per = (p->B + want - 1) / want;
  *b_per_split = per;
  return (p->B + per - 1) / per;
}

int run_fwd(const float* x, size_t x_bs, const float* w, int transpose, float* y, size_t y_bs, const float* bias,
            const float* bias_bc, const float* acc, int relu, int B, int M, int K, int T, void* ws, double flops,
            hipStream_t st) {
  unsigned short* a = reinterpret_cast<unsigned short*>(ws);
  const size_t n2 = (size_t)M * K / 2;
  hipLaunchKernelGGL(c1b_pack_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, w, a, M, K, transpose);
  AIR_CHECK_LAUNCH();
  C1bFwd p;
  p.x = x; p.a = a; p.y = y; p.bias = bias; p.bias_bc = bias_bc; p.acc = acc;
  p.x_bs = x_bs; p.y_bs = y_bs;
  p.B = B; p.M = M; p.K = K; p.T = T; p.relu = relu;
  // 256-row tiles once the layer is wide enough to still fill the chip with them
  const int tm = (M % 256 == 0 && M >= 1024) ? 4 : 2;
  p.tiles_m = M / (64 * tm);
  p.tiles_t = (T + BN - 1) / BN;
  p.total = B * p.tiles_t * p.tiles_m;
  p.per_xcd = (p.total + NXCD - 1) / NXCD;
  AirProfScope prof(AIR_K_C1B_FWD, flops, st);
  if (tm == 4)
    hipLaunchKernelGGL(c1b_fwd_kernel<4>, dim3(p.per_xcd * NXCD), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(c1b_fwd_kernel<2>, dim3(p.per_xcd * NXCD), dim3(256), 0, st, p);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

}  // namespace

extern "C" {

int air_conv1d_bf16_supported(const AirConv1d* p, int pass) {
  if (!shape_ok(p)) return 0;
  switch (pass) {
    case 0: return p->Cout % BM == 0 && p->Cin % BK == 0;  // forward: M = Cout, K = Cin
    case 1: return p->Cin % BM == 0 && p->Cout % BK == 0;  // dgrad:   M = Cin,  K = Cout
    case 2: return p->Cout % BM == 0 && p->Cin % BN == 0;  // wgrad:   M = Cout, N = Cin
    default: return 0;
  }
}

size_t air_conv1d_bf16_ws_bytes(const AirConv1d* p) {
  if (!shape_ok(p)) return 0;
  size_t n = (size_t)p->Cout * p->Cin * sizeof(unsigned short);  // packed bf16 weights
  if (air_conv1d_bf16_supported(p, 2)) {
    int per;
    const size_t part = (size_t)wgrad_nsplit(p, &per) * p->Cout * p->Cin * sizeof(float);
    if (part > n) n = part;
  }
  return n + 256;
}

int air_conv1d_fwd_bf16(const AirConv1d* p, const float* x, const float* w, const float* bias, const float* bias_bc,
                        int relu, float* y, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!shape_ok(p) || !x || !w || !y) return AIR_EINVAL;
  if (!air_conv1d_bf16_supported(p, 0)) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_bf16_ws_bytes(p)) return AIR_EWORKSPACE;
  return run_fwd(x, xbs(p), w, 0, y, ybs(p), bias, bias_bc, nullptr, relu, p->B, p->Cout, p->Cin, p->T, ws,
                 2.0 * p->B * p->T * (double)p->Cout * p->Cin, air_stream(stream));
}

int air_conv1d_dgrad_bf16(const AirConv1d* p, const float* dy, const float* w, float* dx, const float* accumulate,
                          void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!shape_ok(p) || !dy || !w || !dx) return AIR_EINVAL;
  if (!air_conv1d_bf16_supported(p, 1)) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_bf16_ws_bytes(p)) return AIR_EWORKSPACE;
  // A = W^T: w is (Cout, Cin) = [k][m]
  return run_fwd(dy, ybs(p), w, 1, dx, xbs(p), nullptr, nullptr, accumulate, 0, p->B, p->Cin, p->Cout, p->T, ws,
                 2.0 * p->B * p->T * (double)p->Cout * p->Cin, air_stream(stream));
}

int air_conv1d_wgrad_bf16(const AirConv1d* p, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                          air_stream_t stream) {
  if (!shape_ok(p) || !x || !dy || !dw) return AIR_EINVAL;
  if (!air_conv1d_bf16_supported(p, 2)) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_bf16_ws_bytes(p)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  C1bWgrad a;
  a.dy = dy; a.x = x;
  a.dy_bs = ybs(p); a.x_bs = xbs(p);
  a.B = p->B; a.M = p->Cout; a.N = p->Cin; a.T = p->T;
  const int tm = wg_tm(p), tn = wg_tn(p);
  a.tiles_m = a.M / (64 * tm); a.tiles_n = a.N / (64 * tn);
  a.nsplit = wgrad_nsplit(p, &a.b_per_split);
  a.out = a.nsplit > 1 ? reinterpret_cast<float*>(ws) : dw;
  a.total = a.tiles_m * a.tiles_n * a.nsplit;
  a.per_xcd = (a.total + NXCD - 1) / NXCD;
  // 16-byte loads need every row start 8-byte aligned
  a.vec_ok = p->T % 2 == 0 && a.dy_bs % 2 == 0 && a.x_bs % 2 == 0 && (reinterpret_cast<size_t>(x) & 7) == 0 &&
             (reinterpret_cast<size_t>(dy) & 7) == 0;
  {
    AirProfScope prof(AIR_K_C1B_WGRAD, 2.0 * p->B * p->T * (double)p->Cout * p->Cin, st);
    const dim3 grid(a.per_xcd * NXCD), blk(256);
    if (tm == 4 && tn == 4) hipLaunchKernelGGL((c1b_wgrad_kernel<4, 4>), grid, blk, 0, st, a);
    else if (tm == 4) hipLaunchKernelGGL((c1b_wgrad_kernel<4, 2>), grid, blk, 0, st, a);
    else if (tn == 4) hipLaunchKernelGGL((c1b_wgrad_kernel<2, 4>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((c1b_wgrad_kernel<2, 2>), grid, blk, 0, st, a);
    AIR_CHECK_LAUNCH();
  }
  if (a.nsplit > 1) {
    const size_t n = (size_t)a.M * a.N;
    hipLaunchKernelGGL(c1b_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), dw, n, a.nsplit);
    AIR_CHECK_LAUNCH();
  }
  return AIR_OK;
}

}  // extern "C"
