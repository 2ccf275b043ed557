// 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2,3x3) on the f32 MFMA of gfx950.
//
// Replaces the 3x3 nn.Conv2d of PreActBlock (resnet.py:56-61) in forward and data-gradient
// direction.  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A: per 2x2 output tile 16 multiplies
// instead of 36, i.e. 16 independent GEMMs  M_xn[co][tile] = sum_ci U_xn[co][ci] V_xn[ci][tile]
// that run as 16 accumulators of v_mfma_f32_32x32x2_f32 in one wave:
//   * a wave owns 32 output channels x 32 tiles (128 output pixels): 16 accumulators x 16
//     registers = 256 AGPRs, one wave per SIMD; a workgroup = 2 channel halves x 2 tile groups.
//   * per k-step (2 input channels) a lane reads ITS 4x4 input patch from the LDS-staged image
//     rows (8 x ds_read_b64), transforms it in registers (32 adds) and feeds 16 MFMAs; the
//     transformed weights come pre-packed from a small transform kernel (4 x ds_read_b128).
//     Nothing transformed ever touches HBM, and the output transform is per-lane on the
//     accumulators.
//   * K chunks of 4 channels are double buffered through LDS with hand-issued LDS-DMA; the
//     DMA issue, the LDS reads and the transform of step n+1 are placed in the shadow of the
//     MFMAs of step n (one wave per SIMD: nothing else hides them).
// Tile groups are 1 x 32 or 2 x 16 tiles, whichever wastes less of the (H, W) at hand.
#include <stdlib.h>

#include "air_common.h"
#include "air_lds_dma.h"
#include "air_prof.h"
#include "conv_wino.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WCK = 4;                  // input channels per K chunk (= 2 k-steps)
constexpr int WBM = 64;                 // output channels per workgroup
constexpr int USLAB = WCK * 16 * WBM;   // floats of transformed weights per chunk

// packed transformed weights: Up[cot][chunk][cil][q][col 64][j],  U[4q + j] of (co, ci)
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ up, int M,
                                    int Kc, int dgrad) {
  const int Mpad = (M + WBM - 1) / WBM * WBM;
  const int nchunk = Kc / WCK;
  const size_t total = (size_t)Mpad * Kc;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % WBM);
    size_t r = e / WBM;
    const int k = (int)(r % Kc);
    const int cot = (int)(r / Kc);
    const int m = cot * WBM + col;
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = 0.0f;
      if (m < M) v = dgrad ? w[((size_t)k * M + m) * 9 + (8 - t)] : w[((size_t)m * Kc + k) * 9 + t];
      g[t] = v;
    }
    float tmp[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      tmp[0][c] = g[c];
      tmp[1][c] = 0.5f * (g[c] + g[3 + c] + g[6 + c]);
      tmp[2][c] = 0.5f * (g[c] - g[3 + c] + g[6 + c]);
      tmp[3][c] = g[6 + c];
    }
    const int chunk = k / WCK, cil = k % WCK;
    float* o = up + (((size_t)cot * nchunk + chunk) * WCK + cil) * (16 * WBM) + col * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 u;
      u[0] = tmp[i][0];
      u[1] = 0.5f * (tmp[i][0] + tmp[i][1] + tmp[i][2]);
      u[2] = 0.5f * (tmp[i][0] - tmp[i][1] + tmp[i][2]);
      u[3] = tmp[i][2];
      *reinterpret_cast<f32x4*>(o + i * (WBM * 4)) = u;
    }
  }
}

struct WinoArgs {
  const float* x;         // (B, Cin, H, W)
  const float* up;        // packed transformed weights
  float* y;               // (B, Cout, H, W)
  const float* residual;  // same shape as y (may be null)
  int B, Cin, H, W, Cout;
  int TH, TW;             // 2x2 output tiles per image
  int THG, TWG;           // tile groups per image
  int ngroups;            // B * THG * TWG
  int ncot;               // Cout / 64 rounded up
  int dbg;
};

template <int TR>
struct WinoCfg {
  static constexpr int TC = 32 / TR;            // tile columns of a group
  static constexpr int PR = 2 * TR + 2;         // patch rows per channel
  static constexpr int PC = 2 * TC + 2;         // patch columns (even: ds_read_b64 aligned)
  static constexpr int CHS = PR * PC;           // channel pitch
  static constexpr int NE = 2 * WCK * CHS;      // patch elements per workgroup (2 groups)
  static constexpr int NI = (NE + 255) / 256;   // 4-byte DMAs per thread
  static constexpr int PATCH = NI * 256;
  static constexpr int BUF = USLAB + PATCH;     // floats per LDS buffer
  static constexpr int ND = NI + 4;             // DMA instructions per thread and chunk
};

// NBUF LDS buffers form a ring: chunk c lives in buffer c % NBUF and is staged NBUF-1 chunks
// (about (NBUF-1) * 2048 cycles) before it is read, which covers the L2/HBM latency that one
// resident wave per SIMD cannot hide any other way.
constexpr int NBUF = 4;
template <int N>
__device__ __forceinline__ void dma_wait_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TR>
__global__ __launch_bounds__(256) void wino_conv_kernel(WinoArgs a) {
  using C = WinoCfg<TR>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert((NBUF - 2) * C::ND <= 63, "vmcnt is a 6-bit counter");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int cw = wave & 1;   // channel half of the workgroup's 64
  const int gw = wave >> 1;  // tile group of the workgroup's 2

  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lb % a.ncot;
  const int pg = lb / a.ncot;
  const int HWi = a.H * a.W;
  const int nchunk = a.Cin / WCK;

  // origins of both tile groups of this workgroup (every thread stages for both)
  int gb[2], gh0[2], gw0[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int g = min(pg * 2 + k, a.ngroups - 1);
    const int twg = g % a.TWG;
    const int r = g / a.TWG;
    gb[k] = r / a.THG;
    gh0[k] = 2 * TR * (r % a.THG) - 1;
    gw0[k] = 2 * C::TC * twg - 1;
  }
  // chunk-invariant source offsets of this thread's patch elements (clamped into the image;
  // zero padding is applied when the operand is read)
  int goff[C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = min(i * 256 + tid, C::NE - 1);
    const int grp = e / (WCK * C::CHS);
    const int rem = e - grp * (WCK * C::CHS);
    const int cil = rem / C::CHS;
    const int rem2 = rem - cil * C::CHS;
    const int r = rem2 / C::PC;
    const int c = rem2 - r * C::PC;
    const int hi = min(max((grp ? gh0[1] : gh0[0]) + r, 0), a.H - 1);
    const int wi = min(max((grp ? gw0[1] : gw0[0]) + c, 0), a.W - 1);
    goff[i] = ((grp ? gb[1] : gb[0]) * a.Cin + cil) * HWi + hi * a.W + wi;
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  const float* __restrict__ uslab0 = a.up + (size_t)cot * nchunk * USLAB;
  // DMA instruction k (0 .. ND-1) of chunk -> buffer
  auto dma_one = [&](int chunk, int buf, int k) {
    const unsigned base = lds0 + 4u * (buf * C::BUF);
    if (k < C::NI) {
      const float* __restrict__ xc = a.x + (size_t)chunk * WCK * HWi;
      dma4(xc + goff[k < C::NI ? k : 0], base + 4u * (USLAB + k * 256 + wave * 64));
    } else {
      const int i = k - C::NI;
      const float* __restrict__ us = uslab0 + (size_t)chunk * USLAB;
      dma16(us + 4 * (i * 256 + tid), base + 16u * (i * 256 + wave * 64));
    }
  };

  // this lane's tile
  const int g = pg * 2 + gw;
  const bool g_ok = g < a.ngroups;
  const int tr = l31 / C::TC, tc = l31 % C::TC;
  const int h0 = gh0[gw] + 2 * tr, w0 = gw0[gw] + 2 * tc;  // top-left of the 4x4 input patch
  bool okm[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      okm[r][c] = h0 + r >= 0 && h0 + r < a.H && w0 + c >= 0 && w0 + c < a.W;

  // per-lane LDS offsets (floats, within a buffer)
  const int pb_lane = USLAB + (gw * WCK + half) * C::CHS + (2 * tr) * C::PC + 2 * tc;
  const int ub_lane = half * (16 * WBM) + (cw * 32 + l31) * 4;

  f32x16 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = (f32x16){0};

  f32x4 U0[4], U1[4];   // transformed weights of step 0 / 1
  float v0[16], v1[16];  // transformed input of step 0 / 1
  f32x2 d[8];            // raw patch rows: d[2r], d[2r+1] = columns 0-1, 2-3 of row r
  float t[4][4];

  auto ldU = [&](const float* __restrict__ bufp, int s, f32x4(&u)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      u[q] = *reinterpret_cast<const f32x4*>(bufp + ub_lane + s * (2 * 16 * WBM) + q * (WBM * 4));
  };
  auto ldD = [&](const float* __restrict__ bufp, int s) {
    const float* __restrict__ p = bufp + pb_lane + s * (2 * C::CHS);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[2 * r] = *reinterpret_cast<const f32x2*>(p + r * C::PC);
      d[2 * r + 1] = *reinterpret_cast<const f32x2*>(p + r * C::PC + 2);
    }
  };
  // B^T d B in two passes of 4 pieces each: rows (with the zero-padding mask), then columns
  auto row_piece = [&](int c) {
    float x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float raw = (c & 1) ? d[2 * r + (c >> 1)][1] : d[2 * r + (c >> 1)][0];
      x[r] = okm[r][c] ? raw : 0.0f;
    }
    t[0][c] = x[0] - x[2];
    t[1][c] = x[1] + x[2];
    t[2][c] = x[2] - x[1];
    t[3][c] = x[1] - x[3];
    // pin the results to this MFMA slot (the optimiser otherwise sinks them to their use)
    asm volatile("" : "+v"(t[0][c]), "+v"(t[1][c]), "+v"(t[2][c]), "+v"(t[3][c]));
  };
  auto col_piece = [&](int i, float(&v)[16]) {
    v[4 * i + 0] = t[i][0] - t[i][2];
    v[4 * i + 1] = t[i][1] + t[i][2];
    v[4 * i + 2] = t[i][2] - t[i][1];
    v[4 * i + 3] = t[i][1] - t[i][3];
    asm volatile("" : "+v"(v[4 * i]), "+v"(v[4 * i + 1]), "+v"(v[4 * i + 2]), "+v"(v[4 * i + 3]));
  };

  // prologue: chunk 0 staged and transformed (exposed once), chunk 1 in flight
#pragma unroll
  for (int cb = 0; cb < NBUF - 1; ++cb) {
    if (cb < nchunk) {
#pragma unroll
      for (int k = 0; k < C::ND; ++k) dma_one(cb, cb, k);
    }
  }
  if (nchunk >= NBUF - 1)
    dma_wait_n<(NBUF - 2) * C::ND>();  // chunk 0 has landed; the others may still fly
  else
    dma_wait();
  __syncthreads();
  ldU(lds, 0, U0);
  ldD(lds, 0);
#pragma unroll
  for (int c = 0; c < 4; ++c) row_piece(c);
#pragma unroll
  for (int i = 0; i < 4; ++i) col_piece(i, v0);

  constexpr int DPS = (C::ND + 3) / 4;  // DMA instructions per MFMA slot (slots 0-3)
  int cur = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk, cur = (cur + 1 == NBUF ? 0 : cur + 1)) {
    const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
    const int prv = cur == 0 ? NBUF - 1 : cur - 1;  // buffer of chunk - 1 = of chunk + NBUF - 1
    const float* __restrict__ bcur = lds + cur * C::BUF;
    const float* __restrict__ bnxt = lds + nxt * C::BUF;
    // ---- step 0 of this chunk; fetch + transform step 1 in the MFMA shadow
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j == 0) ldU(bcur, 1, U1);
      if (j == 1) ldD(bcur, 1);
      if (j >= 4 && j < 8) row_piece(j - 4);
      if (j >= 8 && j < 12) col_piece(j - 8, v1);
      __builtin_amdgcn_sched_barrier(0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(U0[j >> 2][j & 3], v0[j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- step 1; the next chunk has landed: swap, restage the freed buffer, fetch step 0
    // chunk + 1 must have landed (DMAs complete in order; chunks up to + NBUF - 2 are in flight)
    if (!(a.dbg & 1)) {
    if (chunk + NBUF - 2 < nchunk)
      dma_wait_n<(NBUF - 3) * C::ND>();
    else
      dma_wait();
    __syncthreads();
    }
    const bool more2 = chunk + NBUF - 1 < nchunk && !(a.dbg & 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < 4 && more2) {
#pragma unroll
        for (int k = j * DPS; k < (j + 1) * DPS && k < C::ND; ++k) dma_one(chunk + NBUF - 1, prv, k);
      }
      // (after the last chunk these read stale LDS: unused, but branch-free)
      if (j == 4) ldU(bnxt, 0, U0);
      if (j == 5) ldD(bnxt, 0);
      if (j >= 8 && j < 12) row_piece(j - 8);
      if (j >= 12) col_piece(j - 12, v0);
      __builtin_amdgcn_sched_barrier(0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(U1[j >> 2][j & 3], v1[j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // epilogue: Y = A^T M A per lane; D row i = (r&3) + 8*(r>>2) + 4*half -> channel
  if (!g_ok || (a.dbg & 2)) return;
  const int th = (gh0[gw] + 1) / 2 + tr, tw = (gw0[gw] + 1) / 2 + tc;
  if (th >= a.TH || tw >= a.TW) return;
  const int ho = 2 * th, wo = 2 * tw;
  const bool w1 = wo + 1 < a.W, h1 = ho + 1 < a.H;
  const size_t obase = ((size_t)gb[gw] * a.Cout + cot * WBM + cw * 32) * HWi + (size_t)ho * a.W + wo;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
    const int co = cot * WBM + cw * 32 + i;
    if (co >= a.Cout) continue;
    float s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s0[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
      s1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
    }
    float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
    float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
    const size_t o = obase + (size_t)i * HWi;
    if (a.residual != nullptr) {
      const float r00 = a.residual[o];
      const float r01 = w1 ? a.residual[o + 1] : 0.0f;
      const float r10 = h1 ? a.residual[o + a.W] : 0.0f;
      const float r11 = (w1 && h1) ? a.residual[o + a.W + 1] : 0.0f;
      y00 += r00; y01 += r01; y10 += r10; y11 += r11;
    }
    a.y[o] = y00;
    if (w1) a.y[o + 1] = y01;
    if (h1) a.y[o + a.W] = y10;
    if (w1 && h1) a.y[o + a.W + 1] = y11;
  }
}

int grid_for(size_t n) {
  size_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

bool air_wino_ok(int B, int Kc, int H, int W, int M) {
  static const int off = getenv("AIR_NO_WINOGRAD") ? atoi(getenv("AIR_NO_WINOGRAD")) : 0;
  if (off) return false;
  if (M < 32 || Kc < WCK || Kc % WCK != 0) return false;
  const double ein = (double)B * Kc * H * W, eout = (double)B * M * H * W;
  return ein < 2147483647.0 && eout < 1e18 && H >= 1 && W >= 2;
}

size_t air_wino_packed_elems(int M, int Kc) {
  return (size_t)((M + WBM - 1) / WBM * WBM) * Kc * 16;
}

int air_wino_conv(const float* x, const float* w, float* y, const float* residual, int B, int Kc,
                  int H, int W, int M, int dgrad, float* up, double flops, hipStream_t st) {
  hipLaunchKernelGGL(wino_weights_kernel, dim3(grid_for((size_t)((M + WBM - 1) / WBM * WBM) * Kc)),
                     dim3(256), 0, st, w, up, M, Kc, dgrad);
  AIR_CHECK_LAUNCH();
  WinoArgs a;
  a.x = x; a.up = up; a.y = y; a.residual = residual;
  a.B = B; a.Cin = Kc; a.H = H; a.W = W; a.Cout = M;
  a.TH = (H + 1) / 2; a.TW = (W + 1) / 2;
  // 1 x 32 or 2 x 16 tiles per group: fewer groups = less padding waste
  const long g1 = (long)a.TH * ((a.TW + 31) / 32), g2 = (long)((a.TH + 1) / 2) * ((a.TW + 15) / 16);
  const int trows = g2 < g1 ? 2 : 1;
  a.THG = (a.TH + trows - 1) / trows;
  a.TWG = (a.TW + 32 / trows - 1) / (32 / trows);
  a.ngroups = B * a.THG * a.TWG;
  a.ncot = (M + WBM - 1) / WBM;
  a.dbg = getenv("AIR_WINO_DBG") ? atoi(getenv("AIR_WINO_DBG")) : 0;
  const int nblk = (a.ngroups + 1) / 2 * a.ncot;
  AirProfScope ps(AIR_K_CONV_WINO, flops, st);
  static const bool attr_ok = [] {  // > 64 KB of dynamic LDS needs the opt-in, once per kernel
    return hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               NBUF * WinoCfg<1>::BUF * sizeof(float)) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel<2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               NBUF * WinoCfg<2>::BUF * sizeof(float)) == hipSuccess;
  }();
  if (!attr_ok) return AIR_ELAUNCH;
  if (trows == 2)
    hipLaunchKernelGGL(wino_conv_kernel<2>, dim3(nblk), dim3(256),
                       NBUF * WinoCfg<2>::BUF * sizeof(float), st, a);
  else
    hipLaunchKernelGGL(wino_conv_kernel<1>, dim3(nblk), dim3(256),
                       NBUF * WinoCfg<1>::BUF * sizeof(float), st, a);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}
