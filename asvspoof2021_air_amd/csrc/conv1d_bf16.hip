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
// Two paths (chosen per layer, see `wide()`):
//  1. FUSED forward/dgrad for the 512-channel layers (HBM-bound on the fp32 tensors, so one pass
//     that converts in flight wins): 256 threads = 4 waves own a 128 x 128 output tile, each wave a
//     64 x 64 quadrant = 2 x 2 MFMA tiles (64 accumulator registers); K advances 32 per stage
//     through a double-buffered LDS pair.  An MFMA lane needs 8 CONSECUTIVE k (= input channels) of
//     one column, but memory is contiguous along t: the staging thread reads 8 channel rows at ITS t
//     (each load 256 B coalesced per wave), packs the 8 values and writes one 16-byte
//     [t][k0..k0+7] LDS row segment - the transpose happens in registers.  LDS rows are padded to
//     80 bytes: the 16 lanes ds_read_b128 serves per cycle hit 64 distinct banks.
//  2. bf16-RESIDENT operands + LDS-DMA GEMM for the wide layers and every weight gradient (below).
// Workgroups are numbered so the tiles sharing an X_b time tile (all Cout tiles) run
// back-to-back on ONE XCD and re-read it from that XCD's L2.
#include "air_common.h"
#include "air_options.h"
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

typedef unsigned short u16;
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
  const float* acc;   // accumulate operands: y += acc (batch stride acc_bs) + acc2 (batch stride acc2_bs)
  const float* acc2;
  size_t x_bs, y_bs, acc_bs, acc2_bs;
  int B, M, K, T, relu, tiles_m, tiles_t, total, per_xcd;
};

struct Frags {
  bf16x8 a[2], b[2];
};

__device__ __forceinline__ void mma_stage(const unsigned short* __restrict__ sa, const unsigned short* __restrict__ sb,
                                          int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
  const int r = lane & 31, kg = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk) {
    Frags f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      f.a[i] = *reinterpret_cast<const bf16x8*>(sa + (wm * 64 + i * 32 + r) * LDK + kk * 16 + kg * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      f.b[j] = *reinterpret_cast<const bf16x8*>(sb + (wn * 64 + j * 32 + r) * LDK + kk * 16 + kg * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
  }
}

// Y_b[m][t] = sum_k A[m][k] X_b[k][t]  (+ bias[m] + bias_bc[b][m] + acc_b[m][t], optional ReLU)
__global__ __launch_bounds__(256) void c1b_fwd_kernel(const C1bFwd p) {
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][BN * LDK];
  const int work = xcd_chunked(blockIdx.x, p.per_xcd);
  if (work >= p.total) return;
  const int mt = work % p.tiles_m;
  const int rest = work / p.tiles_m;
  const int tt = rest % p.tiles_t, b = rest / p.tiles_t;
  const int m0 = mt * BM, t0 = tt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // staging roles.  A: 128 rows x 4 16-byte chunks; B: 128 t x 4 groups of 8 channels.
  const int a_row = tid >> 2, a_ch = tid & 3;  // + 64 rows for the second slot
  const int b_t = tid & 127, b_kg = tid >> 7;  // + 2 groups for the second slot
  const bool t_ok = t0 + b_t < p.T;
  const unsigned short* __restrict__ ga = p.a + (size_t)(m0 + a_row) * p.K + a_ch * 8;
  const float* __restrict__ gx = p.x + (size_t)b * p.x_bs + (size_t)(b_kg * 8) * p.T + t0 + b_t;

  uint4 ra0, ra1;
  float rb[2][8];
  auto fetch = [&](int k0) {
    ra0 = *reinterpret_cast<const uint4*>(ga + k0);
    ra1 = *reinterpret_cast<const uint4*>(ga + (size_t)64 * p.K + k0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) rb[s][j] = t_ok ? gx[(size_t)(k0 + s * 16 + j) * p.T] : 0.0f;
  };
  auto stash = [&](int buf) {
    *reinterpret_cast<uint4*>(&sA[buf][a_row * LDK + a_ch * 8]) = ra0;
    *reinterpret_cast<uint4*>(&sA[buf][(a_row + 64) * LDK + a_ch * 8]) = ra1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 v;
      v.x = pack2(rb[s][0], rb[s][1]);
      v.y = pack2(rb[s][2], rb[s][3]);
      v.z = pack2(rb[s][4], rb[s][5]);
      v.w = pack2(rb[s][6], rb[s][7]);
      *reinterpret_cast<uint4*>(&sB[buf][b_t * LDK + (b_kg + s * 2) * 8]) = v;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = p.K / BK;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int cur = s & 1;
    if (s + 1 < nk) fetch((s + 1) * BK);
    mma_stage(sA[cur], sB[cur], wm, wn, lane, acc);
    if (s + 1 < nk) stash(cur ^ 1);
    __syncthreads();
  }

  const int col = lane & 31, half = lane >> 5;
  float* __restrict__ yb = p.y + (size_t)b * p.y_bs;
  const float* __restrict__ ab = p.acc ? p.acc + (size_t)b * p.acc_bs : nullptr;
  const float* __restrict__ ab2 = p.acc2 ? p.acc2 + (size_t)b * p.acc2_bs : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float add = 0.0f;
      if (p.bias) add += p.bias[m];
      if (p.bias_bc) add += p.bias_bc[(size_t)b * p.M + m];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = t0 + wn * 64 + j * 32 + col;
        if (t < p.T) {
          float v = acc[i][j][r] + add;
          if (ab) v += ab[(size_t)m * p.T + t];
          if (ab2) v += ab2[(size_t)m * p.T + t];
          if (p.relu) v = fmaxf(v, 0.0f);
          yb[(size_t)m * p.T + t] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same contraction, persistent and with producer waves (the kernel that runs; the one above
// remains for odd T).  Measured on c1b_fwd_kernel (B = 128, T = 750, 512 -> 512: 0.172 ms):
// staging alone 0.076 ms, MFMA + stores alone 0.097 ms, 3072 workgroup launches alone 0.038 ms - the
// phases ADD UP, and rocprofv3 shows why: 17.6 M L2 requests per launch (13.2 M reads: the 64-byte
// A rows use half of every 128-byte line, the X tile is fetched by all four Cout tiles, 512-byte row
// pieces at 8-byte alignment span 5 lines; 4.4 M writes of 44 bytes on average), 565 cycles mean
// L2 latency, the per-CU L1 stalled on pending misses 60 % of the time.  The tile is REQUEST-bound,
// so this kernel is built around the request count:
//   * 256 (Cout) x 128 (frames) tile when Cout % 256 == 0: an X tile is fetched by two workgroups
//     instead of four (8 consumer waves, each 64 x 64 = 2 x 2 MFMA 32x32x16 tiles);
//   * A is packed stage-major, [K/32][M][32 k]: a stage's A tile is ONE contiguous block of full
//     lines, copied lane-linearly (the bank swizzle is baked into the packed layout);
//   * workgroups stay resident (one or two per CU) and walk their XCD's work items in the old order
//     (neighbours in time share X tiles in that XCD's L2);
//   * two extra waves are PRODUCERS: one issues every `buffer_load_dwordx4 ... lds` of the fp32 X
//     stage [32 k][128 t], the other those of the A stage, into a 3-deep LDS ring, two stages
//     ahead, across tile boundaries - and only they count vmcnt for them.  The consumer waves never
//     wait on a DMA: one s_barrier per stage tells them the stage has landed (and tells the
//     producers the slot two behind is free), so their epilogue stores drain underneath the next
//     tile's MFMAs without disturbing a counted wait (loads and stores of ONE wave may retire out of
//     order with respect to each other, so a wave that did both could only wait for vmcnt(0));
//   * the fp32 -> bf16 rounding (v_cvt_pk_bf16_f32, as before) moves to the reader: a lane of the
//     32x32x16 MFMA needs 8 consecutive k at its t = 4 ds_read2st64_b32 down the k rows.
// Frames >= T of a row's last tile read the next row / utterance or, behind the tensor, the
// descriptor's zeros: they only ever reach output columns that are not stored.
constexpr int FD_NS = 3;
constexpr int FD_XB = BK * BN * 4;   // 16 KB: X stage, fp32 [k][t]

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 fd_rsrc(const void* base, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32));  // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void fd_dma16(i32x4 rsrc, unsigned soff, unsigned m0v, unsigned voff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %0, %2 offen lds"
               :: "s"(rsrc), "s"(m0v), "s"(soff), "v"(voff) : "memory", "m0");
}

// fp32 weights -> bf16, stage-major: 16-byte chunk o = (ks * M + m) * 4 + pc holds k = ks * 32 + c * 8 .. + 7 of
// row m with c = pc ^ ((m >> 2) & 3) (the reader's bank swizzle).  transpose = 1: A[m][k] = w[k][m] (dgrad).
__global__ __launch_bounds__(256) void c1b_pack_stage_kernel(const float* __restrict__ w, uint4* __restrict__ a,
                                                             int M, int K, int transpose) {
  const size_t n = (size_t)M * K / 8;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (size_t)gridDim.x * 256) {
    const int pc = (int)(o & 3), m = (int)((o >> 2) % M), ks = (int)((o >> 2) / M);
    const int k0 = ks * BK + ((pc ^ ((m >> 2) & 3)) * 8);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = transpose ? w[(size_t)(k0 + j) * M + m] : w[(size_t)m * K + k0 + j];
    uint4 q;
    q.x = pack2(v[0], v[1]); q.y = pack2(v[2], v[3]); q.z = pack2(v[4], v[5]); q.w = pack2(v[6], v[7]);
    a[o] = q;
  }
}

struct FdTile {
  int b, m0, t0;
};
template <int TBM>
__device__ __forceinline__ FdTile fd_tile(const C1bFwd& p, int work) {
  FdTile t;
  const int mt = work % p.tiles_m, rest = work / p.tiles_m;
  t.m0 = mt * TBM;
  t.t0 = (rest % p.tiles_t) * BN;
  t.b = rest / p.tiles_t;
  return t;
}

// WM = wave rows of 64 output channels: tile = 64 WM x 128, 2 WM consumer waves + 2 producers
template <int WM>
__global__ __launch_bounds__(128 * WM + 128) void c1b_fwd_ps_kernel(const C1bFwd p) {
  constexpr int TBM = 64 * WM;
  constexpr int AB = TBM * BK * 2;  // A stage bytes
  constexpr int SB = FD_XB + AB;
  constexpr int NCONS = 2 * WM;
  constexpr int RP = WM == 4 ? 16 : 8;  // epilogue rows per pass (LDS: 256 RP bytes per consumer wave)
  extern __shared__ __attribute__((aligned(16))) char fd_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // work items of this XCD: [w_lo, w_hi); this workgroup takes every nwg-th of them
  const int nwg = gridDim.x / NXCD;
  const int w_lo = (blockIdx.x % NXCD) * p.per_xcd, first = w_lo + blockIdx.x / NXCD;
  const int w_hi = w_lo + p.per_xcd < p.total ? w_lo + p.per_xcd : p.total;
  if (first >= w_hi) return;
  const int nmy = (w_hi - first + nwg - 1) / nwg;
  const int nk = p.K / BK;
  const int G = nmy * nk;

  if (wave >= NCONS) {
    // ---------------------------------------------------------------- producers
    const bool px = wave == NCONS;  // X producer; the other one stages A
    const i32x4 rs = px ? fd_rsrc(p.x, (unsigned)(((size_t)(p.B - 1) * p.x_bs + (size_t)p.K * p.T) * 4u))
                        : fd_rsrc(p.a, (unsigned)p.M * p.K * 2u);
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)fd_lds;
    // X: instruction i covers k rows 2 i, 2 i + 1 (lane >> 5), 32 chunks each; rows 8 .. 15 and 24 .. 31 hold
    // their chunks XOR 8.  A: instruction q copies bytes 1024 q .. + 1023 of the stage's block.
    const unsigned xrow = ((unsigned)(lane >> 5) * p.T) * 4u;
    const unsigned xv0 = xrow + (lane & 31) * 16u, xv1 = xrow + ((lane & 31) ^ 8) * 16u;
    const unsigned av = lane * 16u;
    const unsigned xrow2 = 2u * p.T * 4u;
    int it = 0, is = 0, islot = 0;  // next stage to issue: tile index, k stage, ring slot
    FdTile tl = fd_tile<TBM>(p, first);
    auto issue = [&]() {
      const unsigned base = lds0 + islot * SB;
      if (px) {
        const unsigned xo = (unsigned)(((size_t)tl.b * p.x_bs + tl.t0) * 4u) + (unsigned)is * BK * p.T * 4u;
#pragma unroll
        for (int i = 0; i < 16; ++i) fd_dma16(rs, xo + i * xrow2, base + i * 1024, (i & 4) ? xv1 : xv0);
      } else {
        const unsigned ao = ((unsigned)is * p.M + tl.m0) * (BK * 2u);
#pragma unroll
        for (int q = 0; q < AB / 1024; ++q) fd_dma16(rs, ao + q * 1024, base + FD_XB + q * 1024, av);
      }
      islot = islot + 1 == FD_NS ? 0 : islot + 1;
      if (++is == nk) {
        is = 0;
        ++it;
        if (it < nmy) tl = fd_tile<TBM>(p, first + it * nwg);
      }
    };
    issue();
    if (G > 1) issue();
    for (int g = 0; g < G; ++g) {
      // the younger stage may stay in flight: 16 (X) or AB / 1024 (A) DMAs
      if (g + 1 < G) {
        if (px) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AB / 1024) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (g + 2 < G) issue();
    }
    return;
  }

  // ------------------------------------------------------------------ consumers
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, kg = lane >> 5;
  unsigned ao[2], bo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) ao[i] = FD_XB + (wm * 64 + i * 32 + r) * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j) bo[j] = (kg * 8) * 512 + (((wn * 64 + j * 32 + r) ^ (kg << 5)) * 4);
  const int asw = (r >> 2) & 3;

  f32x16 acc[2][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  };
  zero();
  int it = 0, ks = 0, slot = 0;
  for (int g = 0; g < G; ++g) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* st = fd_lds + slot * SB;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(st + ao[i] + (((kk * 2 + kg) ^ asw) * 16));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* col = reinterpret_cast<const float*>(st + bo[j] + kk * 16 * 512);
        uint4 v;
        v.x = pack2(col[0 * 128], col[1 * 128]);
        v.y = pack2(col[2 * 128], col[3 * 128]);
        v.z = pack2(col[4 * 128], col[5 * 128]);
        v.w = pack2(col[6 * 128], col[7 * 128]);
        fb[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    slot = slot + 1 == FD_NS ? 0 : slot + 1;
    if (++ks == nk) {
      ks = 0;
      const FdTile tl = fd_tile<TBM>(p, first + it * nwg);
      ++it;
      // Epilogue.  The MFMA leaves a lane with ONE frame of 16 rows; stored directly that is 64 dword stores
      // per lane whose 128-byte row pieces (8-byte aligned) become 4.4 M L2 write requests of 44 bytes.  Each
      // wave turns RP rows x 64 frames at a time through its own 4 / 2 KB of LDS (no barrier: one wave, LDS
      // operations of a wave execute in order) and stores 16 bytes per lane, 256-byte row pieces.
      float* __restrict__ yb = p.y + (size_t)tl.b * p.y_bs;
      const float* __restrict__ ab = p.acc ? p.acc + (size_t)tl.b * p.acc_bs : nullptr;
      const float* __restrict__ ab2 = p.acc2 ? p.acc2 + (size_t)tl.b * p.acc2_bs : nullptr;
      float* tile = reinterpret_cast<float*>(fd_lds + FD_NS * SB) + wave * (RP * 64);
      const int c4 = (lane & 15) * 4, lrow = lane >> 4;
      const int t = tl.t0 + wn * 64 + c4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int hp = 0; hp < 32 / RP; ++hp) {
#pragma unroll
          for (int e = 0; e < RP / 2; ++e) {
            const int rr = hp * (RP / 2) + e;
            const int row = (rr & 3) + 8 * ((rr >> 2) - hp * (RP / 8)) + 4 * kg;
#pragma unroll
            for (int j = 0; j < 2; ++j) tile[row * 64 + j * 32 + r] = acc[i][j][rr];
          }
#pragma unroll
          for (int q = 0; q < RP / 4; ++q) {
            const int row = lrow + 4 * q;
            const int m = tl.m0 + wm * 64 + i * 32 + hp * RP + row;
            float4 v = *reinterpret_cast<const float4*>(&tile[row * 64 + c4]);
            float add = 0.0f;
            if (p.bias) add += p.bias[m];
            if (p.bias_bc) add += p.bias_bc[(size_t)tl.b * p.M + m];
            const size_t o = (size_t)m * p.T + t;
            if (t + 3 < p.T) {
              if (ab) {
                const f32x4a8 u = *reinterpret_cast<const f32x4a8*>(ab + o);
                v.x += u[0]; v.y += u[1]; v.z += u[2]; v.w += u[3];
              }
              if (ab2) {
                const f32x4a8 u = *reinterpret_cast<const f32x4a8*>(ab2 + o);
                v.x += u[0]; v.y += u[1]; v.z += u[2]; v.w += u[3];
              }
              f32x4a8 out = {v.x + add, v.y + add, v.z + add, v.w + add};
              if (p.relu) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) out[k4] = fmaxf(out[k4], 0.0f);
              }
              *reinterpret_cast<f32x4a8*>(yb + o) = out;
            } else if (t + 1 < p.T) {  // T is even: a row's last piece holds 4 or 2 frames
              if (ab) {
                const f32x2 u = *reinterpret_cast<const f32x2*>(ab + o);
                v.x += u[0]; v.y += u[1];
              }
              if (ab2) {
                const f32x2 u = *reinterpret_cast<const f32x2*>(ab2 + o);
                v.x += u[0]; v.y += u[1];
              }
              f32x2 out = {v.x + add, v.y + add};
              if (p.relu) {
                out[0] = fmaxf(out[0], 0.0f);
                out[1] = fmaxf(out[1], 0.0f);
              }
              *reinterpret_cast<f32x2*>(yb + o) = out;
            }
          }
        }
      }
      zero();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 256 (Cout) x 256 (frames) tile of the same contraction, for M % 256 == 0: 1.5 KB of L2 -> LDS traffic per k
// for 131 kFLOP (85 FLOP per byte; the 256 x 128 tile above: 64), and the X tile is shared by two workgroups
// per 512 output channels as before.  8 waves, each 128 x 64 = 4 x 2 MFMA tiles (128 accumulator registers,
// two waves per SIMD, no producer waves: they would push the kernel to three waves per SIMD and 168
// registers); every wave issues its 6 of the stage's 48 DMAs two stages ahead into a 3-deep ring (144 KB)
// and waits with counted vmcnt.  Its epilogue stores would break that count (loads and stores retire out of
// order with respect to each other), so the first wait of the next tile is vmcnt(0) - once per 16 stages.
// The epilogue's transposition stage lives in the ring slot the tile's last stage just vacated.
constexpr int F2_XB = BK * 256 * 4;  // 32 KB: X stage, fp32 [32 k][256 t]
constexpr int F2_AB = 256 * BK * 2;  // 16 KB: A stage, bf16 [256 m][32 k]
constexpr int F2_SB = F2_XB + F2_AB;
constexpr int F2_LDS = FD_NS * F2_SB;

__global__ __launch_bounds__(512) void c1b_fwd_ps2_kernel(const C1bFwd p) {
  extern __shared__ __attribute__((aligned(16))) char fd_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x / NXCD;
  const int w_lo = (blockIdx.x % NXCD) * p.per_xcd, first = w_lo + blockIdx.x / NXCD;
  const int w_hi = w_lo + p.per_xcd < p.total ? w_lo + p.per_xcd : p.total;
  if (first >= w_hi) return;
  const int nmy = (w_hi - first + nwg - 1) / nwg;
  const int nk = p.K / BK;
  const int G = nmy * nk;

  auto tile_of = [&](int work) {
    FdTile t;
    const int mt = work % p.tiles_m, rest = work / p.tiles_m;
    t.m0 = mt * 256;
    t.t0 = (rest % p.tiles_t) * 256;
    t.b = rest / p.tiles_t;
    return t;
  };

  // DMA roles: wave w stages X rows 4 w .. 4 w + 3 (one 1 KB row per instruction) and 2 KB of the A block
  const i32x4 xrs = fd_rsrc(p.x, (unsigned)(((size_t)(p.B - 1) * p.x_bs + (size_t)p.K * p.T) * 4u));
  const i32x4 ars = fd_rsrc(p.a, (unsigned)p.M * p.K * 2u);
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)fd_lds;
  const unsigned lv = lane * 16u;
  const unsigned xrow = (unsigned)p.T * 4u;
  int it_i = 0, is = 0, islot = 0;
  FdTile ti = tile_of(first);
  auto issue = [&]() {
    const unsigned base = lds0 + islot * F2_SB;
    const unsigned xo = __builtin_amdgcn_readfirstlane(
        (unsigned)(((size_t)ti.b * p.x_bs + ti.t0) * 4u) + (unsigned)(is * BK + 4 * wave) * xrow);
    const unsigned ao = __builtin_amdgcn_readfirstlane(((unsigned)is * p.M + ti.m0) * (BK * 2u) + wave * 2048u);
#pragma unroll
    for (int q = 0; q < 4; ++q) fd_dma16(xrs, xo + q * xrow, base + (4 * wave + q) * 1024, lv);
#pragma unroll
    for (int q = 0; q < 2; ++q) fd_dma16(ars, ao + q * 1024, base + F2_XB + wave * 2048 + q * 1024, lv);
    islot = islot + 1 == FD_NS ? 0 : islot + 1;
    if (++is == nk) {
      is = 0;
      ++it_i;
      if (it_i < nmy) ti = tile_of(first + it_i * nwg);
    }
  };

  const int wm = wave & 1, wn = wave >> 1;
  const int r = lane & 31, kg = lane >> 5;
  const int asw = (r >> 2) & 3;
  unsigned ao4[4], bo2[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) ao4[i] = F2_XB + (wm * 128 + i * 32 + r) * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j) bo2[j] = (kg * 8) * 1024 + (wn * 64 + j * 32 + r) * 4;

  f32x16 acc[4][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  };
  zero();
  issue();
  if (G > 1) issue();
  int slot = 0, g = 0;
  for (int it = 0; it < nmy; ++it) {
    const FdTile tl = tile_of(first + it * nwg);
    const int b = tl.b, m0 = tl.m0, t0 = tl.t0;
    for (int ks = 0; ks < nk; ++ks, ++g) {
      if (g + 1 < G && !(ks == 0 && it > 0)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (g + 2 < G) issue();
      const char* st = fd_lds + slot * F2_SB;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 fa[4], fb[2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          fa[i] = *reinterpret_cast<const bf16x8*>(st + ao4[i] + (((kk * 2 + kg) ^ asw) * 16));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float* col = reinterpret_cast<const float*>(st + bo2[j] + kk * 16 * 1024);
          uint4 v;
          v.x = pack2(col[0 * 256], col[1 * 256]);
          v.y = pack2(col[2 * 256], col[3 * 256]);
          v.z = pack2(col[4 * 256], col[5 * 256]);
          v.w = pack2(col[6 * 256], col[7 * 256]);
          fb[j] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      if (ks + 1 < nk) slot = slot + 1 == FD_NS ? 0 : slot + 1;
    }
    // epilogue through the slot the last stage just vacated (nothing is issued into it before the next
    // iteration's barrier): 16 rows x 64 frames per wave and pass, 16-byte stores
    __syncthreads();
    float* tile = reinterpret_cast<float*>(fd_lds + slot * F2_SB) + wave * (16 * 64);
    slot = slot + 1 == FD_NS ? 0 : slot + 1;
    float* __restrict__ yb = p.y + (size_t)b * p.y_bs;
    const float* __restrict__ ab = p.acc ? p.acc + (size_t)b * p.acc_bs : nullptr;
    const float* __restrict__ ab2 = p.acc2 ? p.acc2 + (size_t)b * p.acc2_bs : nullptr;
    const int c4 = (lane & 15) * 4, lrow = lane >> 4;
    const int t = t0 + wn * 64 + c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int rr = hp * 8 + e;
          const int row = (rr & 3) + 8 * ((rr >> 2) - hp * 2) + 4 * kg;
#pragma unroll
          for (int j = 0; j < 2; ++j) tile[row * 64 + j * 32 + r] = acc[i][j][rr];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = lrow + 4 * q;
          const int m = m0 + wm * 128 + i * 32 + hp * 16 + row;
          float4 v = *reinterpret_cast<const float4*>(&tile[row * 64 + c4]);
          float add = 0.0f;
          if (p.bias) add += p.bias[m];
          if (p.bias_bc) add += p.bias_bc[(size_t)b * p.M + m];
          const size_t o = (size_t)m * p.T + t;
          if (t + 3 < p.T) {
            if (ab) {
              const f32x4a8 u = *reinterpret_cast<const f32x4a8*>(ab + o);
              v.x += u[0]; v.y += u[1]; v.z += u[2]; v.w += u[3];
            }
            if (ab2) {
              const f32x4a8 u = *reinterpret_cast<const f32x4a8*>(ab2 + o);
              v.x += u[0]; v.y += u[1]; v.z += u[2]; v.w += u[3];
            }
            f32x4a8 out = {v.x + add, v.y + add, v.z + add, v.w + add};
            if (p.relu) {
#pragma unroll
              for (int k4 = 0; k4 < 4; ++k4) out[k4] = fmaxf(out[k4], 0.0f);
            }
            *reinterpret_cast<f32x4a8*>(yb + o) = out;
          } else if (t + 1 < p.T) {  // T is even: a row's last piece holds 4 or 2 frames
            if (ab) {
              const f32x2 u = *reinterpret_cast<const f32x2*>(ab + o);
              v.x += u[0]; v.y += u[1];
            }
            if (ab2) {
              const f32x2 u = *reinterpret_cast<const f32x2*>(ab2 + o);
              v.x += u[0]; v.y += u[1];
            }
            f32x2 out = {v.x + add, v.y + add};
            if (p.relu) {
              out[0] = fmaxf(out[0], 0.0f);
              out[1] = fmaxf(out[1], 0.0f);
            }
            *reinterpret_cast<f32x2*>(yb + o) = out;
          }
        }
      }
    }
    zero();
  }
}

int fd_launch2(C1bFwd& p, hipStream_t st) {
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(c1b_fwd_ps2_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS) == hipSuccess;
  if (!attr_ok) return AIR_ELAUNCH;
  p.tiles_m = p.M / 256;
  p.tiles_t = (p.T + 255) / 256;
  p.total = p.B * p.tiles_t * p.tiles_m;
  p.per_xcd = (p.total + NXCD - 1) / NXCD;
  hipLaunchKernelGGL(c1b_fwd_ps2_kernel, dim3(256), dim3(512), F2_LDS, st, p);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

template <int WM>
int fd_launch(C1bFwd& p, hipStream_t st) {
  constexpr int lds = FD_NS * (FD_XB + 64 * WM * BK * 2) + 2 * WM * (WM == 4 ? 16 : 8) * 256;
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(c1b_fwd_ps_kernel<WM>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
  if (!attr_ok) return AIR_ELAUNCH;
  p.tiles_m = p.M / (64 * WM);
  p.total = p.B * p.tiles_t * p.tiles_m;
  p.per_xcd = (p.total + NXCD - 1) / NXCD;
  // resident workgroups: 128 KB of LDS -> one per CU, 80 KB -> two
  hipLaunchKernelGGL(c1b_fwd_ps_kernel<WM>, dim3(WM == 4 ? 256 : 512), dim3(128 * WM + 128), lds, st, p);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

// dw[e] = sum_s partial[s][e], fixed order (deterministic)
__global__ __launch_bounds__(256) void c1b_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                         size_t n, int nsplit) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n / 4; e += (size_t)gridDim.x * 256) {
    float4 s = reinterpret_cast<const float4*>(partial)[e];
    int k = 1;
    for (; k + 7 < nsplit; k += 8) {  // eight loads in flight, summed in index order
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(partial + (size_t)(k + u) * n)[e];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
      }
    }
    for (; k < nsplit; ++k) {
      const float4 v = reinterpret_cast<const float4*>(partial + (size_t)k * n)[e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(dw)[e] = s;
  }
}


// ===================================================================================
// Path 2: bf16-resident operands + LDS-DMA GEMM (the wide layers and every weight gradient).
//
// The fused kernels above stage fp32 and convert in flight; that costs VALU + ds_write work per
// stage and L2 -> LDS traffic at 4 bytes per operand element, which bounds them once K and M are
// large (layer4: 1536 x 1536, measured 330-440 TF) and for every weight gradient (both operands
// re-staged by each tile column/row: 130 TF).  Here one HBM-bound pass per operand writes a bf16
// copy in the layout the MFMA wants - K contiguous, K padded with zeros to a multiple of 64, rows
// padded to the tile - and ONE GEMM kernel, C[m][n] = sum_k A[m][k] B[n][k], serves all three
// passes with `global_load_lds_dwordx4`: 16-byte chunks go global -> LDS directly, no staging
// registers, no conversion, no ds_write.
//   forward  A = W   [Cout][Cin]        B = X^T_b  [Tp][Cin]    (c1b_cvt_t_kernel)
//   dgrad    A = W^T [Cin][Cout]        B = dY^T_b [Tp][Cout]
//   wgrad    A = dY  [Cout][(b, Tp)]    B = X      [Cin][(b, Tp)]   (c1b_cvt_s_kernel), split-K over b
// LDS tile rows are 128 bytes (64 bf16) with the 16-byte chunk index XOR-ed by (row >> 1) & 7 - the
// swizzle is applied to the SOURCE address of the DMA (its LDS destination is lane-linear) and again
// by the reader, so the 16 lanes of a ds_read_b128 cycle cover all 64 banks.

constexpr int GK = 64;         // K per stage
constexpr int TP_ALIGN = 128;  // padded frame count: multiple of the N tile and of GK

typedef float f32x2a8 __attribute__((ext_vector_type(2), aligned(8)));

// X (B, C, T) fp32 (batch stride xbs) -> Xs bf16 [B][C][Tp], zeros for t >= T.  One thread = 8 frames.
__global__ __launch_bounds__(256) void c1b_cvt_s_kernel(const float* __restrict__ x, size_t xbs, int C, int T, int Tp,
                                                        size_t total_chunks, int vec_ok,
                                                        unsigned short* __restrict__ out) {
  const int cpr = Tp / 8;  // chunks per row
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total_chunks; e += (size_t)gridDim.x * 256) {
    const size_t row = e / cpr;  // b * C + c
    const int t = (int)(e - row * cpr) * 8;
    const size_t b = row / C, c = row - b * C;
    const float* __restrict__ src = x + b * xbs + c * (size_t)T;
    float v[8];
    if (vec_ok && t + 7 < T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2a8 w = *reinterpret_cast<const f32x2a8*>(src + t + 2 * q);
        v[2 * q] = w[0];
        v[2 * q + 1] = w[1];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = t + q < T ? src[t + q] : 0.0f;
    }
    uint4 o;
    o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
    reinterpret_cast<uint4*>(out)[e] = o;
  }
}

// X (B, C, T) fp32 -> Xt bf16 [B][Tp][C] (channels contiguous), zeros for t >= T, through an LDS transposition so
// that BOTH sides move full lines: a workgroup owns 64 channels x 64 frames; reads are 16 bytes per lane along t
// (a wave: 4 channel rows x 256 B), the rounded values go to LDS as [t][c] bf16 (rows padded to 132 bytes), and
// each lane writes 32 contiguous bytes of a transposed row - 4 lanes cover the tile's 128-byte row piece, a wave
// 16 rows.  (Round 1's version wrote 32-byte pieces straight from registers: four partial requests per line.)
// grid (Tp / 64, C / 64, B), 256 threads.
__global__ __launch_bounds__(256) void c1b_cvt_t_kernel(const float* __restrict__ x, size_t xbs, int C, int T, int Tp,
                                                        int vec_ok, unsigned short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 66];  // [t][c], 66 elements per row
  const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const size_t b = blockIdx.z;
  const int tid = threadIdx.x;
  {
    const int q = tid & 15, cr = tid >> 4;  // quad of frames, channel row (+ 16 j)
    const int t = t0 + 4 * q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cr + 16 * j;
      const float* __restrict__ src = x + b * xbs + (size_t)(c0 + c) * T + t;
      float v[4];
      if (vec_ok && t + 3 < T) {
        const f32x4a8 w = *reinterpret_cast<const f32x4a8*>(src);
        v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t + e < T ? src[e] : 0.0f;
      }
      const unsigned p01 = pack2(v[0], v[1]), p23 = pack2(v[2], v[3]);
      tile[(4 * q + 0) * 66 + c] = (unsigned short)(p01 & 0xffffu);
      tile[(4 * q + 1) * 66 + c] = (unsigned short)(p01 >> 16);
      tile[(4 * q + 2) * 66 + c] = (unsigned short)(p23 & 0xffffu);
      tile[(4 * q + 3) * 66 + c] = (unsigned short)(p23 >> 16);
    }
  }
  __syncthreads();
  {
    const int tr = tid >> 2, cq = (tid & 3) * 16;  // frame row, 16-channel piece
    const unsigned* __restrict__ row = reinterpret_cast<const unsigned*>(&tile[tr * 66 + cq]);
    uint4 o0, o1;
    o0.x = row[0]; o0.y = row[1]; o0.z = row[2]; o0.w = row[3];
    o1.x = row[4]; o1.y = row[5]; o1.z = row[6]; o1.w = row[7];
    uint4* dst = reinterpret_cast<uint4*>(out + (b * Tp + t0 + tr) * (size_t)C + c0 + cq);
    dst[0] = o0;
    dst[1] = o1;
  }
}

struct NtGemm {
  const unsigned short* a;  // A[m][seg][k]: a + m * a_rs + seg * a_ss + k   (+ batch * a_bs)
  const unsigned short* b;  // B[n][seg][k]
  float* out;               // out[batch][m][n]: out + batch * o_bs + m * o_rs + n
  const float* bias;        // [M]
  const float* bias_bc;     // [batch][M]
  const float* acc;         // out += acc[batch * acc_bs + m * o_rs + n] (+ acc2 likewise; c1b_gemm_ps_kernel: own batch strides)
  const float* acc2;
  size_t acc_bs, acc2_bs;
  unsigned short* out_bf;   // optional bf16 copy of out: out_bf + (batch * M + m) * out_bf_rs + n (c1b_gemm_ps_kernel only)
  int out_bf_rs;
  // bf16-RESIDENT output (c1b_gemm_ps_kernel<BKM, true>: `out` unused): out_bf + batch * out_bf_bs + m * out_bf_rs + n,
  // frames n >= n_valid written as zeros; accumulate operands are bf16 rows of the same row pitch with their own
  // batch strides; rows m >= m_valid (an M below the 256-row tile: the A operand reads zeros there) are not written
  size_t out_bf_bs;
  const unsigned short* acc_h;
  const unsigned short* acc2_h;
  size_t acc_h_bs, acc2_h_bs;
  int m_valid;
  size_t a_rs, a_ss, a_bs, b_rs, b_ss, b_bs, o_rs, o_bs;
  int M, n_valid, kseg, nseg_per_batch, nseg_total, relu, tiles_m, tiles_n, total, per_xcd;
  // c1b_gemm_ps_kernel<BKM, true>, optional: BatchNorm statistics of the STORED output - per (row m, 64-frame segment r)
  // the pair {sum, sum of squares} at hstats[(m * NR + r) * 2], NR = batches * tiles_n * 4, r = (batch * tiles_n + tile) * 4 + wn
  float* hstats;
};

typedef const __attribute__((address_space(1))) void* c1b_gptr;
typedef __attribute__((address_space(3))) void* c1b_lptr;

__global__ __launch_bounds__(256) void c1b_gemm_kernel(const NtGemm p) {
  __shared__ __attribute__((aligned(1024))) unsigned short sA[BM * GK];
  __shared__ __attribute__((aligned(1024))) unsigned short sB[BN * GK];
  const int work = xcd_chunked(blockIdx.x, p.per_xcd);
  if (work >= p.total) return;
  const int mt = work % p.tiles_m;
  const int rest = work / p.tiles_m;
  const int nt = rest % p.tiles_n, batch = rest / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int seg_lo = batch * p.nseg_per_batch;
  const int nseg = min(p.nseg_per_batch, p.nseg_total - seg_lo);

  // DMA roles: wave w issues 4 + 4 instructions per stage; instruction i covers tile rows
  // (4 w + i) * 8 .. + 7, lane l -> row + (l >> 3), LDS chunk position l & 7
  const int rsub = lane >> 3, pos = lane & 7;
  const unsigned short* ga[4];
  const unsigned short* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + rsub;
    const int chunk = pos ^ ((row >> 1) & 7);
    ga[i] = p.a + (size_t)batch * p.a_bs + (size_t)(m0 + row) * p.a_rs + chunk * 8;
    gb[i] = p.b + (size_t)batch * p.b_bs + (size_t)(n0 + row) * p.b_rs + chunk * 8;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int r = lane & 31, kg = lane >> 5;
  const int sw = (r >> 1) & 7;  // tile row = 64 wm + 32 i + r: only r enters (row >> 1) & 7
  for (int seg = 0; seg < nseg; ++seg) {
    const size_t aoff = (size_t)seg * p.a_ss, boff = (size_t)seg * p.b_ss;
    for (int k0 = 0; k0 < p.kseg; k0 += GK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((c1b_gptr)(ga[i] + aoff + k0), (c1b_lptr)(sA + (wave * 4 + i) * 8 * GK), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((c1b_gptr)(gb[i] + boff + k0), (c1b_lptr)(sB + (wave * 4 + i) * 8 * GK), 16, 0, 0);
      }
      __syncthreads();  // drains the DMAs (vmcnt(0)) and publishes the tile
#pragma unroll
      for (int kk = 0; kk < GK / 16; ++kk) {
        const int cpos = ((kk * 2 + kg) ^ sw) * 8;
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sA + (wm * 64 + i * 32 + r) * GK + cpos);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sB + (wn * 64 + j * 32 + r) * GK + cpos);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();  // everyone is done reading before the next stage overwrites the tile
    }
  }

  const int col = lane & 31, half = lane >> 5;
  float* __restrict__ ob = p.out + (size_t)batch * p.o_bs;
  const float* __restrict__ ab = p.acc ? p.acc + (size_t)batch * p.o_bs : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
      float add = 0.0f;
      if (p.bias) add += p.bias[m];
      if (p.bias_bc) add += p.bias_bc[(size_t)batch * p.M + m];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + col;
        if (n < p.n_valid) {
          float v = acc[i][j][q] + add;
          if (ab) v += ab[(size_t)m * p.o_rs + n];
          if (p.relu) v = fmaxf(v, 0.0f);
          ob[(size_t)m * p.o_rs + n] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// The same NT GEMM, persistent, 256 x 256 tiles (used whenever M % 256 == 0 and N % 256 == 0; the kernel
// above serves the 128-row operands of the attention layers).  rocprofv3 on c1b_gemm_kernel, layer4
// forward (0.71 ms, 640 TF): 54 M L2 requests, 36 % of them misses (the 4.7 MB of weights plus the X
// tiles of the workgroups in flight do not fit an XCD's 4 MB), 535 cycles mean read latency, ~50 lines
// outstanding per CU all the time: the tile is bound by what one CU's L1 can have in flight, i.e. by
// L2 -> LDS BYTES, and a 128 x 128 tile moves 1 byte per 64 FLOP.  256 x 256 halves that.
// 8 waves, each 128 x 64 of the tile = 4 x 2 MFMA 32x32x16 tiles (128 accumulator registers, two waves
// per SIMD); every wave issues its eighth of the stage's 64 `buffer_load_dwordx4 ... lds` (K = 64: full
// 128-byte lines per row) one stage ahead into a 2-deep ring (128 KB), one barrier per stage.  A wave
// that mixes DMA loads and epilogue stores can only wait for vmcnt(0) (they retire out of order with
// respect to each other) - that is exactly what the loop does: at its top only the next stage's DMAs
// (and, once per tile, the epilogue's stores) are outstanding.  Resident workgroups (one per CU) walk
// their XCD's tiles; wave-private transposed epilogue as in c1b_fwd_ps_kernel.
constexpr int G2_BM = 256, G2_BN = 256;
constexpr int G2_AB = G2_BM * GK * 2;  // 32 KB
constexpr int G2_BB = G2_BN * GK * 2;  // 32 KB
constexpr int G2_SB = G2_AB + G2_BB;
constexpr int G2_RP = 8;
constexpr int G2_LDS = 2 * G2_SB + 8 * G2_RP * 256;

struct G2Tile {
  int batch, m0, n0, nseg;
};
__device__ __forceinline__ G2Tile g2_tile(const NtGemm& p, int work) {
  G2Tile t;
  const int mt = work % p.tiles_m, rest = work / p.tiles_m;
  t.m0 = mt * G2_BM;
  t.n0 = (rest % p.tiles_n) * G2_BN;
  t.batch = rest / p.tiles_n;
  t.nseg = min(p.nseg_per_batch, p.nseg_total - t.batch * p.nseg_per_batch);
  return t;
}

// bf16-resident epilogue of c1b_gemm_ps_kernel<BKM, true>: 8 rows x 64 columns at a time through the wave's own 2 KB
// of LDS; a lane owns 8 consecutive frames of one row = one 16-byte store (a wave: eight 128-byte row pieces).  Per
// 32-row group the bias values and the accumulate operands of its four 8-row pieces are requested up front (one
// exposed round trip per group instead of one per operand per piece); the number of accumulate operands and the
// ReLU are template parameters picked by a uniform branch (as runtime flags the compiler evaluates every variant
// and selects: 130 VALU instructions per piece instead of 40 - the epilogue, not the k-loop, was the longer half).
template <int NACC, bool RELU>
__device__ __forceinline__ void g2_hout_epilogue(const NtGemm& p, const G2Tile& tl, f32x16 (&acc)[4][2], float* tile,
                                                 int lane, int wm, int wn) {
  const int r = lane & 31, kg = lane >> 5;
  const int c8 = (lane & 7) * 8, lrow = lane >> 3;
  const int n = tl.n0 + wn * 64 + c8;
  u16* __restrict__ ob = p.out_bf + (size_t)tl.batch * p.out_bf_bs;
  const u16* __restrict__ ab = p.acc_h ? p.acc_h + (size_t)tl.batch * p.acc_h_bs : p.acc2_h + (size_t)tl.batch * p.acc2_h_bs;
  const u16* __restrict__ ab2 = p.acc2_h + (size_t)tl.batch * p.acc2_h_bs;
  const bool mask_n = tl.n0 + G2_BN > p.n_valid;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mg = tl.m0 + wm * 128 + i * 32 + lrow;  // piece hp: row mg + 8 hp
    float add[4];
    uint4 ua[4], ub[4];
#pragma unroll
    for (int hp = 0; hp < 4; ++hp) {
      const int m = mg + hp * 8;
      const bool mv = m < p.m_valid;
      add[hp] = 0.0f;
      if (p.bias && mv) add[hp] = p.bias[m];
      if (p.bias_bc && mv) add[hp] += p.bias_bc[(size_t)tl.batch * p.M + m];
      if (NACC >= 1) {
        ua[hp] = make_uint4(0u, 0u, 0u, 0u);
        if (mv) ua[hp] = *reinterpret_cast<const uint4*>(ab + (size_t)m * p.out_bf_rs + n);
      }
      if (NACC >= 2) {
        ub[hp] = make_uint4(0u, 0u, 0u, 0u);
        if (mv) ub[hp] = *reinterpret_cast<const uint4*>(ab2 + (size_t)m * p.out_bf_rs + n);
      }
    }
#pragma unroll
    for (int hp = 0; hp < 4; ++hp) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rr = hp * 4 + e;
#pragma unroll
        for (int j = 0; j < 2; ++j) tile[(e + 4 * kg) * 64 + j * 32 + r] = acc[i][j][rr];
      }
      const int m = mg + hp * 8;
      const float4 v0 = *reinterpret_cast<const float4*>(&tile[lrow * 64 + c8]);
      const float4 v1 = *reinterpret_cast<const float4*>(&tile[lrow * 64 + c8 + 4]);
      f32x2 vv[4] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w}};
      const f32x2 a2 = {add[hp], add[hp]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        vv[q] += a2;
        if (NACC >= 1) {
          const unsigned w = q == 0 ? ua[hp].x : q == 1 ? ua[hp].y : q == 2 ? ua[hp].z : ua[hp].w;
          vv[q] += f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
        }
        if (NACC >= 2) {
          const unsigned w = q == 0 ? ub[hp].x : q == 1 ? ub[hp].y : q == 2 ? ub[hp].z : ub[hp].w;
          vv[q] += f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
        }
        if (RELU) vv[q] = __builtin_elementwise_max(vv[q], f32x2{0.0f, 0.0f});
      }
      if (mask_n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (n + 2 * q >= p.n_valid) vv[q][0] = 0.0f;
          if (n + 2 * q + 1 >= p.n_valid) vv[q][1] = 0.0f;
        }
      }
      const uint4 ow = make_uint4(pack2(vv[0][0], vv[0][1]), pack2(vv[1][0], vv[1][1]), pack2(vv[2][0], vv[2][1]),
                                  pack2(vv[3][0], vv[3][1]));
#ifdef G2_X_NOSTORE
      if (ow.x == 0x12345678u)
#endif
      if (m < p.m_valid) *reinterpret_cast<uint4*>(ob + (size_t)m * p.out_bf_rs + n) = ow;
      if (p.hstats != nullptr) {  // (wave-uniform) of the 8 values as stored; the row's 8 lanes are neighbours
        const unsigned wd[4] = {ow.x, ow.y, ow.z, ow.w};
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = __builtin_bit_cast(float, wd[q] << 16), hi = __builtin_bit_cast(float, wd[q] & 0xffff0000u);
          s1 += lo + hi;
          s2 = fmaf(lo, lo, s2);
          s2 = fmaf(hi, hi, s2);
        }
        // (DPP: lane ^ 1, lane ^ 2, mirror within the row's 8 lanes - no trip through LDS as __shfl_xor makes)
#define G2_DPP_ADD(v, ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
        G2_DPP_ADD(s1, 0xB1); G2_DPP_ADD(s2, 0xB1);
        G2_DPP_ADD(s1, 0x4E); G2_DPP_ADD(s2, 0x4E);
        G2_DPP_ADD(s1, 0x141); G2_DPP_ADD(s2, 0x141);
#undef G2_DPP_ADD
        if ((lane & 7) == 0 && m < p.m_valid) {
          const int NR = p.nseg_total * p.tiles_n * 4, ridx = (tl.batch * p.tiles_n + tl.n0 / G2_BN) * 4 + wn;
          *reinterpret_cast<float2*>(p.hstats + ((size_t)m * NR + ridx) * 2) = make_float2(s1, s2);
        }
      }
    }
  }
}

// BKM: the B operand is K-MAJOR - element (k, n) at b + k * b_rs + n, n contiguous - i.e. an activation's bf16 copy
// [b][channel][Tp] as the producers write it and the weight gradient reads it; no transposed copy (c1b_cvt_t) is
// made for the forward / dgrad GEMM.  Its stage is [64 k][256 n] (512-byte rows, 16-byte chunk c of row k at
// position c ^ ((k & 3) << 2)) and a lane assembles its 8 consecutive k with two `ds_read_b64_tr_b16`: the 16
// lanes of a group point at a [4 k][16 n] block, 4 contiguous n each, and receive one column of it
// (tools/ubench/tr_read.hip prints the mapping).
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <bool BKM, bool HOUT = false>
__global__ __launch_bounds__(512) void c1b_gemm_ps_kernel(const NtGemm p, unsigned a_bytes, unsigned b_bytes) {
  extern __shared__ __attribute__((aligned(1024))) char g2_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x / NXCD;
  const int w_lo = (blockIdx.x % NXCD) * p.per_xcd, first = w_lo + blockIdx.x / NXCD;
  const int w_hi = w_lo + p.per_xcd < p.total ? w_lo + p.per_xcd : p.total;
  if (first >= w_hi) return;
  const int nmy = (w_hi - first + nwg - 1) / nwg;
  const int kst = p.kseg / GK;  // stages per segment

  // DMA roles: wave w stages tile rows 32 w .. 32 w + 31 of A and of B, 4 + 4 instructions per stage;
  // instruction q covers 8 rows (lane >> 3), LDS chunk position lane & 7 holds source chunk pos ^ ((row >> 1) & 7)
  const i32x4 ars = fd_rsrc(p.a, a_bytes), brs = fd_rsrc(p.b, b_bytes);
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)g2_lds;
  const int r8 = lane >> 3, pos = lane & 7;
  const unsigned av0 = (unsigned)((r8 * p.a_rs + ((pos ^ (r8 >> 1)) * 8)) * 2);
  const unsigned av1 = (unsigned)((r8 * p.a_rs + ((pos ^ (r8 >> 1) ^ 4) * 8)) * 2);
  // B, K-major: instruction q of wave w covers k rows 8 w + 2 q + (lane >> 5), 32 chunks each
  const unsigned bv0 = BKM ? (unsigned)((lane >> 5) * p.b_rs * 2 + (((lane & 31) ^ ((lane >> 5) << 2)) * 16))
                           : (unsigned)((r8 * p.b_rs + ((pos ^ (r8 >> 1)) * 8)) * 2);
  const unsigned bv1 = BKM ? (unsigned)((lane >> 5) * p.b_rs * 2 + (((lane & 31) ^ ((2 + (lane >> 5)) << 2)) * 16))
                           : (unsigned)((r8 * p.b_rs + ((pos ^ (r8 >> 1) ^ 4) * 8)) * 2);
  const unsigned arow8 = (unsigned)(8 * p.a_rs * 2), brow8 = (unsigned)((BKM ? 2 : 8) * p.b_rs * 2);
  int it_i = 0, iseg = 0, ik = 0, islot = 0;  // next stage to issue
  G2Tile ti = g2_tile(p, first);
  // One stage = 64 DMA instructions (1 KB each: 32 of A, 32 of B), ALL issued by waves 0 - 3 - one "loader" per
  // SIMD, 16 instructions each - right behind the stage barrier, while waves 4 - 7 (their SIMD partners, raised to
  // priority 1) start the stage's MFMAs at once.  An LDS-DMA instruction holds the issuing wave for ~60 cycles
  // (16 = ~1000 cycles, s_memtime trace): with every wave issuing its eighth of the stage, as rounds 3 - 4 had it,
  // no wave of the workgroup had an MFMA to offer for the first ~600 cycles of a 3400-cycle stage.  Spreading the
  // instructions between the MFMAs instead costs their lead time (layer4: 496 us against 438).
  // profiles/r05_gemm_kloop.md has the stage traces and the clock the kernel really runs at.
  unsigned d_base = 0, d_ao = 0, d_bo = 0;
  constexpr int G2_NP = 8, G2_WR = 64;  // pieces of A (and of B) per loader wave and stage; tile rows per loader wave
  auto issue_setup = [&]() {
    d_base = lds0 + islot * G2_SB + wave * (G2_NP * 1024);
    const size_t koff = (size_t)iseg * p.a_ss + (size_t)ik * GK, koffb = (size_t)iseg * p.b_ss + (size_t)ik * GK;
    d_ao = (unsigned)(((size_t)ti.batch * p.a_bs + (size_t)(ti.m0 + G2_WR * wave) * p.a_rs + koff) * 2);
    d_bo = BKM ? (unsigned)(((size_t)ti.batch * p.b_bs + (size_t)iseg * p.b_ss +
                             ((size_t)ik * GK + (G2_WR / 4) * wave) * p.b_rs + ti.n0) * 2)
               : (unsigned)(((size_t)ti.batch * p.b_bs + (size_t)(ti.n0 + G2_WR * wave) * p.b_rs + koffb) * 2);
  };
  auto issue_piece = [&](const int q) {
#ifndef G2_X_NODMA
    if (q < G2_NP) {
#ifndef G2_X_NOA
      fd_dma16(ars, d_ao + q * arow8, d_base + q * 1024, (q & 1) ? av1 : av0);
#endif
    } else {
#ifndef G2_X_NOB
      fd_dma16(brs, d_bo + (q - G2_NP) * brow8, d_base + G2_AB + (q - G2_NP) * 1024, (q & 1) ? bv1 : bv0);
#endif
    }
#endif
  };
  auto issue_advance = [&]() {
    islot ^= 1;
    if (++ik == kst) {
      ik = 0;
      if (++iseg == ti.nseg) {
        iseg = 0;
        ++it_i;
        if (it_i < nmy) ti = g2_tile(p, first + it_i * nwg);
      }
    }
  };
  auto issue_all = [&]() {
    if (wave < 4) {  // one loader wave per SIMD: its partner starts the stage's MFMAs while it feeds the DMA queue
      issue_setup();
#pragma unroll
      for (int q = 0; q < 2 * G2_NP; ++q) issue_piece(q);
    }
  };
  auto issue = [&]() {
    issue_all();
    issue_advance();
  };

  const int wm = wave & 1, wn = wave >> 1;
  const int r = lane & 31, kg = lane >> 5;
  const int sw = (r >> 1) & 7;
  f32x16 acc[4][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  };
  zero();
#ifdef G2_X_STAGGER
  {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long d = (unsigned long long)((blockIdx.x / NXCD) % 3) * G2_X_STAGGER;
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
#endif
  issue();
  // the partner of a loader wave issues its MFMAs while the loader feeds the DMA queue; the loader is the OLDER wave of
  // the SIMD and would otherwise take the matrix pipe back as soon as it gets there, leaving the partner - and with it
  // the barrier - 1000 cycles behind (stage trace in profiles/r05_gemm_kloop.md)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  int slot = 0;
#ifdef G2_X_TRACE
  int trace_n = 0;
  unsigned long long trace_sum[3] = {0, 0, 0}, trace_last = 0;  // per workgroup and wave: total wait / dma issue / mfma clocks
  const unsigned long long trace_t0 = __builtin_readcyclecounter(), trace_w0 = wall_clock64();
#endif
  for (int it = 0; it < nmy; ++it) {
    const G2Tile tl = g2_tile(p, first + it * nwg);
    const int ng = tl.nseg * kst;
    for (int g = 0; g < ng; ++g) {
#ifdef G2_X_TRACE
      // timing build: s_memtime at four points of the first 128 stages of workgroup 0, waves 0 and 4, into the
      // (oversized) bias buffer: tools/kbench_h_gemm.py KB_TRACE=1 prints the per-stage deltas
      unsigned long long* trc = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + 1024 + (wave >> 2) * 512;
      const bool tron = blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && trace_n < 128;
#define G2_TR(j) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); \
                      if (tron) trc[trace_n * 4 + j] = now_; if (j > 0) trace_sum[j - 1] += now_ - trace_last; trace_last = now_; \
                      __builtin_amdgcn_sched_barrier(0); } while (0)
      G2_TR(0);
#else
#define G2_TR(j)
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef G2_X_NOBAR
      __syncthreads();  // this stage has landed everywhere; everyone is done with the previous one
#endif
      const unsigned short* sA = reinterpret_cast<const unsigned short*>(g2_lds + slot * G2_SB);
      const unsigned short* sB = reinterpret_cast<const unsigned short*>(g2_lds + slot * G2_SB + G2_AB);
#ifdef G2_X_NOMFMA
      if (g + 1 < ng || it + 1 < nmy) issue();
      if (p.kseg < 0)
#endif
      {
        // Fragments of k-step kk + 1 are requested while the MFMAs of k-step kk issue (two register sets), one LDS
        // instruction (pair, for the transposed B reads) behind each of the first six MFMAs.  (Left to the compiler
        // every pair of MFMAs waited for its own just-issued ds_reads.  tools/ubench/g2_loop.hip, the loop alone:
        // 1690 TF with the reads as a burst per k-step, 1830 one by one; in the kernel the two are equal.)
        bf16x8 fa[2][4], fb[2][2];
        auto rdA = [&](bf16x8& q, const int kk, const int i) {
          const int cpos = ((kk * 2 + kg) ^ sw) * 8;
          q = *reinterpret_cast<const bf16x8*>(sA + (wm * 128 + i * 32 + r) * GK + cpos);
        };
        auto rdB = [&](bf16x8& q, const int kk, const int j) {
          if (BKM) {
            const int i16 = lane & 15, gsel = (lane >> 4) & 1;
            const int nl = wn * 64 + j * 32 + gsel * 16 + 4 * (i16 & 3);  // first of this lane's 4 columns
            s16x4 h[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int row = kk * 16 + kg * 8 + 4 * hh + (i16 >> 2);
              const unsigned short* src = sB + row * 256 + (((nl >> 3) ^ ((i16 >> 2) << 2)) << 3) + (nl & 4);
              h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)src);
            }
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 both = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
            q = __builtin_bit_cast(bf16x8, both);
          } else {
            const int cpos = ((kk * 2 + kg) ^ sw) * 8;
            q = *reinterpret_cast<const bf16x8*>(sB + (wn * 64 + j * 32 + r) * GK + cpos);
          }
        };
#ifdef G2_X_NOLDS
#define rdA(q, kk, i) do { for (int e_ = 0; e_ < 8; ++e_) q[e_] = (__bf16)(float)(lane + kk); } while (0)
#define rdB(q, kk, j) do { for (int e_ = 0; e_ < 8; ++e_) q[e_] = (__bf16)(float)(r + kk); } while (0)
#endif
        rdA(fa[0][0], 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) rdB(fb[0][j], 0, j);
#pragma unroll
        for (int i = 1; i < 4; ++i) rdA(fa[0][i], 0, i);
        const bool more = g + 1 < ng || it + 1 < nmy;
#ifndef G2_X_NOMFMA
        __builtin_amdgcn_sched_barrier(0);
        G2_TR(1);
        if (more) issue_all();
        G2_TR(2);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < GK / 16; ++kk) {
          const int cb = kk & 1, nb = cb ^ 1;
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            const int i = n >> 1, j = n & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][i], fb[cb][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < GK / 16) {
              // (in the order the next k-step's MFMAs want them: A0, B0, B1, A1, A2, A3)
              if (n == 0) rdA(fa[nb][0], kk + 1, 0);
              else if (n < 3) rdB(fb[nb][n - 1], kk + 1, n - 1);
              else if (n < 6) rdA(fa[nb][n - 2], kk + 1, n - 2);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#ifndef G2_X_NOMFMA
        if (more) issue_advance();
#endif
        G2_TR(3);
#ifdef G2_X_TRACE
        ++trace_n;
#endif
      }
      slot ^= 1;
    }
#ifdef G2_X_NOEPI
    {  // (every accumulator feeds the test: with four of them, round 3's form, hipcc dropped the MFMAs of the other half)
      float chk = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) chk += acc[i][j][e];
      if (chk == 12345.678f) p.out_bf[0] = 1;
    }
    zero();
    continue;
#endif
    if constexpr (HOUT) {
      float* tile = reinterpret_cast<float*>(g2_lds + 2 * G2_SB) + wave * (G2_RP * 64);
      const int nacc = (p.acc_h != nullptr) + (p.acc2_h != nullptr);
      if (p.relu) {
        if (nacc == 0) g2_hout_epilogue<0, true>(p, tl, acc, tile, lane, wm, wn);
        else if (nacc == 1) g2_hout_epilogue<1, true>(p, tl, acc, tile, lane, wm, wn);
        else g2_hout_epilogue<2, true>(p, tl, acc, tile, lane, wm, wn);
      } else {
        if (nacc == 0) g2_hout_epilogue<0, false>(p, tl, acc, tile, lane, wm, wn);
        else if (nacc == 1) g2_hout_epilogue<1, false>(p, tl, acc, tile, lane, wm, wn);
        else g2_hout_epilogue<2, false>(p, tl, acc, tile, lane, wm, wn);
      }
      zero();
      continue;
    }
    // epilogue: RP rows x 64 columns at a time through this wave's own 2 KB of LDS, 16-byte stores
    float* __restrict__ ob = p.out + (size_t)tl.batch * p.o_bs;
    const float* __restrict__ ab = p.acc ? p.acc + (size_t)tl.batch * p.acc_bs : nullptr;
    const float* __restrict__ ab2 = p.acc2 ? p.acc2 + (size_t)tl.batch * p.acc2_bs : nullptr;
    float* tile = reinterpret_cast<float*>(g2_lds + 2 * G2_SB) + wave * (G2_RP * 64);
    const int c4 = (lane & 15) * 4, lrow = lane >> 4;
    const int n = tl.n0 + wn * 64 + c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int hp = 0; hp < 32 / G2_RP; ++hp) {
#pragma unroll
        for (int e = 0; e < G2_RP / 2; ++e) {
          const int rr = hp * (G2_RP / 2) + e;
          const int row = (rr & 3) + 8 * ((rr >> 2) - hp * (G2_RP / 8)) + 4 * kg;
#pragma unroll
          for (int j = 0; j < 2; ++j) tile[row * 64 + j * 32 + r] = acc[i][j][rr];
        }
#pragma unroll
        for (int q = 0; q < G2_RP / 4; ++q) {
          const int row = lrow + 4 * q;
          const int m = tl.m0 + wm * 128 + i * 32 + hp * G2_RP + row;
          const float4 v = *reinterpret_cast<const float4*>(&tile[row * 64 + c4]);
          float add = 0.0f;
          if (p.bias) add += p.bias[m];
          if (p.bias_bc) add += p.bias_bc[(size_t)tl.batch * p.M + m];
          const size_t o = (size_t)m * p.o_rs + n;
          float vv[4] = {v.x + add, v.y + add, v.z + add, v.w + add};
          if (n + 3 < p.n_valid && ((o & 1) == 0)) {
            if (ab) {
              const f32x4a8 u = *reinterpret_cast<const f32x4a8*>(ab + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) vv[e] += u[e];
            }
            if (ab2) {
              const f32x4a8 u = *reinterpret_cast<const f32x4a8*>(ab2 + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) vv[e] += u[e];
            }
            f32x4a8 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = p.relu ? fmaxf(vv[e], 0.0f) : vv[e];
            *reinterpret_cast<f32x4a8*>(ob + o) = out;
            if (p.out_bf)
              *reinterpret_cast<uint2*>(p.out_bf + ((size_t)tl.batch * p.M + m) * p.out_bf_rs + n) =
                  make_uint2(pack2(out[0], out[1]), pack2(out[2], out[3]));
          } else {
            for (int e = 0; e < 4 && n + e < p.n_valid; ++e) {
              float w1 = vv[e];
              if (ab) w1 += ab[o + e];
              if (ab2) w1 += ab2[o + e];
              w1 = p.relu ? fmaxf(w1, 0.0f) : w1;
              ob[o + e] = w1;
              if (p.out_bf)
                p.out_bf[((size_t)tl.batch * p.M + m) * p.out_bf_rs + n + e] = (unsigned short)(pack2(w1, 0.0f) & 0xffffu);
            }
          }
        }
      }
    }
    zero();
  }
#ifdef G2_X_TRACE
  if ((wave & 3) == 0 && lane == 0) {
    unsigned long long* tot = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + 4096 +
                              (blockIdx.x * 2 + (wave >> 2)) * 4;
    tot[0] = trace_sum[0]; tot[1] = wall_clock64() - trace_w0; tot[2] = trace_sum[2];  // ([1]: 100 MHz ticks)
    tot[3] = __builtin_readcyclecounter() - trace_t0;
  }
#endif
}

// ===================================================================================
// Path 3: the dilated K = 3 convs of the Res2 branches (ecapa_tdnn.py:46: width -> width channels,
// dilation 2 / 3 / 4, padding = dilation), forward and dgrad.  7 of them run back to back per block
// on (B, 64, T) tensors: small, so what matters is one short launch each.  Same in-register
// transpose staging as the fused pointwise kernel, but the [t][ci] LDS tile carries a halo of
// `dil` frames each side and the three taps read it at row offsets 0, dil, 2 dil - the shifted
// operands cost no extra staging.  Zero padding = frames outside [0, T) staged as zeros.
// Workgroup tile: 64 output channels x 128 frames; wave w owns frames 32 w .. 32 w + 31.
constexpr int TAP_MAXD = 4;
constexpr int TAP_ROWS = BN + 2 * TAP_MAXD;  // 136 staged frames
// TIMING-ONLY experiment builds of c1b_tap_kernel (results are garbage; build --variant ... -DTAP_X=<bits>): 1 = no
// activation loads, 2 = no MFMAs, 4 = no output stores, 8 = no weight loads.  profiles/r06_res2_chain.md, section 3.
#ifndef TAP_X
#define TAP_X 0
#endif

struct C1bTap {
  const float* x;
  const unsigned short* a;  // bf16 [3][64 * tiles_m][K]
  float* y;
  const float* bias;
  const float* acc;
  size_t x_bs, y_bs;
  int B, M, K, T, dil, relu, tiles_m, tiles_t, total, per_xcd;
  int Tp;  // c1b_tap_kernel<true>: x and y are bf16 rows of Tp frames (x / y point at unsigned short)
  // c1b_tap_kernel<true>, optional: BatchNorm statistics of the STORED output - per (channel m, 32-frame segment r) the
  // pair {sum, sum of squares} at stats[(m * NR + r) * 2], NR = B * tiles_t * 4, r = (b * tiles_t + tile) * 4 + wave
  float* stats;
  // c1b_tap_kernel<true>, optional (data-gradient launches of the Res2 chain): y is the gradient that joins bn_dy on
  // its way into the BatchNorm whose input was bn_x (conv -> ReLU -> BatchNorm of the PREVIOUS branch); the epilogue
  // also leaves that BatchNorm's five backward sums (h_bn_bwd_partial_kernel's, same arithmetic) per (channel,
  // 32-frame segment) at bsums[(m * NR + r) * 8 + 0..4]
  const unsigned short* bn_x;
  const unsigned short* bn_dy;
  size_t bn_x_bs, bn_dy_bs;
  const float* bn_mean;
  const float* bn_invstd;
  float* bsums;
  // c1b_tap_kernel<true, PRO> (round 6): the operand is COMPUTED while it is staged - the elementwise pass that used to
  // produce it (h_res2_kernel forward, h_bn_bwd_apply_kernel backward: 42 launches per ECAPA step) is gone; its stored
  // results leave this launch as side outputs (the centre 128 frames of the staged tile, 16-byte row pieces).
  //   PRO 1 (forward, branch i >= 1): x = r of branch i - 1; y1 = bf16(x * pa[c] + pb[c]) -> its slice of the concat
  //     (side1), operand t = bf16(y1 + g0) (g0 = this branch's slice of o1) -> side0 (the weight gradient's input)
  //   PRO 2 (data gradient of branch i): x = r of branch i (its BatchNorm's input), g = g0 + g1 (the concat gradient's
  //     slice + the next branch's input gradient; g1 optional), operand dc = bf16(pc (g - k1 - xhat k2)) or 0 where
  //     x <= 0, xhat = (x - pa) pb, pc = gamma pb, k1 = pd invN, k2 = pe invN -> side0 (the weight gradient's dy)
  const unsigned short* g0;
  const unsigned short* g1;
  size_t g0_bs, g1_bs;
  const float *pa, *pb, *pc, *pd, *pe;
  float invN;
  unsigned short* side0;
  unsigned short* side1;
  size_t side0_bs, side1_bs;
};

// fp32 (Cout, Cin, 3) -> bf16 A[tap][m][k].  transpose = 0: m = co, k = ci, tap as is (forward);
// transpose = 1: m = ci, k = co, taps flipped (dgrad: dx[t] = sum_k w[..][2 - k'] dy[t + (k' - 1) d]).
// blockIdx.y = layer of a batch of equally shaped convs (w_stride floats / 3 M K packed elements apart)
__global__ __launch_bounds__(256) void c1b_pack3_kernel(const float* __restrict__ w, unsigned short* __restrict__ a,
                                                        int Cout, int Cin, int transpose, size_t w_stride) {
  const int M = transpose ? Cin : Cout, K = transpose ? Cout : Cin;
  const size_t n = (size_t)3 * M * K;
  w += (size_t)blockIdx.y * w_stride;
  a += (size_t)blockIdx.y * n;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int k = (int)(e % K);
    const int m = (int)((e / K) % M);
    const int tap = (int)(e / ((size_t)K * M));
    const float v = transpose ? w[((size_t)k * Cin + m) * 3 + (2 - tap)] : w[((size_t)m * Cin + k) * 3 + tap];
    f32x2 pr = {v, 0.0f};
    bf16x2 r = __builtin_convertvector(pr, bf16x2);
    a[e] = (unsigned short)(__builtin_bit_cast(unsigned, r) & 0xFFFFu);
  }
}

// HIO: bf16-resident tensors (ecapa_bf16.hip) - x and y are bf16 rows [b][c][Tp]: staging copies the bits, the epilogue
// rounds once and writes zeros for the frames T .. Tp - 1 (tiles cover the whole row).
template <bool HIO, int PRO = 0>
__global__ __launch_bounds__(256) void c1b_tap_kernel(const C1bTap p) {
  static_assert(PRO == 0 || HIO, "the fused prologues exist for the bf16-resident tensors only");
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][3 * 64 * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][TAP_ROWS * LDK];
  const int work = xcd_chunked(blockIdx.x, p.per_xcd);
  if (work >= p.total) return;
  const int mt = work % p.tiles_m;
  const int rest = work / p.tiles_m;
  const int tt = rest % p.tiles_t, b = rest / p.tiles_t;
  const int m0 = mt * 64, t0 = tt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = p.dil;
  const int nrows = BN + 2 * d;  // staged frames: t0 - d .. t0 + 127 + d

  // A: 3 taps x 64 rows x 4 chunks = 768 chunks, 3 per thread
  // B: nrows frames x 4 channel groups of 8: slot s = tid + 256 i -> frame s % TAP_ROWS, group s / TAP_ROWS
  const int XP = HIO ? p.Tp : p.T;  // row pitch of x and y
  const float* __restrict__ xb = HIO ? nullptr : p.x + (size_t)b * p.x_bs;
  const u16* __restrict__ xh = HIO ? reinterpret_cast<const u16*>(p.x) + (size_t)b * p.x_bs : nullptr;
  // per-thread staging roles (3 slots each), fixed for the whole K loop
  const unsigned short* ga[3];
  int la[3], lb_[3];
  const float* gb[3];
  const u16* gh[3];
  bool okb[3], stb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = tid + 256 * i;  // A chunk: tap = c / 256, row = (c % 256) / 4, ch = c % 4
    ga[i] = p.a + ((size_t)(c >> 8) * p.M + m0 + ((c & 255) >> 2)) * p.K + (c & 3) * 8;
    la[i] = ((c >> 8) * 64 + ((c & 255) >> 2)) * LDK + (c & 3) * 8;
    const int fr = c % TAP_ROWS, kg = c / TAP_ROWS;  // B slot: frame, channel group of 8
    const int t = t0 - d + fr;
    stb[i] = kg < 4;
    okb[i] = kg < 4 && fr < nrows && t >= 0 && t < p.T;
    gb[i] = HIO ? nullptr : xb + (size_t)(kg * 8) * p.T + (okb[i] ? t : 0);
    gh[i] = HIO ? xh + (size_t)(kg * 8) * XP + (okb[i] ? t : 0) : nullptr;
    lb_[i] = fr * LDK + kg * 8;
  }
  uint4 ra0, ra1, ra2;
  float rb[3][8];
  unsigned rh[3][8];
#define TAP_FETCH(k0)                                                                        \
  do {                                                                                       \
    if (TAP_X & 8) { ra0 = ra1 = ra2 = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); } else { \
    ra0 = *reinterpret_cast<const uint4*>(ga[0] + (k0));                                     \
    ra1 = *reinterpret_cast<const uint4*>(ga[1] + (k0));                                     \
    ra2 = *reinterpret_cast<const uint4*>(ga[2] + (k0)); }                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                         \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                   \
          if (HIO) rh[i_][j_] = (okb[i_] && !(TAP_X & 1)) ? (unsigned)gh[i_][(size_t)((k0) + j_) * XP] : 0u;   \
          else rb[i_][j_] = okb[i_] ? gb[i_][(size_t)((k0) + j_) * p.T] : 0.0f;              \
        }                                                                                    \
  } while (0)
#define TAP_STASH(buf)                                                                       \
  do {                                                                                       \
    *reinterpret_cast<uint4*>(&sA[buf][la[0]]) = ra0;                                        \
    *reinterpret_cast<uint4*>(&sA[buf][la[1]]) = ra1;                                        \
    *reinterpret_cast<uint4*>(&sA[buf][la[2]]) = ra2;                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) if (stb[i_]) {                          \
      uint4 v_;                                                                              \
      if (HIO) {                                                                             \
        v_.x = rh[i_][0] | (rh[i_][1] << 16);                                                \
        v_.y = rh[i_][2] | (rh[i_][3] << 16);                                                \
        v_.z = rh[i_][4] | (rh[i_][5] << 16);                                                \
        v_.w = rh[i_][6] | (rh[i_][7] << 16);                                                \
      } else {                                                                               \
        v_.x = pack2(rb[i_][0], rb[i_][1]);                                                  \
        v_.y = pack2(rb[i_][2], rb[i_][3]);                                                  \
        v_.z = pack2(rb[i_][4], rb[i_][5]);                                                  \
        v_.w = pack2(rb[i_][6], rb[i_][7]);                                                  \
      }                                                                                      \
      *reinterpret_cast<uint4*>(&sB[buf][lb_[i_]]) = v_;                                     \
    }                                                                                        \
  } while (0)

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  const int r31 = lane & 31, kgl = lane >> 5;
  const int nk = p.K / BK;
  if (PRO) {
    // Round 6: K = 64 (both k-steps = the two buffers) staged in ONE go from 16-byte ROW pieces - 8 frames of one channel
    // - instead of 2-byte loads down the channels: the piece is where the elementwise prologue runs (one coefficient set
    // per piece), its result leaves as a 16-byte side-output store when the piece lies in the tile's centre, and goes
    // to the [frame][k] operand buffers as eight 2-byte LDS writes (the transpose).  Pieces: 18 per channel, frames
    // t0 - 8 .. t0 + 135 (the staged window t0 - d .. t0 + 127 + d, d <= 4, lies inside); lanes run along the channels.
    static_assert(TAP_MAXD <= 8, "the halo fits one piece on either side");
    // weights: both k-steps
    uint4 wa[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 3; ++i) wa[ks][i] = *reinterpret_cast<const uint4*>(ga[i] + ks * BK);
    const u16* __restrict__ q0 = PRO == 3 ? nullptr : p.g0 + (size_t)b * p.g0_bs;
    const u16* __restrict__ q1 = (PRO == 2 && p.g1) ? p.g1 + (size_t)b * p.g1_bs : nullptr;
    u16* __restrict__ o0 = PRO == 3 ? nullptr : p.side0 + (size_t)b * p.side0_bs;
    u16* __restrict__ o1 = PRO == 1 ? p.side1 + (size_t)b * p.side1_bs : nullptr;
    constexpr int NPC = 18, NJ = (64 * NPC + 255) / 256;  // 1152 pieces, 4.5 per thread
    uint4 vx[NJ], v0[NJ], v1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = tid + 256 * j;
      const int c = idx & 63, pc = idx >> 6;
      const int tp = t0 - 8 + pc * 8;  // first frame of the piece
      const bool ok = pc < NPC && tp >= 0 && tp < p.Tp;
      const size_t off = (size_t)c * p.Tp + (ok ? tp : 0);
      // (`off` is a valid address either way: load, then select - a select between pointers would go through scratch)
      vx[j] = *reinterpret_cast<const uint4*>(xh + off);
      if (!ok) vx[j] = make_uint4(0u, 0u, 0u, 0u);
      if (PRO != 3) {
        v0[j] = *reinterpret_cast<const uint4*>(q0 + off);
        if (!ok) v0[j] = make_uint4(0u, 0u, 0u, 0u);
      }
      if (PRO == 2) {
        v1[j] = make_uint4(0u, 0u, 0u, 0u);
        if (q1) {
          v1[j] = *reinterpret_cast<const uint4*>(q1 + off);
          if (!ok) v1[j] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<uint4*>(&sA[ks][la[i]]) = wa[ks][i];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = tid + 256 * j;
      const int c = idx & 63, pc = idx >> 6;
      if (pc >= NPC) continue;
      const int tp = t0 - 8 + pc * 8;
      unsigned wx[4] = {vx[j].x, vx[j].y, vx[j].z, vx[j].w};
      unsigned res[4], res1[4];
      if (PRO == 3) {  // no prologue: the stored operand, zero outside [0, T) like the 2-byte staging
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const unsigned lo = (tp + 2 * h >= 0 && tp + 2 * h < p.T) ? (wx[h] & 0xffffu) : 0u;
          const unsigned hi = (tp + 2 * h + 1 >= 0 && tp + 2 * h + 1 < p.T) ? (wx[h] & 0xffff0000u) : 0u;
          res[h] = lo | hi;
        }
      } else {
        const unsigned w0[4] = {v0[j].x, v0[j].y, v0[j].z, v0[j].w};
        const unsigned w1[4] = {v1[j].x, v1[j].y, v1[j].z, v1[j].w};
        float ca, cb, cc = 0.0f, k1 = 0.0f, k2 = 0.0f;
        ca = p.pa[c];
        cb = p.pb[c];
        if (PRO == 2) {
          cc = p.pc[c] * cb;      // (h_bn_bwd_apply_kernel: sc = gamma[c] * is)
          k1 = p.pd[c] * p.invN;  // dbeta[c] * invN
          k2 = p.pe[c] * p.invN;  // dgamma[c] * invN
        }
        float o[8], y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool live = tp + e >= 0 && tp + e < p.T;
          const float xv = (e & 1) ? __builtin_bit_cast(float, wx[e >> 1] & 0xffff0000u) : __builtin_bit_cast(float, wx[e >> 1] << 16);
          const float g0 = (e & 1) ? __builtin_bit_cast(float, w0[e >> 1] & 0xffff0000u) : __builtin_bit_cast(float, w0[e >> 1] << 16);
          if (PRO == 1) {
            y[e] = live ? fmaf(xv, ca, cb) : 0.0f;  // (h_res2_kernel: y1 = bf16(x * scale + shift), zeros behind T)
            o[e] = g0;
          } else {
            float g = g0;
            if (q1) g += (e & 1) ? __builtin_bit_cast(float, w1[e >> 1] & 0xffff0000u) : __builtin_bit_cast(float, w1[e >> 1] << 16);
            const float xh_ = (xv - ca) * cb;
            float r = cc * ((g + 0.0f) - k1 - xh_ * k2);  // (h_bn_bwd_apply_kernel, rowbias = 0)
            if (!(xv > 0.0f) || !live) r = 0.0f;
            o[e] = r;
          }
        }
        if (PRO == 1) {
#pragma unroll
          for (int h = 0; h < 4; ++h) res1[h] = pack2(y[2 * h], y[2 * h + 1]);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool live = tp + e >= 0 && tp + e < p.T;
            const float yy = (e & 1) ? __builtin_bit_cast(float, res1[e >> 1] & 0xffff0000u) : __builtin_bit_cast(float, res1[e >> 1] << 16);
            o[e] = live ? yy + o[e] : 0.0f;  // t = bf16(y1 + add)
          }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) res[h] = pack2(o[2 * h], o[2 * h + 1]);
        if (pc >= 1 && pc <= 16) {  // the tile's own 128 frames: every (channel, frame) of the tensor is written once
          *reinterpret_cast<uint4*>(o0 + (size_t)c * p.Tp + tp) = make_uint4(res[0], res[1], res[2], res[3]);
          if (PRO == 1) *reinterpret_cast<uint4*>(o1 + (size_t)c * p.Tp + tp) = make_uint4(res1[0], res1[1], res1[2], res1[3]);
        }
      }
      u16* dstb = &sB[c >> 5][0] + (c & 31);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = pc * 8 + e - 8 + d;  // staged frame index: frame t0 - d is row 0
        if (row >= 0 && row < nrows)
          dstb[row * LDK] = (u16)((e & 1) ? (res[e >> 1] >> 16) : (res[e >> 1] & 0xffffu));
      }
    }
    __syncthreads();
  } else {
    TAP_FETCH(0);
    TAP_STASH(0);
    __syncthreads();
  }
  for (int s = 0; s < nk; ++s) {
    const int cur = s & 1;
    if (!PRO && s + 1 < nk) TAP_FETCH((s + 1) * BK);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&sB[cur][(wave * 32 + r31 + tap * d) * LDK + kk * 16 + kgl * 8]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(&sA[cur][(tap * 64 + i * 32 + r31) * LDK + kk * 16 + kgl * 8]);
          // HIO: operands swapped - the accumulators come out transposed (a lane = one channel, see the epilogue)
          if (TAP_X & 2) {
            acc[i][0] += (float)fa[0] + (float)fb[0];  // (keeps the operand reads alive)
          } else
          acc[i] = HIO ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc[i], 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
        }
      }
    }
    if (!PRO && s + 1 < nk) TAP_STASH(cur ^ 1);
    if (!PRO || s + 1 == nk) __syncthreads();  // (PRO: both buffers were complete before the loop)
  }
#undef TAP_FETCH
#undef TAP_STASH

  if (HIO) {
    // Round 4.  With the operands swapped the sums are the same products in the same k order, but transposed: a lane
    // holds channel m0 + 32 i + r31 and 16 frames of it - t0 + 32 wave + (q & 3) + 8 (q >> 2) + 4 kgl, i.e. four groups
    // of 4 consecutive frames.  A group is rounded once and goes to LDS as one 8-byte piece of a [64 channels][128
    // frames] bf16 tile (in the operand buffers the k-loop has left); the tile leaves as 16-byte row pieces, 16 lanes
    // per 256-byte row (round 3: one 2-byte store per value, 32 store instructions per wave - 3.7 of the launch's
    // 13.3 us).  A channel's BatchNorm sums over the wave's 32 frames are 16 in-lane additions and one exchange
    // with the lane that holds the other 16: the statistics pass over the stored tensor (h_bn_partial_kernel) is not
    // needed when p.stats is given.
    constexpr int TPITCH = BN + 8;
    static_assert(64 * TPITCH <= 2 * 3 * 64 * LDK, "the output tile fits the A operand buffers");
    static_assert(64 * TPITCH <= 2 * TAP_ROWS * LDK, "a second tile fits the B operand buffers");
    u16* tile = &sA[0][0];
    const int fl = wave * 32 + 4 * kgl;  // local frame of group 0
    const bool bsum = p.bsums != nullptr;
    if (bsum) {  // the previous branch's BatchNorm input and the other half of its output gradient: row pieces -> LDS
      u16* tileD = &sB[0][0];
      const u16* __restrict__ gx = p.bn_x + (size_t)b * p.bn_x_bs;
      const u16* __restrict__ gd = p.bn_dy + (size_t)b * p.bn_dy_bs;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = tid + 256 * k;
        const int row = c >> 4, col = (c & 15) * 8;
        const size_t go = (size_t)(m0 + row) * p.Tp + t0 + col;
        *reinterpret_cast<uint4*>(&tile[row * TPITCH + col]) = *reinterpret_cast<const uint4*>(gx + go);
        *reinterpret_cast<uint4*>(&tileD[row * TPITCH + col]) = *reinterpret_cast<const uint4*>(gd + go);
      }
      __syncthreads();
    }
    float s1[2], s2[2];
    float q1[2], q2[2], q3[2], q4[2], q5[2];  // bsum: sum g, sum g xhat, sum_{x>0} g, count_{x>0}, sum_{x>0} xhat
    unsigned wq[2][4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ml = i * 32 + r31;
      const float bv = p.bias ? p.bias[m0 + ml] : 0.0f;
      const float mu = bsum ? p.bn_mean[m0 + ml] : 0.0f, is = bsum ? p.bn_invstd[m0 + ml] : 0.0f;
      s1[i] = 0.0f;
      s2[i] = 0.0f;
      q1[i] = q2[i] = q3[i] = q4[i] = q5[i] = 0.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][4 * g + e] + bv;
          if (p.relu) v[e] = fmaxf(v[e], 0.0f);
          if (t0 + fl + 8 * g + e >= p.T) v[e] = 0.0f;
        }
        const unsigned w0 = pack2(v[0], v[1]), w1 = pack2(v[2], v[3]);
        wq[i][g][0] = w0;
        wq[i][g][1] = w1;
        const float r0 = __builtin_bit_cast(float, w0 << 16), r1 = __builtin_bit_cast(float, w0 & 0xffff0000u);
        const float r2 = __builtin_bit_cast(float, w1 << 16), r3 = __builtin_bit_cast(float, w1 & 0xffff0000u);
        if (p.stats != nullptr) {  // of the values as stored (zeros behind T add nothing)
          s1[i] += (r0 + r1) + (r2 + r3);
          s2[i] = fmaf(r0, r0, s2[i]);
          s2[i] = fmaf(r1, r1, s2[i]);
          s2[i] = fmaf(r2, r2, s2[i]);
          s2[i] = fmaf(r3, r3, s2[i]);
        }
        if (bsum) {
          const uint2 ux = *reinterpret_cast<const uint2*>(&tile[ml * TPITCH + fl + 8 * g]);
          const uint2 ud = *reinterpret_cast<const uint2*>(&(&sB[0][0])[ml * TPITCH + fl + 8 * g]);
          const float xv[4] = {__builtin_bit_cast(float, ux.x << 16), __builtin_bit_cast(float, ux.x & 0xffff0000u),
                               __builtin_bit_cast(float, ux.y << 16), __builtin_bit_cast(float, ux.y & 0xffff0000u)};
          const float dv[4] = {__builtin_bit_cast(float, ud.x << 16), __builtin_bit_cast(float, ud.x & 0xffff0000u),
                               __builtin_bit_cast(float, ud.y << 16), __builtin_bit_cast(float, ud.y & 0xffff0000u)};
          const float rr[4] = {r0, r1, r2, r3};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t0 + fl + 8 * g + e < p.T) {
              const float gg = dv[e] + rr[e];  // (h_bn_bwd_partial_kernel: g = dy + dy2)
              const float xh = (xv[e] - mu) * is;
              q1[i] += gg;
              q2[i] = fmaf(gg, xh, q2[i]);
              if (xv[e] > 0.0f) {
                q3[i] += gg;
                q4[i] += 1.0f;
                q5[i] += xh;
              }
            }
        }
      }
    }
    if (bsum) {
      const int NR = p.B * p.tiles_t * 4, ridx = (b * p.tiles_t + tt) * 4 + wave;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float a1 = q1[i] + __shfl_xor(q1[i], 32, 64), a2 = q2[i] + __shfl_xor(q2[i], 32, 64);
        const float a3 = q3[i] + __shfl_xor(q3[i], 32, 64), a4 = q4[i] + __shfl_xor(q4[i], 32, 64);
        const float a5 = q5[i] + __shfl_xor(q5[i], 32, 64);
        if (kgl == 0) {
          float* o = p.bsums + ((size_t)(m0 + i * 32 + r31) * NR + ridx) * 8;
          *reinterpret_cast<float4*>(o) = make_float4(a1, a2, a3, a4);
          *reinterpret_cast<float4*>(o + 4) = make_float4(a5, 0.0f, 0.0f, 0.0f);
        }
      }
      __syncthreads();  // everyone has read the staged tiles: the output tile takes the first one's place
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<uint2*>(&tile[(i * 32 + r31) * TPITCH + fl + 8 * g]) = make_uint2(wq[i][g][0], wq[i][g][1]);
    if (p.stats != nullptr) {
      const int NR = p.B * p.tiles_t * 4, ridx = (b * p.tiles_t + tt) * 4 + wave;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float a1 = s1[i] + __shfl_xor(s1[i], 32, 64), a2 = s2[i] + __shfl_xor(s2[i], 32, 64);
        if (kgl == 0)
          *reinterpret_cast<float2*>(p.stats + ((size_t)(m0 + i * 32 + r31) * NR + ridx) * 2) = make_float2(a1, a2);
      }
    }
    __syncthreads();
    u16* __restrict__ yh = reinterpret_cast<u16*>(p.y) + (size_t)b * p.y_bs;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = tid + 256 * k;
      const int row = c >> 4, col = (c & 15) * 8;
      if (t0 + col < p.Tp && (!(TAP_X & 4) || tile[row * TPITCH + col] == 0x1234))
        *reinterpret_cast<uint4*>(yh + (size_t)(m0 + row) * p.Tp + t0 + col) =
            *reinterpret_cast<const uint4*>(&tile[row * TPITCH + col]);
    }
    return;
  }
  const int t = t0 + wave * 32 + r31;
  if (t >= p.T) return;
  float* __restrict__ yb = p.y + (size_t)b * p.y_bs;
  const float* __restrict__ ab = p.acc ? p.acc + (size_t)b * p.y_bs : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kgl;
      float v = acc[i][q] + (p.bias ? p.bias[m] : 0.0f);
      if (ab) v += ab[(size_t)m * p.T + t];
      if (p.relu) v = fmaxf(v, 0.0f);
      yb[(size_t)m * p.T + t] = v;
    }
}

bool tap_ok(const AirConv1d* p) {
  return p && p->B > 0 && p->T > 0 && p->K == 3 && p->dil >= 1 && p->dil <= TAP_MAXD && p->pad == p->dil;
}

int run_tap(const float* x, size_t x_bs, const float* w, int transpose, float* y, size_t y_bs, const float* bias,
            const float* acc, int relu, int B, int Cout, int Cin, int T, int dil, void* ws, hipStream_t st,
            const unsigned short* w_packed = nullptr) {
  const int M = transpose ? Cin : Cout, K = transpose ? Cout : Cin;
  const unsigned short* a = w_packed;
  if (a == nullptr) {
    unsigned short* pa = reinterpret_cast<unsigned short*>(ws);
    const size_t n = (size_t)3 * M * K;
    hipLaunchKernelGGL(c1b_pack3_kernel, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, st, w, pa, Cout, Cin,
                       transpose, (size_t)0);
    AIR_CHECK_LAUNCH();
    a = pa;
  }
  C1bTap p;
  p.x = x; p.a = a; p.y = y; p.bias = bias; p.acc = acc; p.x_bs = x_bs; p.y_bs = y_bs;
  p.B = B; p.M = M; p.K = K; p.T = T; p.dil = dil; p.relu = relu;
  p.tiles_m = M / 64;
  p.tiles_t = (T + BN - 1) / BN;
  p.total = B * p.tiles_t * p.tiles_m;
  p.per_xcd = (p.total + NXCD - 1) / NXCD;
  AirProfScope prof(AIR_K_C1B_TAP, 2.0 * B * T * (double)Cout * Cin * 3, st);
  p.Tp = 0;
  p.stats = nullptr;
  p.bsums = nullptr; p.bn_x = nullptr; p.bn_dy = nullptr; p.bn_mean = nullptr; p.bn_invstd = nullptr;
  p.bn_x_bs = p.bn_dy_bs = 0;
  hipLaunchKernelGGL(c1b_tap_kernel<false>, dim3(p.per_xcd * NXCD), dim3(256), 0, st, p);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

bool shape_ok(const AirConv1d* p) {
  return p && p->B > 0 && p->T > 0 && p->Cin > 0 && p->Cout > 0 && p->K == 1 && p->pad == 0;
}
size_t xbs(const AirConv1d* p) { return p->x_bstride ? p->x_bstride : (size_t)p->Cin * p->T; }
size_t ybs(const AirConv1d* p) { return p->y_bstride ? p->y_bstride : (size_t)p->Cout * p->T; }

bool gemm_ps_ok(int M);
int launch_cvt_s(const float* x, size_t x_bs, int B, int C, int T, unsigned short* out, hipStream_t st);
int wgrad_nsplit(const AirConv1d* p, int* b_per_split) {
  const bool ps = gemm_ps_ok(p->Cout) && p->Cin % 256 == 0;
  const int tiles = ps ? (p->Cout / 256) * (p->Cin / 256) : (p->Cout / BM) * (p->Cin / BN);
  int want = (ps ? 256 : 512) / tiles;  // resident workgroups: one (256 x 256 kernel) or two per CU
  if (want < 1) want = 1;
  if (want > p->B) want = p->B;
  const int per = (p->B + want - 1) / want;
  *b_per_split = per;
  return (p->B + per - 1) / per;
}

int run_fwd(const float* x, size_t x_bs, const float* w, int transpose, float* y, size_t y_bs, const float* bias,
            const float* bias_bc, const float* acc, int relu, int B, int M, int K, int T, void* ws, double flops,
            hipStream_t st, size_t acc_bs = 0, const float* acc2 = nullptr, size_t acc2_bs = 0) {
  unsigned short* a = reinterpret_cast<unsigned short*>(ws);
  C1bFwd p;
  p.x = x; p.a = a; p.y = y; p.bias = bias; p.bias_bc = bias_bc; p.acc = acc; p.acc2 = acc2;
  p.x_bs = x_bs; p.y_bs = y_bs;
  p.acc_bs = acc_bs ? acc_bs : y_bs;
  p.acc2_bs = acc2_bs ? acc2_bs : y_bs;
  p.B = B; p.M = M; p.K = K; p.T = T; p.relu = relu;
  p.tiles_m = M / BM;
  p.tiles_t = (T + BN - 1) / BN;
  p.total = B * p.tiles_t * p.tiles_m;
  p.per_xcd = (p.total + NXCD - 1) / NXCD;
  const int use_ps = air_opt(AIR_OPT_C1B_PS);
  // the DMA moves 16-byte chunks: frame rows have to start on 8-byte boundaries (even T and strides),
  // and every byte offset has to fit the descriptor's 32 bits
  const bool acc_al = ((reinterpret_cast<size_t>(acc) | reinterpret_cast<size_t>(acc2)) & 7) == 0 && p.acc_bs % 2 == 0 &&
                      p.acc2_bs % 2 == 0 && (reinterpret_cast<size_t>(y) & 7) == 0 && y_bs % 2 == 0;
  const bool ps_ok = use_ps && acc_al && T % 2 == 0 && x_bs % 2 == 0 && (reinterpret_cast<size_t>(x) & 7) == 0 &&
                     ((size_t)(B - 1) * x_bs + (size_t)K * T) * 4 < ((size_t)1 << 32) &&
                     (size_t)M * K * 2 < ((size_t)1 << 32);
  if (ps_ok) {
    const size_t n8 = (size_t)M * K / 8;
    hipLaunchKernelGGL(c1b_pack_stage_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, w,
                       reinterpret_cast<uint4*>(a), M, K, transpose);
    AIR_CHECK_LAUNCH();
    AirProfScope prof(AIR_K_C1B_FWD, flops, st);
    if (M % 256 == 0 && (use_ps & 4)) return fd_launch2(p, st);
    return (M % 256 == 0 && (use_ps & 2)) ? fd_launch<4>(p, st) : fd_launch<2>(p, st);
  }
  const size_t n2 = (size_t)M * K / 2;
  hipLaunchKernelGGL(c1b_pack_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, w, a, M, K, transpose);
  AIR_CHECK_LAUNCH();
  AirProfScope prof(AIR_K_C1B_FWD, flops, st);
  hipLaunchKernelGGL(c1b_fwd_kernel, dim3(p.per_xcd * NXCD), dim3(256), 0, st, p);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}


bool gemm_ps_ok(int M) {
  const int use = air_opt(AIR_OPT_C1B_GEMM_PS);
  return use && M % G2_BM == 0;
}

// *bf_done (optional): set when the kernel that ran wrote g.out_bf itself
int launch_gemm(NtGemm& g, int nbatch, int n_padded, size_t a_bytes, size_t b_bytes, int kid, double flops,
                hipStream_t st, bool* bf_done = nullptr, bool b_kmajor = false, double alg_bytes = -1.0) {
  if (bf_done) *bf_done = false;
  // algorithmic HBM bytes for the profiler (operands once + result once); callers with strided operands state them
  if (alg_bytes < 0) alg_bytes = (double)a_bytes + (double)b_bytes + 4.0 * g.M * (double)n_padded * nbatch;
  if (b_kmajor && !(gemm_ps_ok(g.M) && n_padded % G2_BN == 0 && a_bytes < ((size_t)1 << 32) && b_bytes < ((size_t)1 << 32) &&
                    ((size_t)g.out & 7) == 0 && g.o_bs % 2 == 0 && g.b_rs % 8 == 0))
    return AIR_EUNSUPPORTED;  // only the 256 x 256 kernel reads a K-major B
  if (gemm_ps_ok(g.M) && n_padded % G2_BN == 0 && a_bytes < ((size_t)1 << 32) && b_bytes < ((size_t)1 << 32) &&
      ((size_t)g.out & 7) == 0 && g.o_bs % 2 == 0) {
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(c1b_gemm_ps_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(c1b_gemm_ps_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) == hipSuccess;
    if (!attr_ok) return AIR_ELAUNCH;
    g.tiles_m = g.M / G2_BM;
    g.tiles_n = n_padded / G2_BN;
    g.total = g.tiles_m * g.tiles_n * nbatch;
    g.per_xcd = (g.total + NXCD - 1) / NXCD;
    AirProfScope prof(kid, flops, st, -1.0, alg_bytes);
    if (b_kmajor)
      hipLaunchKernelGGL(c1b_gemm_ps_kernel<true>, dim3(256), dim3(512), G2_LDS, st, g, (unsigned)a_bytes, (unsigned)b_bytes);
    else
      hipLaunchKernelGGL(c1b_gemm_ps_kernel<false>, dim3(256), dim3(512), G2_LDS, st, g, (unsigned)a_bytes, (unsigned)b_bytes);
    AIR_CHECK_LAUNCH();
    if (bf_done) *bf_done = g.out_bf != nullptr;
    return AIR_OK;
  }
  g.tiles_m = g.M / BM;
  g.tiles_n = n_padded / BN;
  g.total = g.tiles_m * g.tiles_n * nbatch;
  g.per_xcd = (g.total + NXCD - 1) / NXCD;
  AirProfScope prof(kid, flops, st, -1.0, alg_bytes);
  hipLaunchKernelGGL(c1b_gemm_kernel, dim3(g.per_xcd * NXCD), dim3(256), 0, st, g);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

// the wide layers take the bf16-resident path for forward/dgrad as well (measured: layer4 faster,
// the 512-channel layers not: their fused kernel is already HBM-bound on the fp32 tensors)
bool wide(int M, int K) { return (size_t)M * K >= (size_t)1024 * 1024 && K % GK == 0; }

size_t gemm_fwd_ws(int B, int M, int K, int T) {
  return align256((size_t)M * K * 2) + align256((size_t)B * round_up(T, TP_ALIGN) * K * 2);
}

// forward / dgrad through the GEMM: y_b[m][t] = sum_k A[m][k] x_b[k][t]
int run_fwd_gemm(const float* x, size_t x_bs, const float* w, int transpose, float* y, size_t y_bs, const float* bias,
                 const float* bias_bc, const float* acc, int relu, int B, int M, int K, int T, void* ws, double flops,
                 hipStream_t st, unsigned short* y_bf = nullptr, bool* bf_done = nullptr) {
  unsigned short* a = reinterpret_cast<unsigned short*>(ws);
  unsigned short* xt = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(ws) + align256((size_t)M * K * 2));
  const int Tp = round_up(T, TP_ALIGN);
  const size_t n2 = (size_t)M * K / 2;
  hipLaunchKernelGGL(c1b_pack_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, w, a, M, K, transpose);
  AIR_CHECK_LAUNCH();
  const int vec_t = T % 2 == 0 && x_bs % 2 == 0 && (reinterpret_cast<size_t>(x) & 7) == 0;
  hipLaunchKernelGGL(c1b_cvt_t_kernel, dim3(Tp / 64, K / 64, B), dim3(256), 0, st, x, x_bs, K, T, Tp, vec_t, xt);
  AIR_CHECK_LAUNCH();
  NtGemm g;
  g.hstats = nullptr;
  g.a = a; g.b = xt; g.out = y; g.bias = bias; g.bias_bc = bias_bc; g.acc = acc;
  g.acc2 = nullptr; g.acc_bs = y_bs; g.acc2_bs = 0;
  g.out_bf = y_bf; g.out_bf_rs = Tp;
  g.a_rs = K; g.a_ss = 0; g.a_bs = 0;
  g.b_rs = K; g.b_ss = 0; g.b_bs = (size_t)Tp * K;
  g.o_rs = T; g.o_bs = y_bs;
  g.M = M; g.n_valid = T; g.kseg = K; g.nseg_per_batch = 1; g.nseg_total = B; g.relu = relu;
  return launch_gemm(g, B, Tp, (size_t)M * K * 2, (size_t)B * Tp * K * 2, AIR_K_C1B_GEMM, flops, st, bf_done);
}


// ===================================================================================
// Path 4: weight gradient of the dilated K = 3 Res2 convs on bf16-resident rows,
//   dw[co][ci][k] = sum_{b,t} dy[b][co][t] x[b][ci][t + (k - 1) dil]        (ecapa_tdnn.py:46, padding = dil)
// for ALL branches of a block in one launch.  A branch is two (B, 64 j, Tp) operands = 25 MB for 2.4 GFLOP: the
// kernel is bound by reading them once.  Work item = (branch, 64 x 64 (co, ci) tile, part): the part's utterances
// are walked in 64-frame stages; dy[64 co][64 t] and x[64 ci][8 + 64 + 8 t] (one 16-byte chunk of halo each side,
// zeros outside [0, Tp)) go through registers into padded LDS rows (144 / 176 bytes: 8 consecutive rows cover all
// banks for 16-byte reads).  t is the contraction index of v_mfma_f32_32x32x16_bf16, so a shifted tap is a shift
// ALONG k: a lane reads the three chunks around its 8 frames and funnel-shifts the pair (v_alignbit for odd
// dilations, plain register selection for even ones) - the two outer taps cost no extra staging.  Products of
// bf16 values are exact in fp32 and the sums are fp32: the same arithmetic as the fp32 contraction of the widened
// operands this replaces (conv_wgrad_kernel<1,3,1> on air_h_to_f32 copies), without the copies.
// Wave w owns the 32 x 32 block (w & 1, w >> 1) of the tile for all three taps; per-part partial sums are
// reduced in fixed order by c1b_tapw_reduce_kernel.
constexpr int TW_MAXB = 16;
constexpr int TW_KC = 64;                 // frames per stage
constexpr int TW_AS = TW_KC + 8;          // A row pitch (elements): 144 bytes
constexpr int TW_BS = TW_KC + 16 + 8;     // B row pitch: 8 + 64 + 8 frames + 8 pad = 176 bytes

struct TapWgrad {
  const u16* x[TW_MAXB];
  const u16* dy[TW_MAXB];
  size_t x_bs[TW_MAXB], dy_bs[TW_MAXB];
  float* dw[TW_MAXB];
  float* partial;  // [branch][tile][part][3][64][64]
  int nb, B, W, Tp, nparts, tiles_c;
};

template <int SUB>
__device__ __forceinline__ bf16x8 tw_shift(const uint4 lo, const uint4 hi) {
  // elements SUB .. SUB + 7 of the 16 bf16 values lo ++ hi
  const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  constexpr int q = SUB / 2;
  unsigned o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o[j] = (SUB & 1) ? __builtin_amdgcn_alignbit(w[q + j + 1], w[q + j], 16) : w[q + j];
  const uint4 v = make_uint4(o[0], o[1], o[2], o[3]);
  return __builtin_bit_cast(bf16x8, v);
}

template <int DIL>
__global__ __launch_bounds__(256) void c1b_tapw_kernel(const TapWgrad a) {
  __shared__ __attribute__((aligned(16))) u16 sA[2][64 * TW_AS];
  __shared__ __attribute__((aligned(16))) u16 sB[2][64 * TW_BS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = blockIdx.x % a.nparts;
  const int rest = blockIdx.x / a.nparts;
  const int tiles = a.tiles_c * a.tiles_c;
  const int tile = rest % tiles, br = rest / tiles;
  const int co0 = (tile / a.tiles_c) * 64, ci0 = (tile % a.tiles_c) * 64;
  const int Tp = a.Tp, spu = Tp / TW_KC;  // stages per utterance
  const int nutt = (a.B - part + a.nparts - 1) / a.nparts;
  const int nst = nutt * spu;
  const u16* __restrict__ xg = a.x[br];
  const u16* __restrict__ yg = a.dy[br];
  const size_t xbs = a.x_bs[br], ybs = a.dy_bs[br];

  // staging roles: A slots tid, tid + 256 of [64 rows][8 chunks]; B slots tid, tid + 256, tid + 512 of [64][10]
  uint4 ra0, ra1, rb0, rb1, rb2;
  auto lda = [&](int s, int e) __attribute__((always_inline)) -> uint4 {
    const int u = s / spu, k0 = (s - u * spu) * TW_KC;
    const int b = part + u * a.nparts;
    const int slot = tid + 256 * e, row = slot >> 3, ch = slot & 7;
    return *reinterpret_cast<const uint4*>(yg + (size_t)b * ybs + (size_t)(co0 + row) * Tp + k0 + ch * 8);
  };
  auto ldb = [&](int s, int e) __attribute__((always_inline)) -> uint4 {
    const int u = s / spu, k0 = (s - u * spu) * TW_KC;
    const int b = part + u * a.nparts;
    const int slot = tid + 256 * e, row = slot / 10, ch = slot - row * 10;
    const int f = k0 - 8 + ch * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (slot < 640 && f >= 0 && f < Tp)
      v = *reinterpret_cast<const uint4*>(xg + (size_t)b * xbs + (size_t)(ci0 + row) * Tp + f);
    return v;
  };
  auto sta = [&](int buf, int e, const uint4 v) __attribute__((always_inline)) {
    const int slot = tid + 256 * e, row = slot >> 3, ch = slot & 7;
    *reinterpret_cast<uint4*>(&sA[buf][row * TW_AS + ch * 8]) = v;
  };
  auto stb = [&](int buf, int e, const uint4 v) __attribute__((always_inline)) {
    const int slot = tid + 256 * e, row = slot / 10, ch = slot - row * 10;
    if (slot < 640) *reinterpret_cast<uint4*>(&sB[buf][row * TW_BS + ch * 8]) = v;
  };
#define TW_LOAD(s) do { ra0 = lda(s, 0); ra1 = lda(s, 1); rb0 = ldb(s, 0); rb1 = ldb(s, 1); rb2 = ldb(s, 2); } while (0)
#define TW_STORE(buf) do { sta(buf, 0, ra0); sta(buf, 1, ra1); stb(buf, 0, rb0); stb(buf, 1, rb1); stb(buf, 2, rb2); } while (0)

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
  const int wm = wave & 1, wn = wave >> 1;
  const int r = lane & 31, kg = lane >> 5;

  if (nst > 0) {
    TW_LOAD(0);
    TW_STORE(0);
    if (nst > 1) TW_LOAD(1);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
      const int buf = s & 1;
      const u16* pa = &sA[buf][(wm * 32 + r) * TW_AS + kg * 8];
      const u16* pb = &sB[buf][(wn * 32 + r) * TW_BS + kg * 8];
#pragma unroll
      for (int kk = 0; kk < TW_KC / 16; ++kk) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(pa + kk * 16);
        const uint4 L = *reinterpret_cast<const uint4*>(pb + kk * 16);
        const uint4 C = *reinterpret_cast<const uint4*>(pb + kk * 16 + 8);
        const uint4 R = *reinterpret_cast<const uint4*>(pb + kk * 16 + 16);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, tw_shift<8 - DIL>(L, C), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, C), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, tw_shift<DIL>(C, R), acc[2], 0, 0, 0);
      }
      if (s + 1 < nst) {
        TW_STORE(buf ^ 1);  // its previous readers passed the barrier that ended stage s - 1
        if (s + 2 < nst) TW_LOAD(s + 2);
      }
      __syncthreads();
    }
  }

  float* __restrict__ po = a.partial + ((size_t)(br * tiles + tile) * a.nparts + part) * (3 * 64 * 64);
  const int col = lane & 31, half = lane >> 5;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
      po[(t * 64 + m) * 64 + wn * 32 + col] = acc[t][q];
    }
#undef TW_LOAD
#undef TW_STORE
}

// dw[co0 + co][ci0 + ci][tap] = sum over parts.  A workgroup owns 64 consecutive (tap, co, ci) outputs of one
// (branch, tile); its four waves take the parts p = w, w + 4, ... (every load instruction one coalesced 256-byte
// segment, 8 in flight per wave) and their sums meet in LDS in wave order: deterministic, whatever the part count.
__global__ __launch_bounds__(256) void c1b_tapw_reduce_kernel(const TapWgrad a) {
  __shared__ float s_part[4][64];
  constexpr int PER = 3 * 64 * 64;
  const int tiles = a.tiles_c * a.tiles_c;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int blocks_per = PER / 64;
  const int bt = blockIdx.x / blocks_per;  // (branch, tile)
  const int e = (blockIdx.x - bt * blocks_per) * 64 + lane;
  const int br = bt / tiles, tile = bt - br * tiles;
  const float* __restrict__ src = a.partial + (size_t)bt * a.nparts * PER + e;
  float acc = 0.0f;
  int p = w;
  for (; p + 28 < a.nparts; p += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + 4 * u) * PER];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; p < a.nparts; p += 4) acc += src[(size_t)p * PER];
  s_part[w][lane] = acc;
  __syncthreads();
  if (w == 0) {
    const float r = ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
    const int tap = e / 4096, co = (e >> 6) & 63, ci = e & 63;
    const int co_g = (tile / a.tiles_c) * 64 + co, ci_g = (tile % a.tiles_c) * 64 + ci;
    a.dw[br][((size_t)co_g * a.W + ci_g) * 3 + tap] = r;
  }
}

}  // namespace

extern "C" {

int air_conv1d_bf16_supported(const AirConv1d* p, int pass) {
  if (tap_ok(p)) {  // dilated K = 3: forward (M = Cout, K = Cin) and dgrad (M = Cin, K = Cout) only
    if (pass == 0) return p->Cout % 64 == 0 && p->Cin % BK == 0;
    if (pass == 1) return p->Cin % 64 == 0 && p->Cout % BK == 0;
    return 0;
  }
  if (!shape_ok(p)) return 0;
  switch (pass) {
    case 0: return p->Cout % BM == 0 && p->Cin % BK == 0;  // forward: M = Cout, K = Cin
    case 1: return p->Cin % BM == 0 && p->Cout % BK == 0;  // dgrad:   M = Cin,  K = Cout
    case 2: return p->Cout % BM == 0 && p->Cin % BN == 0;  // wgrad:   M = Cout, N = Cin
    default: return 0;
  }
}

size_t air_conv1d_bf16_ws_bytes(const AirConv1d* p) {
  if (tap_ok(p)) return (size_t)3 * p->Cout * p->Cin * sizeof(unsigned short) + 256;
  if (!shape_ok(p)) return 0;
  size_t n = (size_t)p->Cout * p->Cin * sizeof(unsigned short);  // packed bf16 weights
  if (wide(p->Cout, p->Cin) || wide(p->Cin, p->Cout)) {
    // forward transposes X (B x Tp x Cin), dgrad transposes dY (B x Tp x Cout): size for the larger one
    const size_t g = gemm_fwd_ws(p->B, p->Cout, p->Cin > p->Cout ? p->Cin : p->Cout, p->T);
    if (g > n) n = g;
  }
  if (air_conv1d_bf16_supported(p, 2)) {
    int per;
    const int Tp = round_up(p->T, TP_ALIGN);
    const size_t part = align256((size_t)wgrad_nsplit(p, &per) * p->Cout * p->Cin * sizeof(float)) +
                        align256((size_t)p->B * Tp * p->Cout * 2) + align256((size_t)p->B * Tp * p->Cin * 2);
    if (part > n) n = part;
  }
  return n + 256;
}

int air_conv1d_pointwise_bf16_kmajor(const AirConv1d* p, const unsigned short* xb, size_t xb_bstride, const float* w,
                                     int dgrad, const float* bias, const float* bias_bc, int relu, const float* accumulate,
                                     size_t acc_bstride, const float* accumulate2, size_t acc2_bstride, float* y,
                                     unsigned short* y_bf16, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!shape_ok(p) || !xb || !w || !y) return AIR_EINVAL;
  if (!air_conv1d_bf16_supported(p, dgrad ? 1 : 0)) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_bf16_ws_bytes(p)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  const int M = dgrad ? p->Cin : p->Cout, K = dgrad ? p->Cout : p->Cin, B = p->B, T = p->T;
  const int Tp = round_up(T, TP_ALIGN);
  if (K % GK != 0) return AIR_EUNSUPPORTED;
  unsigned short* a = reinterpret_cast<unsigned short*>(ws);
  const size_t n2 = (size_t)M * K / 2;
  hipLaunchKernelGGL(c1b_pack_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, w, a, M, K, dgrad ? 1 : 0);
  AIR_CHECK_LAUNCH();
  NtGemm g;
  g.hstats = nullptr;
  g.a = a; g.b = xb; g.out = y; g.bias = bias; g.bias_bc = bias_bc; g.acc = accumulate; g.acc2 = accumulate2;
  g.out_bf = y_bf16; g.out_bf_rs = Tp;
  g.a_rs = K; g.a_ss = 0; g.a_bs = 0;
  g.b_rs = Tp; g.b_ss = 0; g.b_bs = xb_bstride ? xb_bstride : (size_t)K * Tp;
  g.o_rs = T; g.o_bs = dgrad ? xbs(p) : ybs(p);
  g.acc_bs = acc_bstride ? acc_bstride : g.o_bs;
  g.acc2_bs = acc2_bstride ? acc2_bstride : g.o_bs;
  if (((reinterpret_cast<size_t>(accumulate) | reinterpret_cast<size_t>(accumulate2)) & 7) || g.acc_bs % 2 || g.acc2_bs % 2)
    return AIR_EUNSUPPORTED;
  g.M = M; g.n_valid = T; g.kseg = K; g.nseg_per_batch = 1; g.nseg_total = B; g.relu = relu;
  return launch_gemm(g, B, Tp, (size_t)M * K * 2, ((size_t)(B - 1) * g.b_bs + (size_t)K * Tp) * 2, AIR_K_C1B_GEMM,
                     2.0 * B * T * (double)p->Cout * p->Cin, st, nullptr, true);
}

/* Dilated K = 3 conv of a Res2 branch on bf16-resident rows: forward (dgrad = 0: y = relu?(W * x + bias)) or data
 * gradient (dgrad = 1, w_packed from air_conv1d_tap_pack_bf16(transpose = 1): dx = W^T * dy).  w_packed required. */
size_t air_h_conv1d_tap_stats_bytes(int B, int Cout, int Tp) {
  if (B <= 0 || Cout <= 0 || Tp <= 0 || Tp % BN != 0) return 0;
  return (size_t)Cout * B * (Tp / BN) * 4 * 2 * sizeof(float);
}

int air_h_conv1d_tap(int B, int Cin, int Cout, int T, int Tp, int dil, const unsigned short* x, size_t x_bs,
                     const unsigned short* w_packed, int dgrad, const float* bias, int relu, unsigned short* y, size_t y_bs,
                     air_stream_t stream) {
  return air_h_conv1d_tap_ex(B, Cin, Cout, T, Tp, dil, x, x_bs, w_packed, dgrad, bias, relu, y, y_bs, nullptr, stream);
}

int air_h_conv1d_tap_ex(int B, int Cin, int Cout, int T, int Tp, int dil, const unsigned short* x, size_t x_bs,
                        const unsigned short* w_packed, int dgrad, const float* bias, int relu, unsigned short* y,
                        size_t y_bs, void* stats, air_stream_t stream) {
  return air_h_conv1d_tap_ex2(B, Cin, Cout, T, Tp, dil, x, x_bs, w_packed, dgrad, bias, relu, y, y_bs, stats, nullptr, 0,
                              nullptr, 0, nullptr, nullptr, nullptr, stream);
}

size_t air_h_conv1d_tap_bwd_sums_bytes(int B, int C, int Tp) {
  if (B <= 0 || C <= 0 || Tp <= 0 || Tp % BN != 0) return 0;
  return (size_t)C * B * (Tp / BN) * 4 * 8 * sizeof(float);
}

int air_h_conv1d_tap_ex2(int B, int Cin, int Cout, int T, int Tp, int dil, const unsigned short* x, size_t x_bs,
                         const unsigned short* w_packed, int dgrad, const float* bias, int relu, unsigned short* y,
                         size_t y_bs, void* stats, const unsigned short* bn_x, size_t bn_x_bs,
                         const unsigned short* bn_dy, size_t bn_dy_bs, const float* bn_mean, const float* bn_invstd,
                         void* bn_sums, air_stream_t stream) {
  if (!x || !w_packed || !y || B <= 0 || T <= 0 || Tp < T || Tp % BN != 0) return AIR_EINVAL;
  if (stats != nullptr && (reinterpret_cast<size_t>(stats) & 7)) return AIR_EINVAL;
  if (bn_sums != nullptr && (!bn_x || !bn_dy || !bn_mean || !bn_invstd || (reinterpret_cast<size_t>(bn_sums) & 15) ||
                             ((reinterpret_cast<size_t>(bn_x) | reinterpret_cast<size_t>(bn_dy)) & 15) ||
                             bn_x_bs % 8 || bn_dy_bs % 8))
    return AIR_EINVAL;
  const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  if (dil < 1 || dil > TAP_MAXD || M % 64 != 0 || K % BK != 0) return AIR_EUNSUPPORTED;
  C1bTap p = {};
  p.x = reinterpret_cast<const float*>(x); p.a = w_packed; p.y = reinterpret_cast<float*>(y); p.bias = bias; p.acc = nullptr;
  p.x_bs = x_bs ? x_bs : (size_t)K * Tp; p.y_bs = y_bs ? y_bs : (size_t)M * Tp;
  p.B = B; p.M = M; p.K = K; p.T = T; p.dil = dil; p.relu = relu; p.Tp = Tp;
  p.stats = reinterpret_cast<float*>(stats);
  p.bsums = reinterpret_cast<float*>(bn_sums);
  p.bn_x = bn_x; p.bn_dy = bn_dy; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd;
  p.bn_x_bs = bn_x_bs ? bn_x_bs : (size_t)M * Tp; p.bn_dy_bs = bn_dy_bs ? bn_dy_bs : (size_t)M * Tp;
  p.tiles_m = M / 64;
  p.tiles_t = Tp / BN;
  p.total = B * p.tiles_t * p.tiles_m;
  p.per_xcd = (p.total + NXCD - 1) / NXCD;
  hipStream_t st = air_stream(stream);
  AirProfScope prof(AIR_K_C1B_TAP, 2.0 * B * T * (double)Cout * Cin * 3, st);
  if (M == 64 && K == 64 && air_opt(AIR_OPT_TAP_ROWS) && ((reinterpret_cast<size_t>(x) & 15) == 0) && p.x_bs % 8 == 0)
    hipLaunchKernelGGL((c1b_tap_kernel<true, 3>), dim3(p.per_xcd * NXCD), dim3(256), 0, st, p);  // (same sums, same order)
  else
    hipLaunchKernelGGL(c1b_tap_kernel<true>, dim3(p.per_xcd * NXCD), dim3(256), 0, st, p);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_conv1d_tap_pro_ok(int Cin, int Cout) { return Cin == 64 && Cout == 64 ? 1 : 0; }

int air_h_conv1d_tap_pro(int B, int Cin, int Cout, int T, int Tp, int dil, const unsigned short* x, size_t x_bs,
                         const unsigned short* w_packed, int dgrad, const float* bias, int relu, unsigned short* y,
                         size_t y_bs, void* stats, const unsigned short* bn_x, size_t bn_x_bs,
                         const unsigned short* bn_dy, size_t bn_dy_bs, const float* bn_mean, const float* bn_invstd,
                         void* bn_sums, const AirTapPrologue* pro, air_stream_t stream) {
  if (pro == nullptr || pro->kind == 0)
    return air_h_conv1d_tap_ex2(B, Cin, Cout, T, Tp, dil, x, x_bs, w_packed, dgrad, bias, relu, y, y_bs, stats, bn_x, bn_x_bs,
                                bn_dy, bn_dy_bs, bn_mean, bn_invstd, bn_sums, stream);
  if (!x || !w_packed || !y || B <= 0 || T <= 0 || Tp < T || Tp % BN != 0) return AIR_EINVAL;
  if (!air_h_conv1d_tap_pro_ok(Cin, Cout)) return AIR_EUNSUPPORTED;  // the whole K = 64 tile lives in the two staging buffers
  if (stats != nullptr && (reinterpret_cast<size_t>(stats) & 7)) return AIR_EINVAL;
  if (bn_sums != nullptr && (!bn_x || !bn_dy || !bn_mean || !bn_invstd || (reinterpret_cast<size_t>(bn_sums) & 15) ||
                             ((reinterpret_cast<size_t>(bn_x) | reinterpret_cast<size_t>(bn_dy)) & 15) ||
                             bn_x_bs % 8 || bn_dy_bs % 8))
    return AIR_EINVAL;
  if (dil < 1 || dil > TAP_MAXD) return AIR_EUNSUPPORTED;
  const int M = 64, K = 64;
  if ((pro->kind == 1) == (dgrad != 0) || (pro->kind != 1 && pro->kind != 2)) return AIR_EINVAL;
  if (!pro->g0 || !pro->side0 || !pro->pa || !pro->pb) return AIR_EINVAL;
  if (pro->kind == 1 && !pro->side1) return AIR_EINVAL;
  if (pro->kind == 2 && (!pro->pc || !pro->pd || !pro->pe)) return AIR_EINVAL;
  if ((reinterpret_cast<size_t>(pro->side0) | reinterpret_cast<size_t>(pro->side1) | reinterpret_cast<size_t>(x) |
       reinterpret_cast<size_t>(pro->g0) | reinterpret_cast<size_t>(pro->g1)) & 15)
    return AIR_EINVAL;  // 16-byte row pieces
  if (pro->side0_bs % 8 || pro->side1_bs % 8 || x_bs % 8 || pro->g0_bs % 8 || pro->g1_bs % 8) return AIR_EINVAL;
  // a side output must not be a tensor this launch reads with a halo (neighbouring workgroups would race)
  if (y == x || y == pro->g0 || y == pro->g1) return AIR_EINVAL;
  if (pro->side0 == x || pro->side0 == pro->g0 || pro->side0 == pro->g1 || pro->side1 == x || pro->side1 == pro->g0 ||
      (pro->side1 != nullptr && pro->side1 == pro->side0))
    return AIR_EINVAL;
  C1bTap p = {};
  p.x = reinterpret_cast<const float*>(x); p.a = w_packed; p.y = reinterpret_cast<float*>(y); p.bias = bias; p.acc = nullptr;
  p.x_bs = x_bs ? x_bs : (size_t)K * Tp; p.y_bs = y_bs ? y_bs : (size_t)M * Tp;
  p.B = B; p.M = M; p.K = K; p.T = T; p.dil = dil; p.relu = relu; p.Tp = Tp;
  p.stats = reinterpret_cast<float*>(stats);
  p.bsums = reinterpret_cast<float*>(bn_sums);
  p.bn_x = bn_x; p.bn_dy = bn_dy; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd;
  p.bn_x_bs = bn_x_bs ? bn_x_bs : (size_t)M * Tp; p.bn_dy_bs = bn_dy_bs ? bn_dy_bs : (size_t)M * Tp;
  p.g0 = pro->g0; p.g1 = pro->kind == 2 ? pro->g1 : nullptr;
  p.g0_bs = pro->g0_bs ? pro->g0_bs : (size_t)K * Tp; p.g1_bs = pro->g1_bs ? pro->g1_bs : (size_t)K * Tp;
  p.pa = pro->pa; p.pb = pro->pb; p.pc = pro->pc; p.pd = pro->pd; p.pe = pro->pe;
  p.invN = (float)(1.0 / ((double)B * (double)T));  // (air_h_bn_bwd_ex: (float)invN)
  p.side0 = pro->side0; p.side1 = pro->side1;
  p.side0_bs = pro->side0_bs ? pro->side0_bs : (size_t)K * Tp; p.side1_bs = pro->side1_bs ? pro->side1_bs : (size_t)K * Tp;
  p.tiles_m = 1;
  p.tiles_t = Tp / BN;
  p.total = B * p.tiles_t;
  p.per_xcd = (p.total + NXCD - 1) / NXCD;
  hipStream_t st = air_stream(stream);
  AirProfScope prof(AIR_K_C1B_TAP, 2.0 * B * T * (double)Cout * Cin * 3, st);
  if (pro->kind == 1)
    hipLaunchKernelGGL((c1b_tap_kernel<true, 1>), dim3(p.per_xcd * NXCD), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((c1b_tap_kernel<true, 2>), dim3(p.per_xcd * NXCD), dim3(256), 0, st, p);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

static int tapw_parts(int nb, int B, int W) {
  const int tiles = (W / 64) * (W / 64);
  int parts = 1024 / (nb * tiles);
  parts = parts < 1 ? 1 : parts;
  return parts < B ? parts : B;
}

size_t air_h_conv1d_tap_wgrad_ws_bytes(int n_branches, int B, int W) {
  if (n_branches <= 0 || B <= 0 || W <= 0 || W % 64 != 0) return 0;
  const size_t tiles = (size_t)(W / 64) * (W / 64);
  return (size_t)n_branches * tiles * tapw_parts(n_branches, B, W) * (3 * 64 * 64) * sizeof(float);
}

/* Weight gradients of the n_branches dilated K = 3 convs of one Res2 block (W -> W channels each) in one launch:
 * dw[i] (W, W, 3) fp32 = sum_{b,t} dy[i][b][co][t] x[i][b][ci][t + (k - 1) dil] over bf16-resident operands. */
int air_h_conv1d_tap_wgrad(int n_branches, int B, int W, int T, int Tp, int dil, const unsigned short* const* x,
                           const size_t* x_bs, const unsigned short* const* dy, const size_t* dy_bs, float* const* dw,
                           void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!x || !dy || !dw || !ws || n_branches <= 0 || B <= 0 || T <= 0 || Tp < T || Tp % TW_KC != 0) return AIR_EINVAL;
  if (n_branches > TW_MAXB || W % 64 != 0 || dil < 2 || dil > 4) return AIR_EUNSUPPORTED;
  if (ws_bytes < air_h_conv1d_tap_wgrad_ws_bytes(n_branches, B, W)) return AIR_EINVAL;
  TapWgrad a = {};
  for (int i = 0; i < n_branches; ++i) {
    if (!x[i] || !dy[i] || !dw[i]) return AIR_EINVAL;
    a.x[i] = x[i]; a.dy[i] = dy[i]; a.dw[i] = dw[i];
    a.x_bs[i] = (x_bs && x_bs[i]) ? x_bs[i] : (size_t)W * Tp;
    a.dy_bs[i] = (dy_bs && dy_bs[i]) ? dy_bs[i] : (size_t)W * Tp;
  }
  a.partial = reinterpret_cast<float*>(ws);
  a.nb = n_branches; a.B = B; a.W = W; a.Tp = Tp; a.tiles_c = W / 64;
  a.nparts = tapw_parts(n_branches, B, W);
  const int tiles = a.tiles_c * a.tiles_c;
  const int grid = n_branches * tiles * a.nparts;
  hipStream_t st = air_stream(stream);
  {
    AirProfScope prof(AIR_K_C1B_TAPW, 2.0 * n_branches * B * T * (double)W * W * 3, st);
    if (dil == 2) hipLaunchKernelGGL(c1b_tapw_kernel<2>, dim3(grid), dim3(256), 0, st, a);
    else if (dil == 3) hipLaunchKernelGGL(c1b_tapw_kernel<3>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(c1b_tapw_kernel<4>, dim3(grid), dim3(256), 0, st, a);
    AIR_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(c1b_tapw_reduce_kernel, dim3(n_branches * tiles * (3 * 64 * 64 / 64)), dim3(256), 0, st, a);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_h_tp(int T) { return T > 0 ? round_up(T, G2_BN) : 0; }

size_t air_h_conv1d_ws_bytes(int Cout, int Cin) { return align256((size_t)round_up(Cout, G2_BM) * Cin * 2) + 256; }

int air_h_conv1d_pointwise(int B, int Cin, int Cout, int T, int Tp, const unsigned short* x, size_t x_bs, const float* w,
                           int dgrad, const float* bias, const float* bias_bc, int relu, const unsigned short* acc,
                           size_t acc_bs, const unsigned short* acc2, size_t acc2_bs, unsigned short* y, size_t y_bs,
                           void* ws, size_t ws_bytes, air_stream_t stream) {
  return air_h_conv1d_pointwise_ex(B, Cin, Cout, T, Tp, x, x_bs, w, dgrad, bias, bias_bc, relu, acc, acc_bs, acc2, acc2_bs,
                                   y, y_bs, nullptr, ws, ws_bytes, stream);
}

size_t air_h_conv1d_pointwise_stats_bytes(int B, int Cout, int Tp) {
  if (B <= 0 || Cout <= 0 || Tp <= 0 || Tp % G2_BN != 0) return 0;
  return (size_t)Cout * B * (Tp / G2_BN) * 4 * 2 * sizeof(float);
}

int air_h_conv1d_pointwise_ex(int B, int Cin, int Cout, int T, int Tp, const unsigned short* x, size_t x_bs,
                              const float* w, int dgrad, const float* bias, const float* bias_bc, int relu,
                              const unsigned short* acc, size_t acc_bs, const unsigned short* acc2, size_t acc2_bs,
                              unsigned short* y, size_t y_bs, void* stats, void* ws, size_t ws_bytes,
                              air_stream_t stream) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || Tp < T) return AIR_EINVAL;
  if (stats != nullptr && (reinterpret_cast<size_t>(stats) & 7)) return AIR_EINVAL;
  const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  if (Tp % G2_BN != 0 || K % GK != 0 || M % 8 != 0 || !air_opt(AIR_OPT_C1B_GEMM_PS)) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_h_conv1d_ws_bytes(dgrad ? Cin : Cout, dgrad ? Cout : Cin)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  unsigned short* a = reinterpret_cast<unsigned short*>(ws);
  const size_t n2 = (size_t)M * K / 2;
  hipLaunchKernelGGL(c1b_pack_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, w, a, M, K, dgrad ? 1 : 0);
  AIR_CHECK_LAUNCH();
  NtGemm g = {};
  g.a = a; g.b = x; g.out = nullptr; g.bias = bias; g.bias_bc = bias_bc; g.acc = nullptr; g.acc2 = nullptr;
  g.acc_bs = 0; g.acc2_bs = 0;
  g.out_bf = y; g.out_bf_rs = Tp; g.out_bf_bs = y_bs ? y_bs : (size_t)M * Tp;
  g.acc_h = acc; g.acc2_h = acc2;
  g.acc_h_bs = acc_bs ? acc_bs : g.out_bf_bs; g.acc2_h_bs = acc2_bs ? acc2_bs : g.out_bf_bs;
  g.m_valid = M;
  g.a_rs = K; g.a_ss = 0; g.a_bs = 0;
  g.b_rs = Tp; g.b_ss = 0; g.b_bs = x_bs ? x_bs : (size_t)K * Tp;
  g.o_rs = Tp; g.o_bs = g.out_bf_bs;
  g.M = M; g.n_valid = T; g.kseg = K; g.nseg_per_batch = 1; g.nseg_total = B; g.relu = relu;
  g.hstats = reinterpret_cast<float*>(stats);
  const size_t a_bytes = (size_t)M * K * 2, b_bytes = ((size_t)(B - 1) * g.b_bs + (size_t)K * Tp) * 2;
  // 16-byte rows everywhere: row pitch Tp % 8 == 0 (above), bases and batch strides multiples of 8 elements
  if (((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(acc) |
        reinterpret_cast<size_t>(acc2)) & 15) || g.b_bs % 8 || g.out_bf_bs % 8 || g.acc_h_bs % 8 || g.acc2_h_bs % 8 ||
      a_bytes >= ((size_t)1 << 32) || b_bytes >= ((size_t)1 << 32))
    return AIR_EUNSUPPORTED;
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(c1b_gemm_ps_kernel<true, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) == hipSuccess;
  if (!attr_ok) return AIR_ELAUNCH;
  g.tiles_m = (M + G2_BM - 1) / G2_BM;
  g.tiles_n = Tp / G2_BN;
  g.total = g.tiles_m * g.tiles_n * B;
  g.per_xcd = (g.total + NXCD - 1) / NXCD;
  // algorithmic HBM bytes: packed weights + x rows read once, y rows written once, the accumulate operands read once
  const double alg_bytes = (double)a_bytes + 2.0 * B * (double)T * (K + M) + 2.0 * B * (double)T * M * ((acc ? 1 : 0) + (acc2 ? 1 : 0));
  AirProfScope prof(AIR_K_C1B_GEMM, 2.0 * B * T * (double)Cout * Cin, st, -1.0, alg_bytes);
  hipLaunchKernelGGL((c1b_gemm_ps_kernel<true, true>), dim3(256), dim3(512), G2_LDS, st, g, (unsigned)a_bytes,
                     (unsigned)b_bytes);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_conv1d_fwd_bf16_ex(const AirConv1d* p, const float* x, const float* w, const unsigned short* w_packed,
                           const float* bias, const float* bias_bc, int relu, float* y, unsigned short* y_bf16, void* ws,
                           size_t ws_bytes, air_stream_t stream) {
  if ((!shape_ok(p) && !tap_ok(p)) || !x || (!w && !(w_packed && p->K == 3)) || !y) return AIR_EINVAL;
  if (w_packed && p->K != 3) return AIR_EUNSUPPORTED;
  if (!air_conv1d_bf16_supported(p, 0)) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_bf16_ws_bytes(p)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  int rc;
  bool bf_done = false;
  if (p->K == 3) {
    if (bias_bc) return AIR_EUNSUPPORTED;
    rc = run_tap(x, xbs(p), w, 0, y, ybs(p), bias, nullptr, relu, p->B, p->Cout, p->Cin, p->T, p->dil, ws, st, w_packed);
  } else if (wide(p->Cout, p->Cin) && p->Cin % 64 == 0) {
    rc = run_fwd_gemm(x, xbs(p), w, 0, y, ybs(p), bias, bias_bc, nullptr, relu, p->B, p->Cout, p->Cin, p->T, ws,
                      2.0 * p->B * p->T * (double)p->Cout * p->Cin, st, y_bf16, &bf_done);
  } else {
    rc = run_fwd(x, xbs(p), w, 0, y, ybs(p), bias, bias_bc, nullptr, relu, p->B, p->Cout, p->Cin, p->T, ws,
                 2.0 * p->B * p->T * (double)p->Cout * p->Cin, st);
  }
  if (rc != AIR_OK) return rc;
  // y_bf16: the output's bf16 copy in the weight-gradient operand layout - from the GEMM's epilogue when that
  // kernel ran, by a conversion pass otherwise
  if (y_bf16 && !bf_done) return launch_cvt_s(y, ybs(p), p->B, p->Cout, p->T, y_bf16, st);
  return AIR_OK;
}

int air_conv1d_fwd_bf16(const AirConv1d* p, const float* x, const float* w, const float* bias, const float* bias_bc,
                        int relu, float* y, void* ws, size_t ws_bytes, air_stream_t stream) {
  return air_conv1d_fwd_bf16_ex(p, x, w, nullptr, bias, bias_bc, relu, y, nullptr, ws, ws_bytes, stream);
}

int air_conv1d_dgrad_bf16_ex(const AirConv1d* p, const float* dy, const float* w, const unsigned short* w_packed,
                             float* dx, const float* accumulate, size_t acc_bstride, const float* accumulate2,
                             size_t acc2_bstride, void* ws, size_t ws_bytes, air_stream_t stream) {
  if ((!shape_ok(p) && !tap_ok(p)) || !dy || (!w && !(w_packed && p->K == 3)) || !dx) return AIR_EINVAL;
  if (w_packed && p->K != 3) return AIR_EUNSUPPORTED;
  if (!air_conv1d_bf16_supported(p, 1)) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_bf16_ws_bytes(p)) return AIR_EWORKSPACE;
  const bool plain = !accumulate2 && (acc_bstride == 0 || acc_bstride == xbs(p));
  if (p->K == 3) {
    if (!plain) return AIR_EUNSUPPORTED;
    return run_tap(dy, ybs(p), w, 1, dx, xbs(p), nullptr, accumulate, 0, p->B, p->Cout, p->Cin, p->T, p->dil, ws,
                   air_stream(stream), w_packed);
  }
  // A = W^T: w is (Cout, Cin) = [k][m]
  if (wide(p->Cin, p->Cout) && p->Cout % 64 == 0) {
    if (!plain) return AIR_EUNSUPPORTED;
    return run_fwd_gemm(dy, ybs(p), w, 1, dx, xbs(p), nullptr, nullptr, accumulate, 0, p->B, p->Cin, p->Cout, p->T, ws,
                        2.0 * p->B * p->T * (double)p->Cout * p->Cin, air_stream(stream));
  }
  return run_fwd(dy, ybs(p), w, 1, dx, xbs(p), nullptr, nullptr, accumulate, 0, p->B, p->Cin, p->Cout, p->T, ws,
                 2.0 * p->B * p->T * (double)p->Cout * p->Cin, air_stream(stream), acc_bstride, accumulate2, acc2_bstride);
}

int air_conv1d_dgrad_bf16(const AirConv1d* p, const float* dy, const float* w, float* dx, const float* accumulate,
                          void* ws, size_t ws_bytes, air_stream_t stream) {
  return air_conv1d_dgrad_bf16_ex(p, dy, w, nullptr, dx, accumulate, 0, nullptr, 0, ws, ws_bytes, stream);
}

namespace {
int launch_cvt_s(const float* x, size_t x_bs, int B, int C, int T, unsigned short* out, hipStream_t st) {
  const int Tp = round_up(T, TP_ALIGN);
  const int vec = T % 2 == 0 && x_bs % 2 == 0 && (reinterpret_cast<size_t>(x) & 7) == 0;
  const size_t ch = (size_t)B * C * Tp / 8;
  hipLaunchKernelGGL(c1b_cvt_s_kernel, dim3((unsigned)((ch + 255) / 256)), dim3(256), 0, st, x, x_bs, C, T, Tp, ch, vec, out);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}
}  // namespace

int air_conv1d_bf16_tp(int T) { return T > 0 ? round_up(T, TP_ALIGN) : 0; }

size_t air_conv1d_tap_pack_elems(int Cout, int Cin) { return (size_t)3 * Cout * Cin; }

int air_conv1d_tap_pack_bf16(const float* w, size_t w_stride, int n_layers, int Cout, int Cin, int transpose,
                             unsigned short* out, air_stream_t stream) {
  if (!w || !out || n_layers <= 0 || Cout <= 0 || Cin <= 0) return AIR_EINVAL;
  const size_t n = (size_t)3 * Cout * Cin;
  hipLaunchKernelGGL(c1b_pack3_kernel, dim3((unsigned)((n + 255) / 256), n_layers), dim3(256), 0, air_stream(stream), w, out,
                     Cout, Cin, transpose, w_stride);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_conv1d_cvt_bf16(const float* x, size_t x_bstride, int B, int C, int T, unsigned short* out, air_stream_t stream) {
  if (!x || !out || B <= 0 || C <= 0 || T <= 0) return AIR_EINVAL;
  return launch_cvt_s(x, x_bstride ? x_bstride : (size_t)C * T, B, C, T, out, air_stream(stream));
}

int air_conv1d_wgrad_bf16_pre(const AirConv1d* p, const float* x, const float* dy, const unsigned short* x_bf16,
                              size_t x_bf16_bstride, const unsigned short* dy_bf16, size_t dy_bf16_bstride, float* dw,
                              void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!shape_ok(p) || (!x && !x_bf16) || (!dy && !dy_bf16) || !dw) return AIR_EINVAL;
  if (!air_conv1d_bf16_supported(p, 2)) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_bf16_ws_bytes(p)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  // bf16 copies of both operands ([b][row][Tp], zero-padded frames) - made here unless the caller holds them
  // already (written by the tensor's producer, or converted once for several layers) - then the split-K GEMM
  const int B = p->B, M = p->Cout, N = p->Cin, T = p->T, Tp = round_up(T, TP_ALIGN);
  int per;
  const int nsplit = wgrad_nsplit(p, &per);
  char* base = reinterpret_cast<char*>(ws);
  float* partial = reinterpret_cast<float*>(base);
  unsigned short* dys = reinterpret_cast<unsigned short*>(base + align256((size_t)nsplit * M * N * sizeof(float)));
  unsigned short* xs = dys + align256((size_t)B * Tp * M * 2) / 2;
  size_t a_ss = (size_t)M * Tp, b_ss = (size_t)N * Tp;
  if (dy_bf16) {
    dys = const_cast<unsigned short*>(dy_bf16);
    if (dy_bf16_bstride) a_ss = dy_bf16_bstride;
  } else {
    const int rc = launch_cvt_s(dy, ybs(p), B, M, T, dys, st);
    if (rc != AIR_OK) return rc;
  }
  if (x_bf16) {
    xs = const_cast<unsigned short*>(x_bf16);
    if (x_bf16_bstride) b_ss = x_bf16_bstride;
  } else {
    const int rc = launch_cvt_s(x, xbs(p), B, N, T, xs, st);
    if (rc != AIR_OK) return rc;
  }
  NtGemm g;
  g.hstats = nullptr;
  g.a = dys; g.b = xs; g.out = nsplit > 1 ? partial : dw; g.bias = nullptr; g.bias_bc = nullptr; g.acc = nullptr;
  g.acc2 = nullptr; g.acc_bs = 0; g.acc2_bs = 0;
  g.out_bf = nullptr; g.out_bf_rs = 0;
  g.a_rs = Tp; g.a_ss = a_ss; g.a_bs = (size_t)per * g.a_ss;
  g.b_rs = Tp; g.b_ss = b_ss; g.b_bs = (size_t)per * g.b_ss;
  g.o_rs = N; g.o_bs = (size_t)M * N;
  g.M = M; g.n_valid = N; g.kseg = Tp; g.nseg_per_batch = per; g.nseg_total = B; g.relu = 0;
  int rc = launch_gemm(g, nsplit, N, ((size_t)(B - 1) * a_ss + (size_t)M * Tp) * 2, ((size_t)(B - 1) * b_ss + (size_t)N * Tp) * 2,
                       AIR_K_C1B_GEMM, 2.0 * B * T * (double)M * N, st, nullptr, false,
                       2.0 * B * (double)T * (M + N) + 4.0 * M * (double)N);
  if (rc != AIR_OK) return rc;
  if (nsplit > 1) {
    const size_t n = (size_t)M * N;
    hipLaunchKernelGGL(c1b_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, partial, dw, n,
                       nsplit);
    AIR_CHECK_LAUNCH();
  }
  return AIR_OK;
}

/* Weight gradient of a K = 1 layer from bf16-resident operands: dw[co][ci] = sum_{b,t} dy[b][co][t] x[b][ci][t], both
 * operands bf16 rows of Tp frames (zeros behind T), batch strides in elements (0 = dense).  Same split-K GEMM and
 * fixed-order reduction as air_conv1d_wgrad_bf16_pre; ws of air_conv1d_bf16_ws_bytes for the layer. */
int air_h_conv1d_wgrad(int B, int Cin, int Cout, int T, int Tp, const unsigned short* x, size_t x_bs,
                       const unsigned short* dy, size_t dy_bs, float* dw, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!x || !dy || !dw || B <= 0 || T <= 0 || Tp < T || Tp % TP_ALIGN != 0) return AIR_EINVAL;
  AirConv1d q = {B, Cin, T, Cout, 1, 1, 0, 0, 0};
  if (!air_conv1d_bf16_supported(&q, 2)) return AIR_EUNSUPPORTED;
  int per;
  const int nsplit = wgrad_nsplit(&q, &per);
  const int M = Cout, N = Cin;
  if (!ws || ws_bytes < align256((size_t)nsplit * M * N * sizeof(float))) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  float* partial = reinterpret_cast<float*>(ws);
  const size_t a_ss = dy_bs ? dy_bs : (size_t)M * Tp, b_ss = x_bs ? x_bs : (size_t)N * Tp;
  NtGemm g = {};
  g.a = dy; g.b = x; g.out = nsplit > 1 ? partial : dw;
  g.a_rs = Tp; g.a_ss = a_ss; g.a_bs = (size_t)per * g.a_ss;
  g.b_rs = Tp; g.b_ss = b_ss; g.b_bs = (size_t)per * g.b_ss;
  g.o_rs = N; g.o_bs = (size_t)M * N;
  g.M = M; g.n_valid = N; g.kseg = Tp; g.nseg_per_batch = per; g.nseg_total = B; g.relu = 0;
  int rc = launch_gemm(g, nsplit, N, ((size_t)(B - 1) * a_ss + (size_t)M * Tp) * 2, ((size_t)(B - 1) * b_ss + (size_t)N * Tp) * 2,
                       AIR_K_C1B_GEMM, 2.0 * B * T * (double)M * N, st, nullptr, false,
                       2.0 * B * (double)T * (M + N) + 4.0 * M * (double)N);
  if (rc != AIR_OK) return rc;
  if (nsplit > 1) {
    const size_t n = (size_t)M * N;
    hipLaunchKernelGGL(c1b_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, partial, dw, n,
                       nsplit);
    AIR_CHECK_LAUNCH();
  }
  return AIR_OK;
}

int air_conv1d_wgrad_bf16(const AirConv1d* p, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                          air_stream_t stream) {
  if (!x || !dy) return AIR_EINVAL;
  return air_conv1d_wgrad_bf16_pre(p, x, dy, nullptr, 0, nullptr, 0, dw, ws, ws_bytes, stream);
}

}  // extern "C"
