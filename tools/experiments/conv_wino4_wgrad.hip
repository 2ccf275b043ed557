// Weight gradient of the 3x3 / stride 1 / pad 1 convolution (resnet.py:56-61) as Winograd F(3x3,4x4) on
// v_mfma_f32_16x16x4_f32:
//   dW[co][ci] (3x3) = sum over 4x4 dy tiles of  A^T [ (G g G^T) .* (B^T d B) ] A
// with g the 4x4 tile of dy[co], d the 6x6 patch of x[ci] around it (the forward kernel's patch and its
// B^T d B, conv_wino4.hip).  36 multiplies per (co, ci, tile) instead of 144 (F(3x3,2x2), conv_wino.hip:
// 16 per 2x2 tile = 64): 36 GEMMs  M_p[co][ci] = sum_tiles Gd_p[co][tile] V_p[tile][ci]  whose K is the
// tile stream, 4 tiles (one MFMA k) per step.
//   * a wave owns 32 co (2 MFMA row blocks) x 16 ci: 72 accumulators x 4 registers, 64 pinned to AGPRs
//     and 8 to VGPRs as in conv_wino4.hip; a workgroup = 2 co halves x 2 ci blocks = 64 co x 32 ci; the
//     tile stream is K-split over workgroups, partial 3x3 results go to reduce_partials_kernel.
//   * BOTH operands are computed, not loaded: a lane transforms the dy tile of ITS two channels and ITS tile
//     (A layout: lane = 16 tile + co; 2 x 80 operations) and the x patch of ITS channel and tile (B layout:
//     lane = 16 tile + ci; 144 operations), so the MFMA block is 72 back-to-back MFMAs with only the DMA
//     issue in between, and the VALU work is one batch (the f32 MFMA shares the issue port with the VALU).
//   * G's scale factors (1/4, -1/6, -1/6, 1/24, 1/24, 1) are applied once, in the output transform.
//   * staging: hand-issued LDS-DMA of 16-byte chunks through 3-deep rings.  x: [row 6][ci 32][8 chunks]
//     (image columns 16 tg - 4 ...: a chunk is inside or outside the row as a whole; the chunk index is
//     XOR-ed with (ci >> 1) & 7 on the SOURCE side so 16 channels read 16 different bank groups);
//     dy: [co 64][16 chunks = 4 rows x 4 tiles], chunk index XOR-ed with co & 15.  Zero padding = the buffer
//     descriptor's out-of-range rule; per step one select per operand picks the out-of-range offset for the
//     chunks this tile row / tile group does not have.  The chunk that straddles the right image edge
//     (W % 4 != 0) is cleaned up in LDS by the reading wave.
#include <stdlib.h>

#include <type_traits>

#include "air_common.h"
#include "air_lds_dma.h"
#include "air_prof.h"
#include "conv_wino.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int G4_CO = 64, G4_CI = 32;
constexpr int G4_DYF = G4_CO * 64;        // floats of dy per step: [co][16 chunks]
constexpr int G4_XROWF = G4_CI * 32;      // floats of one staged x row: [ci][8 chunks]
constexpr int G4_XF = 6 * G4_XROWF;       // floats of x per step
constexpr int G4_BUF = G4_DYF + G4_XF;    // 10240 floats = 40 KB
constexpr int G4_NBUF = 3;
constexpr int G4_ND = 10;                 // DMAs per thread and step: 4 dy + 6 x
constexpr unsigned G4_OOB = 0x80000000u;
constexpr unsigned G4_XBIAS = 65536;      // the x descriptor starts this many bytes before x (row -1, column -4)

struct G4Args {
  const float* x;    // (B, Cin, H, W)
  const float* dy;   // (B, Cout, H, W)
  float* partial;    // [nsplit][9][Cout][Cin]
  int B, Cin, H, W, Cout;
  int TH, TW4;       // tile rows per image, groups of 4 tiles per tile row
  int nstep;         // B * TH * TW4
  int ncob, ncib, nsplit;
};

__device__ __forceinline__ i32x4 g4_rsrc(const void* base, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int MIMM>
__device__ __forceinline__ void g4_dma16(i32x4 rsrc, unsigned soff, unsigned mbase, unsigned v0) {
  asm volatile("s_add_i32 m0, %1, %4\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %0, %2 offen lds"
               :: "s"(rsrc), "s"(mbase), "s"(soff), "v"(v0), "n"(MIMM) : "memory", "m0", "scc");
}
template <int N>
__device__ __forceinline__ void g4_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#define G4_MFMA_A(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define G4_MFMA_V(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

// B^T d (conv_wino4.hip): 12 operations
__device__ __forceinline__ void g4_bt6(float d0, float d1, float d2, float d3, float d4, float d5, float& t0,
                                       float& t1, float& t2, float& t3, float& t4, float& t5) {
  t0 = __builtin_fmaf(4.0f, d0, __builtin_fmaf(-5.0f, d2, d4));
  const float a = __builtin_fmaf(-4.0f, d2, d4), b = __builtin_fmaf(-4.0f, d1, d3);
  t1 = a + b;
  t2 = a - b;
  const float c = d4 - d2, e = d3 - d1;
  t3 = __builtin_fmaf(2.0f, e, c);
  t4 = __builtin_fmaf(-2.0f, e, c);
  t5 = __builtin_fmaf(4.0f, d1, __builtin_fmaf(-5.0f, d3, d5));
}
// G g without its row scales: [g0, g0+g1+g2+g3, g0-g1+g2-g3, g0+2g1+4g2+8g3, g0-2g1+4g2-8g3, g3]: 8 operations
__device__ __forceinline__ void g4_g6(float g0, float g1, float g2, float g3, float& u0, float& u1, float& u2,
                                      float& u3, float& u4, float& u5) {
  const float e = g0 + g2, o = g1 + g3;
  const float p = __builtin_fmaf(4.0f, g2, g0), q = __builtin_fmaf(4.0f, g3, g1);
  u0 = g0;
  u1 = e + o;
  u2 = e - o;
  u3 = __builtin_fmaf(2.0f, q, p);
  u4 = __builtin_fmaf(-2.0f, q, p);
  u5 = g3;
}
// A^T m, A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,1]]
__device__ __forceinline__ void g4_at3(float m0, float m1, float m2, float m3, float m4, float m5, float& y0,
                                       float& y1, float& y2) {
  const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  y0 = m0 + s1 + s2;
  y1 = __builtin_fmaf(2.0f, d2, d1);
  y2 = __builtin_fmaf(4.0f, s2, s1) + m5;
}

__global__ __launch_bounds__(256) void wino4_wgrad_kernel(G4Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 1;    // co half (32 channels) of the workgroup's 64
  const int bw = wave >> 1;   // ci block (16 channels) of the workgroup's 32
  const int HWi = a.H * a.W;
  const int wrem = a.W & 3;

  int lb = blockIdx.x;
  const int split = lb % a.nsplit;
  lb /= a.nsplit;
  const int cib = lb % a.ncib;
  const int cob = lb / a.ncib;
  const int q0 = (int)((long long)split * a.nstep / a.nsplit), q1 = (int)((long long)(split + 1) * a.nstep / a.nsplit);

  const i32x4 xrs = g4_rsrc(reinterpret_cast<const char*>(a.x) - G4_XBIAS, (unsigned)a.B * a.Cin * HWi * 4u + G4_XBIAS);
  const i32x4 drs = g4_rsrc(a.dy, (unsigned)a.B * a.Cout * HWi * 4u);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  const unsigned mD0 = lds0 + wave * 1024u;                 // dy DMA i: + i * 4096
  const unsigned mX0 = lds0 + G4_DYF * 4u + wave * 1024u;   // x  DMA i (= patch row i): + i * 4096

  // ---- per-lane DMA constants (one set serves every DMA of its operand)
  // dy slot: co_l = tid >> 4 (+ 16 i), LDS chunk tid & 15 holds logical chunk (tid & 15) ^ (co_l & 15) = 4 r + c
  const int d_ch = (tid & 15) ^ ((tid >> 4) & 15);
  const int d_r = d_ch >> 2, d_c = d_ch & 3;
  const unsigned d_lane = (unsigned)((((tid >> 4) * a.H + d_r) * a.W + 4 * d_c) * 4);
  // x slot: ci_l = tid >> 3, LDS chunk tid & 7 holds logical chunk (tid & 7) ^ ((ci_l >> 1) & 7) = cc (< 6 valid)
  const int x_ci = tid >> 3;
  const int x_cc = (tid & 7) ^ ((x_ci >> 1) & 7);
  const unsigned x_lane = (unsigned)((x_ci * HWi + 4 * x_cc) * 4);

  // ---- staging cursor: step q -> (b, th, tg); runs 3 steps ahead of the compute cursor, clamped at the end
  int sq = q0, sb, sth, stg, sbuf = 0;
  {
    stg = sq % a.TW4;
    const int r = sq / a.TW4;
    sth = r % a.TH;
    sb = r / a.TH;
  }
  unsigned dvo = 0, xvo = 0, dso = 0, xso = 0, mD = mD0, mX = mX0;
  unsigned xrow_ok = 0;  // bit i: patch row i of the staged step lies inside the image
  auto stage_setup = [&]() {
    // dy chunk (r, c): image row 4 th + r, columns 16 tg + 4 c ...
    const bool dok = 4 * sth + d_r < a.H && 16 * stg + 4 * d_c < a.W;
    dvo = dok ? d_lane : G4_OOB;
    dso = __builtin_amdgcn_readfirstlane((unsigned)((((sb * a.Cout + cob * G4_CO) * a.H + 4 * sth) * a.W + 16 * stg) * 4));
    // x chunk cc: columns 16 tg - 4 + 4 cc ...; row i of the patch: image row 4 th - 1 + i
    const int col0 = 16 * stg - 4 + 4 * x_cc;
    const bool xok = x_cc < 6 && col0 >= 0 && col0 < a.W;
    xvo = xok ? x_lane : G4_OOB;
    xso = __builtin_amdgcn_readfirstlane(   // (the descriptor's bias keeps row -1 / column -4 non-negative)
        (unsigned)((((sb * a.Cin + cib * G4_CI) * a.H + 4 * sth - 1) * a.W + 16 * stg - 4) * 4 + (int)G4_XBIAS));
    unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) ok |= (4 * sth - 1 + i >= 0 && 4 * sth - 1 + i < a.H) ? (1u << i) : 0u;
    xrow_ok = __builtin_amdgcn_readfirstlane(ok);
  };
  auto stage_advance = [&]() {
    sbuf = sbuf + 1 == G4_NBUF ? 0 : sbuf + 1;
    if (sq + 1 < q1) {
      ++sq;
      if (++stg == a.TW4) {
        stg = 0;
        if (++sth == a.TH) {
          sth = 0;
          ++sb;
        }
      }
    }
    mD = __builtin_amdgcn_readfirstlane(mD0 + (unsigned)sbuf * (G4_BUF * 4u));
    mX = __builtin_amdgcn_readfirstlane(mX0 + (unsigned)sbuf * (G4_BUF * 4u));
    stage_setup();
  };
  auto dma_unit = [&](auto unit_tag) {
    constexpr int u = decltype(unit_tag)::value;
    if constexpr (u < 4) {
      g4_dma16<u * 4096>(drs, dso + (unsigned)u * 16u * (unsigned)HWi * 4u, mD, dvo);
    } else {
      constexpr int i = u - 4;
      const unsigned vo = (xrow_ok >> i) & 1u ? xvo : G4_OOB;  // wave-uniform choice of the offset register
      g4_dma16<i * 4096>(xrs, xso + (unsigned)i * (unsigned)a.W * 4u, mX, vo);
    }
  };
#define G4_UNIT(U_) dma_unit(std::integral_constant<int, U_>{})
  auto dma_all = [&]() {
    G4_UNIT(0); G4_UNIT(1); G4_UNIT(2); G4_UNIT(3); G4_UNIT(4);
    G4_UNIT(5); G4_UNIT(6); G4_UNIT(7); G4_UNIT(8); G4_UNIT(9);
    stage_advance();
  };

  // ---- compute-side lane constants
  const int j16 = tid & 15, t4 = (tid >> 4) & 3;   // channel within a 16-block, tile of the step's 4
  // dy: channel co_l = 32 cw + 16 cbk + j16, chunk (4 r + t4) ^ (co_l & 15) = (4 r + t4) ^ j16
  int dyo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dyo[r] = (cw * 32 + j16) * 64 + (((4 * r + t4) ^ j16) << 2);
  // x: channel ci_l = 16 bw + j16; chunks t4, t4 + 1, t4 + 2 XOR-ed with (ci_l >> 1) & 7
  const int ci_l = bw * 16 + j16, xk = (ci_l >> 1) & 7;
  const int xo0 = G4_DYF + ci_l * 32 + ((t4 ^ xk) << 2) + 3;
  const int xo1 = G4_DYF + ci_l * 32 + (((t4 + 1) ^ xk) << 2);
  const int xo2 = G4_DYF + ci_l * 32 + (((t4 + 2) ^ xk) << 2);

  f32x4 accA[64], accV[8];
#pragma unroll
  for (int i = 0; i < 64; ++i) accA[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < 8; ++i) accV[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  float Gd[2][36], V[36];
  float rx[36], rd[2][16];

  // right image edge: the chunk that starts at column W - W % 4 carries the next row's pixels behind column W
  auto fix_edges = [&](float* buf, int tg) {
    if (wrem == 0) return;
    const int ec = a.W - wrem;
    const int xc = (ec - (16 * tg - 4)) >> 2;  // x chunk of that column
    const int lane = tid & 63;
    if (xc >= 0 && xc < 6) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pr = lane + 64 * h;  // (ci 16) x (row 6) pairs of this wave's channels
        if (pr < 96) {
          const int c = bw * 16 + pr / 6, r = pr % 6;
          float* p = buf + G4_DYF + r * G4_XROWF + c * 32 + ((xc ^ ((c >> 1) & 7)) << 2);
          if (wrem <= 1) p[1] = 0.0f;
          if (wrem <= 2) p[2] = 0.0f;
          p[3] = 0.0f;
        }
      }
    }
    const int dc = (ec - 16 * tg) >> 2;  // dy chunk column
    if (dc >= 0 && dc < 4) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = cw * 32 + (lane & 31), r = (lane >> 5) + 2 * h;
        float* p = buf + c * 64 + (((4 * r + dc) ^ (c & 15)) << 2);
        if (wrem <= 1) p[1] = 0.0f;
        if (wrem <= 2) p[2] = 0.0f;
        p[3] = 0.0f;
      }
    }
  };
  auto read_raw = [&](const float* buf) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      rx[6 * r] = buf[r * G4_XROWF + xo0];
      const f32x4 m = *reinterpret_cast<const f32x4*>(buf + r * G4_XROWF + xo1);
      rx[6 * r + 1] = m[0]; rx[6 * r + 2] = m[1]; rx[6 * r + 3] = m[2]; rx[6 * r + 4] = m[3];
      rx[6 * r + 5] = buf[r * G4_XROWF + xo2];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(buf + k * (16 * 64) + dyo[r]);
        rd[k][4 * r] = m[0]; rd[k][4 * r + 1] = m[1]; rd[k][4 * r + 2] = m[2]; rd[k][4 * r + 3] = m[3];
      }
  };
  auto transform = [&]() {
    float t[36];
#pragma unroll
    for (int r = 0; r < 6; ++r)
      g4_bt6(rx[6 * r], rx[6 * r + 1], rx[6 * r + 2], rx[6 * r + 3], rx[6 * r + 4], rx[6 * r + 5],
             t[6 * r], t[6 * r + 1], t[6 * r + 2], t[6 * r + 3], t[6 * r + 4], t[6 * r + 5]);
#pragma unroll
    for (int c = 0; c < 6; ++c)
      g4_bt6(t[c], t[6 + c], t[12 + c], t[18 + c], t[24 + c], t[30 + c],
             V[c], V[6 + c], V[12 + c], V[18 + c], V[24 + c], V[30 + c]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float h[4][6];
#pragma unroll
      for (int r = 0; r < 4; ++r)  // along the row: 4 columns -> 6
        g4_g6(rd[k][4 * r], rd[k][4 * r + 1], rd[k][4 * r + 2], rd[k][4 * r + 3],
              h[r][0], h[r][1], h[r][2], h[r][3], h[r][4], h[r][5]);
#pragma unroll
      for (int c = 0; c < 6; ++c)  // down the column: 4 rows -> 6
        g4_g6(h[0][c], h[1][c], h[2][c], h[3][c],
              Gd[k][c], Gd[k][6 + c], Gd[k][12 + c], Gd[k][18 + c], Gd[k][24 + c], Gd[k][30 + c]);
    }
  };

  if (q0 < q1) {
    // prologue: stage q0, q0+1, q0+2; wait for the first; its operands
    stage_setup();
    dma_all();
    dma_all();
    dma_all();
    g4_wait<2 * G4_ND>();
    __syncthreads();
    int ctg = q0 % a.TW4;
    fix_edges(lds, ctg);
    read_raw(lds);
    transform();
    int cur = 0;
    for (int q = q0; q < q1; ++q) {
      const int nxt = cur + 1 == G4_NBUF ? 0 : cur + 1;
      g4_wait<G4_ND>();   // step q + 1 has landed (q + 2 may still fly)
      __syncthreads();    // ... for every wave, and everyone is done with buffer `cur` (read last step)
      const int ntg = ctg + 1 == a.TW4 ? 0 : ctg + 1;
      float* bn = lds + nxt * G4_BUF;
      fix_edges(bn, ntg);
      read_raw(bn);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1");
#pragma unroll
      for (int g = 0; g < 18; ++g) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int p = 4 * (g % 9) + qq, k = g / 9, acc = k * 36 + p;
          if (acc < 64) G4_MFMA_A(accA[acc < 64 ? acc : 0], Gd[k][p], V[p]);
          else G4_MFMA_V(accV[acc >= 64 ? acc - 64 : 0], Gd[k][p], V[p]);
        }
        // restaging of step q + 3 into the buffer step q left, one DMA per group boundary
        switch (g) {
          case 0: G4_UNIT(0); break;
          case 1: G4_UNIT(1); break;
          case 2: G4_UNIT(2); break;
          case 3: G4_UNIT(3); break;
          case 4: G4_UNIT(4); break;
          case 5: G4_UNIT(5); break;
          case 6: G4_UNIT(6); break;
          case 7: G4_UNIT(7); break;
          case 8: G4_UNIT(8); break;
          case 9: G4_UNIT(9); break;
          default: break;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_nop 15");
      stage_advance();
      transform();
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
      ctg = ntg;
    }
  }
  asm volatile("s_nop 15");
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // dW = A^T (s s^T .* M) A, s = (1/4, -1/6, -1/6, 1/24, 1/24, 1); D row (l >> 4) * 4 + r -> co, column -> ci
  const float sc[6] = {0.25f, -1.0f / 6.0f, -1.0f / 6.0f, 1.0f / 24.0f, 1.0f / 24.0f, 1.0f};
  float* __restrict__ out = a.partial + (size_t)split * 9 * a.Cout * a.Cin;
  const int ci = cib * G4_CI + bw * 16 + j16;
#pragma unroll
  for (int kr = 0; kr < 8; ++kr) {
    const int k = kr >> 2, r = kr & 3;
    const int co = cob * G4_CO + cw * 32 + k * 16 + 4 * t4 + r;
    float T[3][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float m[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int acc = k * 36 + 6 * i + j;
        m[i] = (acc < 64 ? accA[acc < 64 ? acc : 0][r] : accV[acc >= 64 ? acc - 64 : 0][r]) * (sc[i] * sc[j]);
      }
      g4_at3(m[0], m[1], m[2], m[3], m[4], m[5], T[0][j], T[1][j], T[2][j]);
    }
#pragma unroll
    for (int ta = 0; ta < 3; ++ta) {
      float y0, y1, y2;
      g4_at3(T[ta][0], T[ta][1], T[ta][2], T[ta][3], T[ta][4], T[ta][5], y0, y1, y2);
      out[((size_t)(3 * ta + 0) * a.Cout + co) * a.Cin + ci] = y0;
      out[((size_t)(3 * ta + 1) * a.Cout + co) * a.Cin + ci] = y1;
      out[((size_t)(3 * ta + 2) * a.Cout + co) * a.Cin + ci] = y2;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

bool air_wino4_wgrad_ok(int B, int Cin, int H, int W, int Cout) {
  static const int off = getenv("AIR_NO_WINO4_WGRAD") ? atoi(getenv("AIR_NO_WINO4_WGRAD")) : 0;
  if (off) return false;
  if (Cin % G4_CI != 0 || Cout % G4_CO != 0 || W < 4) return false;
  const double ein = (double)B * Cin * H * W, eout = (double)B * Cout * H * W;
  return ein * 4.0 + 131072.0 < 2147483648.0 && eout * 4.0 < 2147483648.0;
}

int air_wino4_wgrad_nsplit(int B, int Cin, int H, int W, int Cout) {
  const int nstep = B * ((H + 3) / 4) * (((W + 3) / 4 + 3) / 4);
  int n = 256 / ((Cin / G4_CI) * (Cout / G4_CO));
  if (n < 1) n = 1;
  if (n > nstep) n = nstep;
  return n;
}

int air_wino4_wgrad_partials(const float* x, const float* dy, float* partial, int B, int Cin, int H, int W,
                             int Cout, double flops, hipStream_t st) {
  G4Args a;
  a.x = x; a.dy = dy; a.partial = partial;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
  a.TH = (H + 3) / 4;
  a.TW4 = ((W + 3) / 4 + 3) / 4;
  a.nstep = B * a.TH * a.TW4;
  a.ncob = Cout / G4_CO; a.ncib = Cin / G4_CI;
  a.nsplit = air_wino4_wgrad_nsplit(B, Cin, H, W, Cout);
  const size_t ldsb = (size_t)G4_NBUF * G4_BUF * sizeof(float);
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_wgrad_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) == hipSuccess;
  if (!attr_ok) return AIR_ELAUNCH;
  AirProfScope ps(AIR_K_CONV_WINO4_WG, flops, st);
  hipLaunchKernelGGL(wino4_wgrad_kernel, dim3(a.ncob * a.ncib * a.nsplit), dim3(256), ldsb, st, a);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}
