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

#include <type_traits>

#include "air_common.h"
#include "air_options.h"
#include "air_lds_dma.h"
#include "air_prof.h"
#include "conv_wino.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2a4 __attribute__((ext_vector_type(2), aligned(4)));

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
  int nitems;             // work items: ceil(ngroups / 2) * ncot
  long long* trace;       // debug: cycle stamps of workgroup 0 (null in production)
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

// Position of a tile group / of a work item, wave-uniform.  A work item = (64 output
// channels) x (2 consecutive tile groups); items are numbered cot-fastest.
struct GroupPos {
  int b, thg, twg;
};
struct ItemPos {
  int cot;
  GroupPos g[2];
};
// Groups are numbered DOWN the image first (tile row fastest): the two groups of a work item and
// the items a workgroup walks next are vertical neighbours, which share 2 of their 4 input rows
// while those are still in L1/L2 (numbered along the row, FETCH_SIZE showed 3x the input bytes).
__device__ __forceinline__ void group_init(GroupPos& g, int idx, const WinoArgs& a) {
  g.thg = idx % a.THG;
  const int r = idx / a.THG;
  g.twg = r % a.TWG;
  g.b = r / a.TWG;  // == B for the padding group behind an odd group count
}
__device__ __forceinline__ void item_init(ItemPos& it, int item, const WinoArgs& a) {
  it.cot = item % a.ncot;
  const int pg = item / a.ncot;
  group_init(it.g[0], 2 * pg, a);
  group_init(it.g[1], 2 * pg + 1, a);
}
// ---- inline-asm building blocks -------------------------------------------------------
// A wave's VALU / LDS / DMA instructions do NOT run under its (or a sibling wave's) f32 MFMAs on
// this part: tools/ubench/mfma_rot.hip measures 64 + 8 + 4 n cycles per MFMA with n VALU
// instructions in between, at 1, 2 or 4 waves per SIMD alike.  So the kernel is built to issue
// as few non-MFMA instructions per MFMA as possible:
//   * staging is buffer_load ... lds: one 32-bit offset VGPR per element, the chunk advance is
//     the SGPR soffset, the zero padding is the buffer's out-of-range rule (no mask
//     instructions, no address arithmetic in the loop), 4 DMAs share one M0 write through the
//     instruction offset (it moves the LDS and the global address alike, so the per-lane offset
//     is biased by it);
//   * the 4x4 input transform is 16 v_pk_add_f32 (source-half selects + negate modifiers).
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned XBIAS = 4096;         // the x descriptor starts this many bytes before x
constexpr unsigned OOB = 0x80000000u;    // byte offset beyond any tensor we accept: reads as zero

__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32));  // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// One DMA = 3 instructions: M0 = (wave's LDS base for this buffer) + MIMM, wait state, load.
// IMM = instruction offset (moves the LDS and the global address alike).
template <int MIMM, int IMM>
__device__ __forceinline__ void dma4_buf(i32x4 rsrc, unsigned soff, unsigned mbase, unsigned v0) {
  asm volatile("s_add_i32 m0, %1, %4\n\ts_nop 0\n\t"
               "buffer_load_dword %3, %0, %2 offen offset:%5 lds"
               :: "s"(rsrc), "s"(mbase), "s"(soff), "v"(v0), "n"(MIMM), "n"(IMM) : "memory", "m0", "scc");
}
template <int MIMM>
__device__ __forceinline__ void dma16_buf(i32x4 rsrc, unsigned soff, unsigned mbase, unsigned v0) {
  asm volatile("s_add_i32 m0, %1, %4\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %0, %2 offen lds"
               :: "s"(rsrc), "s"(mbase), "s"(soff), "v"(v0), "n"(MIMM) : "memory", "m0", "scc");
}
// (a.x - b.x, a.y - b.y) / (a.x + b.x, a.y + b.y)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// columns of B^T d B from a transformed row (lo = t0,t1; hi = t2,t3):
// (t0 - t2, t1 + t2) and (t2 - t1, t1 - t3)
__device__ __forceinline__ f32x2 pk_col01(f32x2 lo, f32x2 hi) {
  f32x2 r;
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]"
               : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ f32x2 pk_col23(f32x2 lo, f32x2 hi) {
  f32x2 r;
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]"
               : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// Persistent workgroups: each of the (at most 256) workgroups walks a contiguous range of work
// items, and the K-chunk stream runs on ACROSS item boundaries, so the staging of the next
// item's first chunks overlaps the tail and the epilogue of the current one.
// TRACE: debug build that stamps cycle counters of workgroup 0 (tools/wino_trace.py); the
// production instance carries none of it.
template <int TR, bool TRACE = false>
__global__ __launch_bounds__(256) void wino_conv_kernel(WinoArgs a) {
  using C = WinoCfg<TR>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert((NBUF - 1) * C::ND <= 63, "vmcnt is a 6-bit counter");
  static_assert(C::NI <= 12, "patch DMAs are issued as at most 3 groups of 4");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 1;   // channel half of the workgroup's 64
  const int gw = wave >> 1;  // tile group of the workgroup's 2
  const int HWi = a.H * a.W;
  const int nchunk = a.Cin / WCK;

  const int lw = xcd_remap(blockIdx.x, gridDim.x);
  const int i0 = (int)((long long)lw * a.nitems / (int)gridDim.x);
  const int i1 = (int)((long long)(lw + 1) * a.nitems / (int)gridDim.x);
  if (i0 >= i1) return;
  if (TRACE && a.trace != nullptr && blockIdx.x == 0 && tid == 0) {
    a.trace[62] = clock64();
    a.trace[63] = wall_clock64();
  }
  const int S = (i1 - i0) * nchunk;  // chunks in this workgroup's stream

  // ---- staging state: runs NBUF-1 chunks ahead of the compute state
  const i32x4 xrs = make_rsrc(reinterpret_cast<const char*>(a.x) - XBIAS,
                              (unsigned)a.B * a.Cin * HWi * 4u + XBIAS);
  const i32x4 urs = make_rsrc(a.up, (unsigned)a.ncot * WBM * a.Cin * 64u);
  // item-invariant decomposition of this thread's patch elements: grp<<24 | cil<<16 | r<<8 | c
  // (kept in LDS behind the ring, not in registers: it is needed once per item, and a value the
  // register allocator spills to scratch comes back through vmcnt, behind the DMAs in flight)
  int* epack = reinterpret_cast<int*>(lds + NBUF * C::BUF) + tid;
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = i * 256 + tid;
    const int grp = e / (WCK * C::CHS);
    const int rem = e - grp * (WCK * C::CHS);
    const int cil = rem / C::CHS;
    const int rem2 = rem - cil * C::CHS;
    const int r = rem2 / C::PC;
    epack[i * 256] = e < C::NE ? (grp << 24) | (cil << 16) | (r << 8) | (rem2 - r * C::PC) : -1;
  }
  unsigned voff[C::NI];  // byte offsets of the patch elements into the x descriptor; padding and
                         // unused slots point out of range and arrive as zeros
  unsigned usoff = 0;    // byte offset of the staging item's weight slabs
  auto set_voff = [&](int item) {
    ItemPos D;
    item_init(D, item, a);
    const int b0 = D.g[0].b, b1 = D.g[1].b;
    const int h00 = 2 * TR * D.g[0].thg - 1, h01 = 2 * TR * D.g[1].thg - 1;
    const int w00 = 2 * C::TC * D.g[0].twg - 1, w01 = 2 * C::TC * D.g[1].twg - 1;
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
      const int ep = epack[i * 256];
      const int grp = (ep >> 24) & 1, cil = (ep >> 16) & 255, r = (ep >> 8) & 255, c = ep & 255;
      const int bb = grp ? b1 : b0;
      const int hi = (grp ? h01 : h00) + r, wi = (grp ? w01 : w00) + c;
      const bool ok = ep >= 0 && bb < a.B && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      const unsigned off = (unsigned)(((bb * a.Cin + cil) * HWi + hi * a.W + wi) * 4) + XBIAS;
      voff[i] = (ok ? off : OOB) - (unsigned)(i & 3) * 1024u;  // minus the instruction offset
    }
    usoff = __builtin_amdgcn_readfirstlane((unsigned)D.cot * (unsigned)nchunk * (USLAB * 4u));
  };
  set_voff(i0);
  // Everything a DMA needs besides its per-lane offset is wave-uniform and lives in SGPRs that
  // move once per chunk: the x / weight-slab soffsets and the wave's two LDS bases in the target
  // buffer.  Behind the end of the stream the staging state stays on the last chunk (it is staged
  // again into free buffers), so issue counts - and with them the vmcnt waits - never vary.
  int dItem = i0, dChunk = 0, dBuf = 0, dLeft = S;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  const unsigned mP0 = lds0 + 4u * USLAB + wave * 256u, mU0 = lds0 + wave * 1024u;
  unsigned xso = 0, uso = usoff, mP = mP0, mU = mU0;
  unsigned uvoff[4];
  auto dma_unit = [&](int u) {
    if (u < C::NI) {
      const unsigned vo = voff[u < C::NI ? u : 0];
#define AIR_PATCH_DMA(G_, K_) \
  if (u == 4 * G_ + K_) dma4_buf<G_ * 4096, K_ * 1024>(xrs, xso, mP, vo);
      AIR_PATCH_DMA(0, 0) AIR_PATCH_DMA(0, 1) AIR_PATCH_DMA(0, 2) AIR_PATCH_DMA(0, 3)
      AIR_PATCH_DMA(1, 0) AIR_PATCH_DMA(1, 1) AIR_PATCH_DMA(1, 2) AIR_PATCH_DMA(1, 3)
      AIR_PATCH_DMA(2, 0) AIR_PATCH_DMA(2, 1) AIR_PATCH_DMA(2, 2) AIR_PATCH_DMA(2, 3)
#undef AIR_PATCH_DMA
    } else {
      const int q = u - C::NI;
      if (q == 0) dma16_buf<0>(urs, uso, mU, uvoff[0]);
      if (q == 1) dma16_buf<4096>(urs, uso, mU, uvoff[1]);
      if (q == 2) dma16_buf<8192>(urs, uso, mU, uvoff[2]);
      if (q == 3) dma16_buf<12288>(urs, uso, mU, uvoff[3]);
    }
  };
  auto dma_advance = [&]() {
    dBuf = dBuf + 1 == NBUF ? 0 : dBuf + 1;
    if (dLeft > 1) {
      --dLeft;
      if (++dChunk == nchunk) {
        dChunk = 0;
        ++dItem;
        set_voff(dItem);
      }
    }
    xso = __builtin_amdgcn_readfirstlane((unsigned)dChunk * (unsigned)(WCK * HWi * 4));
    uso = __builtin_amdgcn_readfirstlane(usoff + (unsigned)dChunk * (USLAB * 4u));
    mP = __builtin_amdgcn_readfirstlane(mP0 + (unsigned)dBuf * (C::BUF * 4u));
    mU = __builtin_amdgcn_readfirstlane(mU0 + (unsigned)dBuf * (C::BUF * 4u));
  };

  // ---- compute state
  // Per-lane constants are RE-DERIVED from the thread id at the top of every item (behind an
  // opaque asm so the derivation is not hoisted): kept live across the epilogue they get
  // spilled to scratch, and their reload would put vmcnt waits - which also drain the DMAs in
  // flight - into the chunk loop.
  int pb_lane, ub_lane;  // LDS offsets (floats, within a buffer) of the lane's patch / weights
  auto lane_consts = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int l31_ = t & 31, half_ = (t >> 5) & 1;
    const int tr_ = l31_ / C::TC, tc_ = l31_ % C::TC;  // this lane's tile within its group
    pb_lane = USLAB + (gw * WCK + half_) * C::CHS + (2 * tr_) * C::PC + 2 * tc_;
    ub_lane = half_ * (16 * WBM) + (cw * 32 + l31_) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) uvoff[q] = t * 16u + q * 4096u;
  };
  lane_consts();

  f32x16 acc[16];

  f32x4 U0[4], U1[4];     // transformed weights of step 0 / 1
  f32x2 V0[8], V1[8];     // transformed input of step 0 / 1: V[2i] = v[4i], v[4i+1]; V[2i+1] = v[4i+2], v[4i+3]
  f32x2 d[8];             // raw patch rows: d[2r], d[2r+1] = columns 0-1, 2-3 of row r
  f32x2 tl[4], th[4];     // row-transformed: columns 0-1 / 2-3

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
  // B^T d B as 16 packed adds in 8 pieces (the asm statements pin each piece to its MFMA slot)
  auto row_piece = [&](int k) {  // transformed row k, both column pairs
    if (k == 0) { tl[0] = pk_sub(d[0], d[4]); th[0] = pk_sub(d[1], d[5]); }
    if (k == 1) { tl[1] = pk_add(d[2], d[4]); th[1] = pk_add(d[3], d[5]); }
    if (k == 2) { tl[2] = pk_sub(d[4], d[2]); th[2] = pk_sub(d[5], d[3]); }
    if (k == 3) { tl[3] = pk_sub(d[2], d[6]); th[3] = pk_sub(d[3], d[7]); }
  };
  auto col_piece = [&](int i, f32x2(&V)[8]) {
    V[2 * i] = pk_col01(tl[i], th[i]);
    V[2 * i + 1] = pk_col23(tl[i], th[i]);
  };

  // Y = A^T M A per lane for the item at (eb, ethg, etwg, ecot); D row i = (r&3) + 8*(r>>2) +
  // 4*half -> channel
  auto epilogue = [&](int eb, int ethg, int etwg, int ecot) {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int l31 = t & 31, half = (t >> 5) & 1;
    const int tr = l31 / C::TC, tc = l31 % C::TC;
    const int th_ = TR * ethg + tr, tw_ = C::TC * etwg + tc;
    // (accumulators are read in uniform control flow; only the stores are predicated.  Read
    // under a divergent branch, all 256 of them get copied to VGPRs up front and spill.)
    const bool valid = eb < a.B && th_ < a.TH && tw_ < a.TW;
    const int ho = 2 * th_, wo = 2 * tw_;
    const bool w1 = wo + 1 < a.W, h1 = ho + 1 < a.H;
    // per-lane part of the address once; the row part is wave-uniform (kept out of VGPRs: the
    // compiler hoists whatever is item-invariant here, and 16 hoisted 64-bit lane offsets spill)
    const size_t olane = (size_t)(4 * half) * HWi + (size_t)ho * a.W + wo;
    const int co0 = ecot * WBM + cw * 32;
    float* __restrict__ yb = a.y + ((size_t)eb * a.Cout + co0) * HWi + olane;
    const float* __restrict__ rb =
        a.residual != nullptr ? a.residual + ((size_t)eb * a.Cout + co0) * HWi + olane : nullptr;
    // Output transform on whole accumulators (16 rows at a time), so every AGPR is read exactly
    // once: s0 = m[j] + m[4+j] + m[8+j], s1 = m[4+j] - m[8+j] - m[12+j] per column j, then
    // y[.][0] = s[0] + s[1] + s[2], y[.][1] = s[1] - s[2] - s[3].  (Row-wise extraction makes the
    // compiler copy whole 16-register tuples per element and spill.)
    // Two passes (output row 0, then row 1) keep the live set at 3 x 16 registers next to the
    // prefetched operands of the next item, so nothing spills around the epilogue.
#pragma unroll
    for (int half_row = 0; half_row < 2; ++half_row) {
      f32x16 ya, yb2;  // y[half_row][0], y[half_row][1] for the 16 D rows
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x16 sj = half_row == 0 ? acc[j] + acc[4 + j] + acc[8 + j]
                                        : acc[4 + j] - acc[8 + j] - acc[12 + j];
        if (j == 0) ya = sj;
        if (j == 1) { ya += sj; yb2 = sj; }
        if (j == 2) { ya += sj; yb2 -= sj; }
        if (j == 3) yb2 -= sj;
        __builtin_amdgcn_sched_barrier(0);
      }
      // Stores.  The common case is branch-free per row: the channel test is wave-uniform
      // (Cout % 32 == 0), an even W makes every valid lane own both columns, so one exec region
      // (valid lanes) holds 16 x [8-byte residual load] + [8-byte store].
      if (co0 + 32 <= a.Cout && valid && (half_row == 0 || h1)) {
        const size_t hoff = half_row ? (size_t)a.W : 0;
        if ((a.W & 1) == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const size_t orow =
                (size_t)__builtin_amdgcn_readfirstlane(((r & 3) + 8 * (r >> 2)) * HWi) + hoff;
            f32x2a4 v;
            v[0] = ya[r];
            v[1] = yb2[r];
            if (rb != nullptr) {
              const f32x2a4 rv = *reinterpret_cast<const f32x2a4*>(rb + orow);
              v[0] += rv[0];
              v[1] += rv[1];
            }
            *reinterpret_cast<f32x2a4*>(yb + orow) = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const size_t orow =
                (size_t)__builtin_amdgcn_readfirstlane(((r & 3) + 8 * (r >> 2)) * HWi) + hoff;
            float v0 = ya[r], v1 = yb2[r];
            if (rb != nullptr) {
              const float* __restrict__ rp = rb + orow;
              v0 += rp[0];
              if (w1) v1 += rp[1];
            }
            float* __restrict__ yp = yb + orow;
            yp[0] = v0;
            if (w1) yp[1] = v1;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: fill the ring, wait for the first chunk, transform its first step
#pragma nounroll
  for (int cb = 0; cb < NBUF; ++cb) {
#pragma unroll
    for (int u = 0; u < C::ND; ++u) dma_unit(u);
    dma_advance();
  }
  dma_wait_n<(NBUF - 1) * C::ND>();  // chunk 0 has landed; the others may still fly
  __syncthreads();
  ldU(lds, 0, U0);
  ldD(lds, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) row_piece(k);
#pragma unroll
  for (int i = 0; i < 4; ++i) col_piece(i, V0);

  int ntr = 0;
  auto stamp = [&]() {
    if (TRACE && a.trace != nullptr && blockIdx.x == 0 && tid == 0 && ntr < 60) {
      a.trace[ntr] = clock64();
      a.trace[64 + ntr] = wall_clock64();
      ++ntr;
    }
  };
  stamp();
  int cur = 0;
  // one K chunk = 2 k-steps.  FIRST: the item's first chunk, whose step 0 starts the
  // accumulators from a zero C operand (no separate zeroing pass over 256 AGPRs).
  constexpr int NU1 = 8;  // DMAs issued in step 1 (slots 0, 2, .. 14); the rest in the next step 0
  static_assert(C::ND - NU1 <= 7, "step 0 has 7 even slots from 2 on");
  long long tS0 = 0, tWait = 0, tBar = 0, tS1 = 0;
  const bool tracing = TRACE && a.trace != nullptr && blockIdx.x == 0;
  // PEND: a chunk's staging began behind the previous barrier (false only for the very first
  // chunk of the stream, whose predecessors the prologue staged whole)
  auto chunk_body = [&](auto first_tag, auto pend_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr bool PEND = decltype(pend_tag)::value;
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (tracing) c0 = clock64();
    const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
    const float* __restrict__ bcur = lds + cur * C::BUF;
    const float* __restrict__ bnxt = lds + nxt * C::BUF;
    // ---- step 0 of this chunk; fetch + transform step 1 in between
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j == 0) ldU(bcur, 1, U1);
      if (j == 1) ldD(bcur, 1);
      if (j >= 4 && j < 8) row_piece(j - 4);
      if (j >= 8 && j < 12) col_piece(j - 8, V1);
      // second half of the DMAs of the chunk whose staging began behind the last barrier
      if (PEND && j >= 2 && (j & 1) == 0 && NU1 + (j - 2) / 2 < C::ND) dma_unit(NU1 + (j - 2) / 2);
      if (PEND && j == 15) dma_advance();
      __builtin_amdgcn_sched_barrier(0);
      if (FIRST)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(U0[j >> 2][j & 3], V0[j >> 1][j & 1],
                                                     (f32x16){0}, 0, 0, 0);
      else
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(U0[j >> 2][j & 3], V0[j >> 1][j & 1], acc[j],
                                                     0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- step 1.  Stream chunk p+1 must have landed (DMAs complete in order; chunks up to
    // p+NBUF-2 are in flight).  Behind an epilogue its stores share the counter: drain.
    if (tracing) c1 = clock64();
    // (vmcnt also counts the epilogue's stores, but loads retire in order: "at most N operations
    // outstanding" still means at most the N youngest LOADS are, and chunk p+1 has >= N younger)
    dma_wait_n<(NBUF - 2) * C::ND>();
    if (tracing) c2 = clock64();
    __syncthreads();
    if (tracing) c3 = clock64();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // this chunk's buffer is free (every wave read its last operands before the barrier):
      // restage it with stream chunk p + NBUF, first half of the DMAs here
      if ((j & 1) == 0) dma_unit(j / 2);
      // (behind the last chunk of the stream these read stale LDS: unused, but branch-free)
      if (j == 5) ldU(bnxt, 0, U0);
      if (j == 7) ldD(bnxt, 0);
      if (j >= 8 && j < 12) row_piece(j - 8);
      if (j >= 12) col_piece(j - 12, V0);
      __builtin_amdgcn_sched_barrier(0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(U1[j >> 2][j & 3], V1[j >> 1][j & 1], acc[j],
                                                   0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tracing) {
      const long long c4 = clock64();
      tS0 += c1 - c0; tWait += c2 - c1; tBar += c3 - c2; tS1 += c4 - c3;
    }
    cur = nxt;
  };
  for (int item = i0; item < i1; ++item) {
    if (item != i0) lane_consts();
    if (item == i0)
      chunk_body(std::true_type{}, std::false_type{});
    else
      chunk_body(std::true_type{}, std::true_type{});
    for (int chunk = 1; chunk < nchunk; ++chunk) chunk_body(std::false_type{}, std::true_type{});
    stamp();
    {
      ItemPos P;
      item_init(P, item, a);
      epilogue(gw ? P.g[1].b : P.g[0].b, gw ? P.g[1].thg : P.g[0].thg,
               gw ? P.g[1].twg : P.g[0].twg, P.cot);
    }
    // Compiler-visible vmcnt(0): whatever it spilled around the epilogue has come back, so it
    // puts no vmcnt waits (which would also drain the DMAs in flight) into the chunk loop.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    stamp();
  }
  if (tracing && lane == 0 && wave < 4) {
    a.trace[96 + wave * 4 + 0] = tS0; a.trace[96 + wave * 4 + 1] = tWait;
    a.trace[96 + wave * 4 + 2] = tBar; a.trace[96 + wave * 4 + 3] = tS1;
  }
}

// ---------------------------------------------------------------------------------------
// Weight gradient of the same 3x3 / stride 1 / pad 1 convolution as Winograd F(3x3,2x2):
//   dW[co][ci] (3x3) = sum over 2x2 output tiles of  A'^T [ (G' g G'^T) .* (B^T d B) ] A'
// with g the 2x2 tile of dy[co], d the 4x4 patch of x[ci] around it (the same B^T d B as the
// forward transform).  16 multiplies per (co, ci, tile) instead of 36: 16 GEMMs
// M_xn[co][ci] = sum_tiles Gd_xn[co][tile] * V_xn[ci][tile] whose K dimension is the tile index.
//   * a wave owns 32 co x 32 ci x 16 accumulators; a workgroup 64 x 64; the tile stream is split
//     over workgroups (K-split) and the partial 3x3 results are reduced by reduce_partials_kernel.
//   * an MFMA k-step is 2 neighbouring tiles: a lane transforms the patch of ITS input channel
//     (16 packed adds) and the dy tile of ITS output channel (6 packed adds) for one tile.
//   * a stage = 16 tiles of one tile row: x rows of 64 channels (one dwordx4 DMA per channel: the
//     24 idle lanes write zeros that the next channel's DMA, issued by the same wave, overwrites;
//     a pad takes the last one's) and dy rows of 64 channels (one dword DMA per channel), channel
//     pitches chosen so that 32 lanes reading 32 channels hit 32 distinct LDS bank pairs.
//   * zero padding = the buffer descriptor's out-of-range rule per 16-byte lane; a lane chunk that
//     straddles the right image edge brings the next row's first pixels instead of zeros: the
//     two cells a valid tile can see (columns W, W+1) are zeroed in LDS by the reading wave.
#ifndef W2_DMA_KS
#define W2_DMA_KS 4  // k-steps over which the next stage's 32 DMAs are issued
#endif
#ifndef W2_TX_AT
#define W2_TX_AT 6   // MFMA slots after which the next k-step's patch / dy transforms are placed
#define W2_TG_AT 10
#endif
constexpr int WG_SEG = 16;                    // tiles per stage
constexpr int WG_XC = 40;                     // staged x columns per row: image cols 32 seg - 4 ...
constexpr int WG_XP = 4 * WG_XC + 2;          // floats per channel (odd half: conflict-free b64 / b32x2)
constexpr int WG_XWAVE = 16 * WG_XP + 96;     // a wave's 16 channels + pad for the last DMA's idle lanes
constexpr int WG_XF = 4 * WG_XWAVE;
constexpr int WG_DP = 66;                     // dy floats per channel: 2 rows x 32 columns + 2
constexpr int WG_DF = 64 * WG_DP;
constexpr int WG_BUF = WG_XF + WG_DF;         // floats per stage buffer (59.9 KB)

struct WinoWgArgs {
  const float* x;    // (B, Cin, H, W)
  const float* dy;   // (B, Cout, H, W)
  float* partial;    // [nsplit][9][Cout][Cin]
  int B, Cin, H, W, Cout;
  int TH, NTS;       // tile rows, tile-row segments per row
  int nseg;          // B * TH * NTS
  int ncob, ncib, nsplit;
};

// (x + y, x - y) of one register pair
__device__ __forceinline__ f32x2 pk_sumdiff(f32x2 q) {
  f32x2 r;
  asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]"
               : "=v"(r) : "v"(q));
  return r;
}
template <int MIMM>
__device__ __forceinline__ void dma16_at(i32x4 rsrc, unsigned soff, unsigned mbase, unsigned v0) {
  asm volatile("s_add_i32 m0, %1, %4\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %0, %2 offen lds"
               :: "s"(rsrc), "s"(mbase), "s"(soff), "v"(v0), "n"(MIMM) : "memory", "m0", "scc");
}
template <int MIMM>
__device__ __forceinline__ void dma4_at(i32x4 rsrc, unsigned soff, unsigned mbase, unsigned v0) {
  asm volatile("s_add_i32 m0, %1, %4\n\ts_nop 0\n\t"
               "buffer_load_dword %3, %0, %2 offen lds"
               :: "s"(rsrc), "s"(mbase), "s"(soff), "v"(v0), "n"(MIMM) : "memory", "m0", "scc");
}

__global__ __launch_bounds__(256) void wino_wgrad_kernel(WinoWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = wave & 1;   // co half of the workgroup's 64
  const int ch = wave >> 1;  // ci half
  const int HWi = a.H * a.W;

  int lb = blockIdx.x;
  const int split = lb % a.nsplit;
  lb /= a.nsplit;
  const int cib = lb % a.ncib;
  const int cob = lb / a.ncib;
  const int per = (a.nseg + a.nsplit - 1) / a.nsplit;
  const int s0 = split * per, s1 = min(a.nseg, s0 + per);

  const i32x4 xrs = make_rsrc(a.x, (unsigned)a.B * a.Cin * HWi * 4u);
  const i32x4 drs = make_rsrc(a.dy, (unsigned)a.B * a.Cout * HWi * 4u);
  const unsigned hw4 = (unsigned)HWi * 4u;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  // LDS bases (bytes) of this wave's DMA destinations inside a buffer
  const unsigned mX0 = lds0 + wave * (WG_XWAVE * 4u);
  const unsigned mD0 = lds0 + WG_XF * 4u + wave * (16 * WG_DP * 4u);

  // stage geometry -> per-lane source offsets (one VGPR each) and the wave's channel bases
  int edge_col = -1;  // staged column of image column W if a valid tile of the stage sees it
  auto set_stage = [&](int s, unsigned& vx, unsigned& vd, unsigned& xs, unsigned& dsf, int& ecol) {
    const int seg = s % a.NTS;
    const int r = s / a.NTS;
    const int th = r % a.TH;
    const int b = r / a.TH;
    {
      const int row = 2 * th - 1 + lane / 10, col0 = 32 * seg - 4 + 4 * (lane % 10);
      const bool ok = lane < 40 && row >= 0 && row < a.H && col0 >= 0 && col0 < a.W;
      vx = ok ? (unsigned)((row * a.W + col0) * 4) : OOB;
    }
    {
      const int row = 2 * th + half, col = 32 * seg + l31;
      vd = (row < a.H && col < a.W) ? (unsigned)((row * a.W + col) * 4) : OOB;
    }
    xs = __builtin_amdgcn_readfirstlane((unsigned)((b * a.Cin + cib * 64 + wave * 16) * HWi) * 4u);
    dsf = __builtin_amdgcn_readfirstlane((unsigned)((b * a.Cout + cob * 64 + wave * 16) * HWi) * 4u);
    const int wc = a.W - (32 * seg - 4);  // staged column of image column W
    ecol = (wc >= 4 && wc < WG_XC) ? wc : -1;
  };

  // DMA unit u (0 .. 31) of the stage being staged into buffer `buf`: 16 x channels, 16 dy channels
  unsigned n_vox = OOB, n_vod = OOB, n_xs = 0, n_ds = 0, mXn = 0, mDn = 0;
  int n_edge = -1;
  auto dma_unit = [&](int u) {
#define AIR_XU(K_) if (u == K_) dma16_at<K_ * WG_XP * 4>(xrs, n_xs + K_ * hw4, mXn, n_vox);
#define AIR_DU(K_) if (u == 16 + K_) dma4_at<K_ * WG_DP * 4>(drs, n_ds + K_ * hw4, mDn, n_vod);
    AIR_XU(0) AIR_XU(1) AIR_XU(2) AIR_XU(3) AIR_XU(4) AIR_XU(5) AIR_XU(6) AIR_XU(7)
    AIR_XU(8) AIR_XU(9) AIR_XU(10) AIR_XU(11) AIR_XU(12) AIR_XU(13) AIR_XU(14) AIR_XU(15)
    AIR_DU(0) AIR_DU(1) AIR_DU(2) AIR_DU(3) AIR_DU(4) AIR_DU(5) AIR_DU(6) AIR_DU(7)
    AIR_DU(8) AIR_DU(9) AIR_DU(10) AIR_DU(11) AIR_DU(12) AIR_DU(13) AIR_DU(14) AIR_DU(15)
#undef AIR_XU
#undef AIR_DU
  };

  // per-lane LDS read offsets (floats within a buffer)
  const int cil = ch * 32 + l31, col = mt * 32 + l31;
  const int xb_lane = (cil >> 4) * WG_XWAVE + (cil & 15) * WG_XP + 2 * half + 3;
  const int db_lane = WG_XF + col * WG_DP + 2 * half;

  f32x16 acc[16];
  f32x2 d[8], g[2];      // raw: patch rows (cols 0-1, 2-3), dy tile rows
  // operands, two sets (k-step ks multiplies set ks & 1 while the transform of k-step ks + 1 fills the other between
  // its MFMAs): V[2i], V[2i+1]; Gd row i = Q[i].x, Sd[i].x, Sd[i].y, Q[i].y
  f32x2 V[2][8], Q[2][4], Sd[2][4];
  f32x2 tl[4], th2[4];

  auto ld = [&](const float* __restrict__ bufp, int ks) {
    const float* __restrict__ px = bufp + xb_lane + 4 * ks;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[2 * r][0] = px[r * WG_XC];
      d[2 * r][1] = px[r * WG_XC + 1];
      d[2 * r + 1][0] = px[r * WG_XC + 2];
      d[2 * r + 1][1] = px[r * WG_XC + 3];
    }
    const float* __restrict__ pd = bufp + db_lane + 4 * ks;
    g[0] = *reinterpret_cast<const f32x2*>(pd);
    g[1] = *reinterpret_cast<const f32x2*>(pd + 32);
  };
  auto transform_x = [&](int set) {  // patch -> V[set]
    tl[0] = pk_sub(d[0], d[4]); th2[0] = pk_sub(d[1], d[5]);
    tl[1] = pk_add(d[2], d[4]); th2[1] = pk_add(d[3], d[5]);
    tl[2] = pk_sub(d[4], d[2]); th2[2] = pk_sub(d[5], d[3]);
    tl[3] = pk_sub(d[2], d[6]); th2[3] = pk_sub(d[3], d[7]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      V[set][2 * i] = pk_col01(tl[i], th2[i]);
      V[set][2 * i + 1] = pk_col23(tl[i], th2[i]);
    }
  };
  auto transform_g = [&](int set) {  // dy tile -> Q[set], Sd[set]
    // G' g G'^T without its 1/2 factors (they are applied in the output transform)
    Q[set][0] = g[0];
    Q[set][1] = pk_add(g[0], g[1]);
    Q[set][2] = pk_sub(g[0], g[1]);
    Q[set][3] = g[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) Sd[set][i] = pk_sumdiff(Q[set][i]);
  };
  auto opA = [&](int set, int i, int j) -> float {
    return j == 0 ? Q[set][i][0] : (j == 1 ? Sd[set][i][0] : (j == 2 ? Sd[set][i][1] : Q[set][i][1]));
  };
  // zero the cells (columns W, W+1 of the 4 patch rows of this lane's channel) that a chunk
  // straddling the right image edge filled with the next row's pixels
  auto fix_edge = [&](float* bufp, int ecol) {
    float* px = bufp + (cil >> 4) * WG_XWAVE + (cil & 15) * WG_XP + ecol;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      px[r * WG_XC] = 0.0f;
      if (ecol + 1 < WG_XC) px[r * WG_XC + 1] = 0.0f;
    }
  };

  if (s0 < s1) {
    // prologue: stage s0 whole, wait, fix, first operands
    set_stage(s0, n_vox, n_vod, n_xs, n_ds, n_edge);
    mXn = mX0; mDn = mD0;
#pragma unroll
    for (int u = 0; u < 32; ++u) dma_unit(u);
    edge_col = n_edge;
    dma_wait();
    __syncthreads();
  }
  // HALF: the stage lies in the last tile row of an image with an odd row count - the dy tiles' second row is beyond
  // the image (zeros), so row 3 of G' g G'^T is zero and the four products M[3][*] with it: 12 MFMAs per k-step
  // instead of 16 (F(3x3, 1x2) in effect).  The ResNet's 9 / 5 / 3-row maps have 1 of 5 / 3 / 2 tile rows like that:
  // 5 / 8 / 12.5 % of the launch's MFMAs (round 4).
  // (a wave-uniform branch around those four MFMAs, not a second instantiation of the stage: the 256 accumulator
  // registers + 256 VGPRs are full, and a duplicated body tripled the spills - 0.31 -> 0.42 ms on layer1, which has
  // no such row at all)
  auto stage_body = [&](auto first_tag, bool hs, int s, int cur) {
    constexpr bool FIRST = decltype(first_tag)::value;
    float* bcur = lds + cur * WG_BUF;
    if (edge_col >= 0) fix_edge(bcur, edge_col);
    const bool more = s + 1 < s1;
    if (more) {
      set_stage(s + 1, n_vox, n_vod, n_xs, n_ds, n_edge);
      const unsigned nb = (unsigned)(cur ^ 1) * (WG_BUF * 4u);
      mXn = __builtin_amdgcn_readfirstlane(mX0 + nb);
      mDn = __builtin_amdgcn_readfirstlane(mD0 + nb);
    }
    ld(bcur, 0);
    transform_x(0);  // the stage's first k-step: exposed; the other seven are transformed under the MFMAs before them
    transform_g(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cs = ks & 1, ns = cs ^ 1;
      if (ks < 7) ld(bcur, ks + 1);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        // the next stage's 32 DMAs: one per second MFMA slot of the first four k-steps
#if W2_DMA_KS == 2
        if (ks < 2 && more) dma_unit(ks * 16 + j);
#else
        if (ks < 4 && (j & 1) == 0 && more) dma_unit(ks * 8 + j / 2);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (j >= 12 && hs) {
          if (FIRST && ks == 0) acc[j] = (f32x16){0};
        } else if (FIRST && ks == 0)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA(cs, j >> 2, j & 3), V[cs][j >> 1][j & 1],
                                                       (f32x16){0}, 0, 0, 0);
        else
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(opA(cs, j >> 2, j & 3), V[cs][j >> 1][j & 1],
                                                       acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // the next k-step's operands, in the shadow of this one's MFMAs (its LDS reads were issued 6 MFMAs ago)
        if (ks < 7 && j == W2_TX_AT) transform_x(ns);
        if (ks < 7 && j == W2_TG_AT) transform_g(ns);
      }
    }
    edge_col = n_edge;
    dma_wait();
    __syncthreads();
  };
#ifndef W2_HALF_ROWS
#define W2_HALF_ROWS 1  // A/B: 0 = every stage issues all 16 products (rounds 1 - 3)
#endif
  auto is_half = [&](int s) { return W2_HALF_ROWS && 2 * ((s / a.NTS) % a.TH) + 1 >= a.H; };
  if (s0 < s1) {
    stage_body(std::true_type{}, is_half(s0), s0, 0);
    int cur = 1;
    for (int s = s0 + 1; s < s1; ++s, cur ^= 1) stage_body(std::false_type{}, is_half(s), s, cur);
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = (f32x16){0};
  }

  // Output transform R = A'^T (s s^T .* M) A', s = (1, 1/2, 1/2, 1), A'^T = [[1,1,1,0],[0,1,-1,0],
  // [0,1,1,-1]], on whole accumulators; partial[split][tap][co][ci]: lanes = consecutive ci.
  float* __restrict__ out = a.partial + (size_t)split * 9 * a.Cout * a.Cin;
  const int ci = cib * 64 + ch * 32 + l31;
#pragma unroll
  for (int ra = 0; ra < 3; ++ra) {
    f32x16 T[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ra == 0) T[j] = acc[j] + 0.5f * (acc[4 + j] + acc[8 + j]);
      if (ra == 1) T[j] = 0.5f * (acc[4 + j] - acc[8 + j]);
      if (ra == 2) T[j] = 0.5f * (acc[4 + j] + acc[8 + j]) - acc[12 + j];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) {
      f32x16 R;
      if (rb == 0) R = T[0] + 0.5f * (T[1] + T[2]);
      if (rb == 1) R = 0.5f * (T[1] - T[2]);
      if (rb == 2) R = 0.5f * (T[1] + T[2]) - T[3];
      const int tap = ra * 3 + rb;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int co = cob * 64 + mt * 32 + i;
        out[((size_t)tap * a.Cout + co) * a.Cin + ci] = R[r];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

int grid_for(size_t n) {
  size_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

static long long* g_wino_trace = nullptr;
extern "C" void air_dbg_wino_trace(long long* p) { g_wino_trace = p; }

bool air_wino_ok(int B, int Kc, int H, int W, int M) {
  if (air_opt(AIR_OPT_NO_WINOGRAD) & 1) return false;
  if (M < 32 || M % 32 != 0 || Kc < WCK || Kc % WCK != 0) return false;
  // buffer-descriptor staging: byte offsets stay below the out-of-range marker (2 GiB)
  const double ein = (double)B * Kc * H * W;
  return ein * 4.0 + 8192.0 < 2147483648.0 && (double)air_wino_packed_elems(M, Kc) * 4.0 < 4294967296.0 &&
         H >= 1 && W >= 2;
}

size_t air_wino_packed_elems(int M, int Kc) {
  return (size_t)((M + WBM - 1) / WBM * WBM) * Kc * 16;
}

bool air_wino_wgrad_ok(int B, int Cin, int H, int W, int Cout) {
  if (air_opt(AIR_OPT_NO_WINOGRAD) & 2) return false;
  if (Cin % 64 != 0 || Cout % 64 != 0 || W < 2) return false;
  const double ein = (double)B * Cin * H * W, eout = (double)B * Cout * H * W;
  return ein * 4.0 < 2147483648.0 && eout * 4.0 < 2147483648.0;
}

int air_wino_wgrad_nsplit(int B, int Cin, int H, int W, int Cout) {
  const int nseg = B * ((H + 1) / 2) * (((W + 1) / 2 + WG_SEG - 1) / WG_SEG);
  const int total = air_opt(AIR_OPT_WINO_WGRAD_WGS);
  int n = total / ((Cin / 64) * (Cout / 64));
  if (n < 1) n = 1;
  if (n > nseg) n = nseg;
  return n;
}

int air_wino_wgrad_partials(const float* x, const float* dy, float* partial, int B, int Cin, int H,
                            int W, int Cout, double flops, hipStream_t st) {
  WinoWgArgs a;
  a.x = x; a.dy = dy; a.partial = partial;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
  a.TH = (H + 1) / 2;
  a.NTS = ((W + 1) / 2 + WG_SEG - 1) / WG_SEG;
  a.nseg = B * a.TH * a.NTS;
  a.ncob = Cout / 64; a.ncib = Cin / 64;
  a.nsplit = air_wino_wgrad_nsplit(B, Cin, H, W, Cout);
  const size_t ldsb = 2 * WG_BUF * sizeof(float);
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) == hipSuccess;
  if (!attr_ok) return AIR_ELAUNCH;
  AirProfScope ps(AIR_K_CONV_WINO_WG, flops, st);
  hipLaunchKernelGGL(wino_wgrad_kernel, dim3(a.ncob * a.ncib * a.nsplit), dim3(256), ldsb, st, a);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_wino_weights(const float* w, float* up, int M, int Kc, int dgrad, hipStream_t st) {
  hipLaunchKernelGGL(wino_weights_kernel, dim3(grid_for((size_t)((M + WBM - 1) / WBM * WBM) * Kc)),
                     dim3(256), 0, st, w, up, M, Kc, dgrad);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_wino_conv(const float* x, const float* w, float* y, const float* residual, int B, int Kc,
                  int H, int W, int M, int dgrad, float* up, double flops, hipStream_t st) {
  if (w != nullptr) {  // w == nullptr: `up` already holds the transformed weights (air_wino_weights)
    const int rc = air_wino_weights(w, up, M, Kc, dgrad, st);
    if (rc != AIR_OK) return rc;
  }
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
  a.trace = g_wino_trace;
  a.nitems = (a.ngroups + 1) / 2 * a.ncot;
  const int ncu = air_stream_cus(st);
  const int nblk = a.nitems < ncu ? a.nitems : ncu;  // one persistent workgroup per CU the stream can use
  const size_t lds1 = (NBUF * WinoCfg<1>::BUF + WinoCfg<1>::NI * 256) * sizeof(float);
  const size_t lds2 = (NBUF * WinoCfg<2>::BUF + WinoCfg<2>::NI * 256) * sizeof(float);
  static const bool attr_ok = [=] {  // > 64 KB of dynamic LDS needs the opt-in, once per kernel
    const void* ks[4] = {reinterpret_cast<const void*>(wino_conv_kernel<1, false>),
                         reinterpret_cast<const void*>(wino_conv_kernel<1, true>),
                         reinterpret_cast<const void*>(wino_conv_kernel<2, false>),
                         reinterpret_cast<const void*>(wino_conv_kernel<2, true>)};
    bool ok = true;
    for (int i = 0; i < 4; ++i)
      ok = ok && hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(i < 2 ? lds1 : lds2)) == hipSuccess;
    return ok;
  }();
  if (!attr_ok) return AIR_ELAUNCH;
  AirProfScope ps(AIR_K_CONV_WINO, flops, st);
  if (a.trace != nullptr) {
    if (trows == 2)
      hipLaunchKernelGGL((wino_conv_kernel<2, true>), dim3(nblk), dim3(256), lds2, st, a);
    else
      hipLaunchKernelGGL((wino_conv_kernel<1, true>), dim3(nblk), dim3(256), lds1, st, a);
  } else if (trows == 2) {
    hipLaunchKernelGGL((wino_conv_kernel<2, false>), dim3(nblk), dim3(256), lds2, st, a);
  } else {
    hipLaunchKernelGGL((wino_conv_kernel<1, false>), dim3(nblk), dim3(256), lds1, st, a);
  }
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}
