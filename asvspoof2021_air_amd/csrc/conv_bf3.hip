// Stride-2 3x3 convolution forward on the bf16 matrix cores with fp32-EQUIVALENT arithmetic ("bf16 x 3").
//
// Replaces, for the three stride-2 conv1 layers of the ResNet (resnet.py:59: layer2.0 / layer3.0 / layer4.0 .conv1, 3.2 %
// + 3.5 % + 4.2 % of the forward FLOPs), the direct kernel on v_mfma_f32_32x32x2_f32 (conv2d.hip, ~90 TF of a
// 157 TF peak).  Every fp32 operand is split EXACTLY into three bf16 planes, x = hi + mid + lo (8 + 8 + 8 mantissa bits,
// round to nearest even at each step), and the product is taken as the six terms
//     hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: the dropped terms (mid.lo, lo.mid, lo.lo) are below 2^-24 of the
// product.  Emulated against fp64 the six-product form sits BELOW the fp32 fma chain it replaces (1.6e-7 .. 5.7e-7 of
// the output scale for K = 576 .. 4608 against 6.6e-7 .. 1.4e-6: profiles/r05_split_bf16.md), an order of magnitude
// inside the direct kernels' parity bound (tests/_budget.py: conv_rtol 1e-5).  Ceiling 2500 / 6 = 417 TF.
// Domain of the exact split (tests/test_split_bf16_cpu.py): 2^-100 <= |x| <= 3.38e38 and 0.  A value within 2^-8 of FLT_MAX
// rounds its hi plane to infinity (the fp32 kernels would carry it), the low planes of values below ~2^-110 are bf16
// denormals - activations and weights of this network are O(1e-4 .. 1e2).
//
// Mapping (wave64, one 4-wave workgroup per CU):
//   * a wave owns 64 output channels x NT x 32 consecutive output pixels of one output row; the four waves of a
//     workgroup take four neighbouring pixel groups and SHARE the weights;
//   * weights: split into planes once per optimiser step by bf3_pack_kernel (air_conv2d_prepack), laid out in MFMA
//     fragment order; a 54 KB slab = 16 input channels x 9 taps x 64 channels x 3 planes is staged per K chunk by
//     LDS-DMA, double buffered (108 KB);
//   * activations: NO LDS - a lane loads the 8 channels x 3 columns of its pixel straight from global memory (the rows of
//     one channel plane are contiguous: 128-byte segments per 32 lanes), splits them in registers (11 VALU per pair of
//     values: v_cvt_pk_bf16_f32 + shifts + subtractions) and keeps the nine fragments of a kernel row; the loads of
//     the NEXT row are issued right behind the split, under that row's 3 x 24 MFMAs;
//   * one K step = one tap x 16 channels: 6 products x 2 channel tiles x NT pixel tiles MFMAs.
#include <type_traits>

#include "air_common.h"
#include "air_lds_dma.h"
#include "air_options.h"
#include "air_prof.h"
#include <utility>

#include "conv_bf3.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef BF3_EXP
#define BF3_EXP 0
#endif
#ifndef BF3W_ASM_MFMA
#define BF3W_ASM_MFMA 1
#endif
#ifndef BF3_SCHED
#define BF3_SCHED 1
#endif
constexpr int NWAVE = 4;
constexpr int CK = 16;                                  // input channels per K chunk = the MFMA's K
constexpr int TAPS = 9;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, r);
}

// x = hi + mid + lo, each a bf16 (exact: 24 significant bits over three 8-bit pieces)
__device__ __forceinline__ void split3(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = v[2 * q], b = v[2 * q + 1];
    const unsigned h = pack2(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = pack2(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    hi[q] = h;
    mid[q] = m;
    lo[q] = pack2(sa, sb);
  }
}

// fp32 weights (Cout, Cin, 3, 3) -> planes in fragment order:
//   word(((cot * nchunk + chunk) * 9 + tap) * 2 + mt) * 3 + plane) * 64 + lane) = 8 bf16: channel co = cot * 64 + mt * 32
//   + (lane & 31), input channels chunk * 16 + 8 * (lane >> 5) + 0 .. 7
// ntaps = 10: tap 9 = the block's 1x1 / stride 2 shortcut (wsc, (Cout, Cin)), computed by the same launch
__global__ __launch_bounds__(256) void bf3_pack_kernel(const float* __restrict__ w, const float* __restrict__ wsc,
                                                       u32x4* __restrict__ wp, int Cout, int Cin, int ntaps, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = e & 63;
  int blk = e >> 6;
  const int mt = blk & 1;
  blk >>= 1;
  const int tap = blk % ntaps;
  blk /= ntaps;
  const int nchunk = Cin / CK;
  const int chunk = blk % nchunk, cot = blk / nchunk;
  const int co = cot * 64 + mt * 32 + (lane & 31);
  const int ci0 = chunk * CK + 8 * (lane >> 5);
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = tap < TAPS ? w[((size_t)co * Cin + ci0 + i) * TAPS + tap] : wsc[(size_t)co * Cin + ci0 + i];
  u32x4 hi, mid, lo;
  split3(v, hi, mid, lo);
  u32x4* dst = wp + ((size_t)(e >> 6) * 3) * 64 + lane;
  dst[0] = hi;
  dst[64] = mid;
  dst[128] = lo;
}

// (timing-only experiment, BF3_EXP bit 4: what a BatchNorm + ReLU prologue on the lane-loaded activations would cost -
// scale 1 / shift 0 from a table, so that results on non-negative inputs do not change: profiles/r06_bf3_prologue.md)
#if BF3_EXP & 4
__device__ float g_bf3_exp_scale[1024] = {[0 ... 1023] = 1.0f};
__device__ float g_bf3_exp_shift[1024] = {};
#endif

struct Bf3Args {
  const float* x;
  const u32x4* wp;
  float* y;
  float* ysc;   // SC: the shortcut's output (B, Cout, Ho, Wo)
  int B, Cin, H, W, Cout, Ho, Wo;
  int WT;       // pixel groups (NT x 32 pixels) per output row
  int ngroups;  // B * Ho * WT
  int ncot;     // Cout / 64
};

// SC: also the block's 1x1 / stride 2 shortcut on the same input (resnet.py:61-66) - its operand is the centre tap's B
// fragment, which this kernel has split already: 6 x 2 x NT more MFMAs per chunk against a tenth weight tap, a second
// set of accumulators, no second pass over x
template <int NT, bool SC>
__global__ __launch_bounds__(NWAVE * 64, 1) void conv_s2_bf3_kernel(const Bf3Args a) {
  constexpr int NTAPS = SC ? 10 : TAPS;
  constexpr int SLAB_U4 = NTAPS * 2 * 3 * 64;  // 16-byte words per (channel tile, chunk) slab
  constexpr int SLAB_BYTES = SLAB_U4 * 16;     // 55,296 (61,440 with the shortcut)
  __shared__ __attribute__((aligned(16))) u32x4 slab[2 * SLAB_U4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lb % a.ncot;
  const int pxg = lb / a.ncot;
  const int grp = pxg * NWAVE + wave;
  const bool grp_ok = grp < a.ngroups;
  const int grpc = min(grp, a.ngroups - 1);  // (a surplus wave recomputes the last group and stores nothing)
  const int wt = grpc % a.WT;
  const int rowid = grpc / a.WT;
  const int ho = rowid % a.Ho, b = rowid / a.Ho;
  const int wo0 = wt * (32 * NT);
  const int hi0 = 2 * ho - 1;
  const int H = a.H, W = a.W;
  const int HW = H * W;
  const int nchunk = a.Cin / CK;

  // Columns of this lane's pixel wo in tile j: 2 wo - 1, 2 wo, 2 wo + 1.  The lane loads (2 wo, 2 wo + 1) as ONE 8-byte
  // word (16 loads per kernel row instead of 48; 4-byte aligned when W is odd - global memory takes that) and gets
  // 2 wo - 1 from its left neighbour's second element (ds_bpermute); lane 0 of tile j > 0 takes it from lane 31 of tile
  // j - 1, lane 0 of tile 0 from one extra load.  Indices are clamped into the row; what a clamp changed is masked:
  //   * left edge (wo0 == 0): column -1 is padding -> the extra value is 0;
  //   * odd W: the last pixel's 2 wo = W - 1 -> its pair is loaded one column early: c1 = second element, c2 = padding.
  unsigned off2[NT];   // float offset of the pair inside a channel plane row, + this half's 8 channels
  bool fix_last[NT];   // this lane's 2 wo == W - 1 (odd W only)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int c1 = 2 * (wo0 + 32 * j + l31);
    fix_last[j] = c1 == W - 1;
    off2[j] = (unsigned)(8 * half * HW + min(c1, W - 2));
  }
  const unsigned offl = (unsigned)(8 * half * HW + max(2 * wo0 - 1, 0));
  const bool left_pad = wo0 == 0;         // wave-uniform
  const bool odd_w = (W & 1) != 0;        // uniform
  const int src_lane4 = 4 * (l31 > 0 ? lane - 1 : lane + 31);
  bool rowok[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) rowok[kh] = hi0 + kh >= 0 && hi0 + kh < H;

  // buffer loads: one descriptor per wave over its utterance's (Cin, H, W) block; the per-lane byte offset (this half's
  // 8 channels + the column pair) never changes, the (chunk, channel, row) part travels in the SCALAR offset operand -
  // no vector arithmetic per load
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (size_t)b * a.Cin * HW), (short)0, (int)((unsigned)a.Cin * (unsigned)HW * 4u), 0x00020000);
  unsigned voff2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) voff2[j] = 4u * off2[j];
  const unsigned voffl = 4u * offl;
  f32x2 ld2[NT][8];  // the loads in flight: [tile][channel] = columns (2 wo, 2 wo + 1)
  float ldl[8];      // column 2 wo0 - 1 (used by lane 0 of tile 0)
  auto load_row = [&](int chunk, int kh) {
    const unsigned s0 = 4u * (unsigned)((chunk * CK) * HW + (hi0 + kh) * W);  // uniform
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned so = s0 + 4u * (unsigned)(i * HW);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        ld2[j][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, voff2[j], so, 0));
      ldl[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, voffl, so, 0));  // (masked in gather_row)
    }
  };
  // raw[tile][kernel column][channel] from the landed loads
  float raw[NT][3][8];
  auto gather_row = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float prev = left_pad ? 0.0f : ldl[i];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float c1 = ld2[j][i].x, c2 = ld2[j][i].y;
        if (odd_w) {
          c1 = fix_last[j] ? c2 : c1;
          c2 = fix_last[j] ? 0.0f : c2;
        }
        const float t = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, c2)));
        raw[j][0][i] = l31 > 0 ? t : prev;
        raw[j][1][i] = c1;
        raw[j][2][i] = c2;
        prev = t;  // (for lane 0: lane 31's second element of this tile)
      }
    }
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(slab)));
  const u32x4* __restrict__ wsrc = a.wp + (size_t)cot * nchunk * SLAB_U4;
  auto dma = [&](int chunk, int buf) {
    const u32x4* __restrict__ src = wsrc + (size_t)chunk * SLAB_U4;
#pragma unroll
    for (int i = 0; i < (SLAB_U4 / 64 + NWAVE - 1) / NWAVE; ++i) {
      const int blk = wave + NWAVE * i;  // 1 KB block (wave-uniform)
      if (blk < SLAB_U4 / 64)
        dma16(reinterpret_cast<const float*>(src + blk * 64 + lane), lds0 + (unsigned)(buf * SLAB_BYTES + blk * 1024));
    }
  };

  f32x16 acc[2][NT], accs[SC ? 2 : 1][SC ? NT : 1];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[m][j][r] = 0.0f;
        if (SC) accs[m][j][r] = 0.0f;
      }

  // The K loop walks the VALID kernel rows (a padding row contributes nothing and is skipped: wave-uniform) of every
  // chunk as one flat sequence, software-pipelined two rows deep:
  //   row it + 2: loads in flight | row it + 1: gathered and split into planes (VALU) | row it: 3 x 24 MFMAs
  // so that the split of the next row can be issued in the shadow of this row's MFMAs.
  const int kh0 = rowok[0] ? 0 : 1;                                         // valid rows are contiguous
  const int nv = (rowok[0] ? 1 : 0) + (rowok[1] ? 1 : 0) + (rowok[2] ? 1 : 0);
  const int total = nchunk * nv;
  auto split_row = [&](u32x4 (&bp)[3][NT][3]) {
    gather_row();
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#if BF3_EXP & 2
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bp[c][j][0][q] = __builtin_bit_cast(unsigned, raw[j][c][q]);
          bp[c][j][1][q] = __builtin_bit_cast(unsigned, raw[j][c][q + 4]);
          bp[c][j][2][q] = __builtin_bit_cast(unsigned, raw[j][c][(q + 2) & 7]);
        }
#else
        split3(raw[j][c], bp[c][j][0], bp[c][j][1], bp[c][j][2]);
#endif
      }
  };
  u32x4 bpA[3][NT][3], bpB[3][NT][3];  // [kernel column][tile][plane] of two consecutive rows
  // (chunk, r) of the row being computed, one and two rows ahead
  int c0 = 0, r0 = 0, c1, r1, c2, r2;
  auto advance = [&](int c, int r, int& cn, int& rn) {
    rn = r + 1;
    cn = c;
    if (rn == nv) { rn = 0; cn = c + 1; }
  };
  advance(c0, r0, c1, r1);
  advance(c1, r1, c2, r2);

  dma(0, 0);
  load_row(0, kh0);
  split_row(bpA);                              // row 0
  load_row(total > 1 ? c1 : 0, kh0 + (total > 1 ? r1 : 0));  // row 1 in flight
  dma_wait();
  __syncthreads();
  if (nchunk > 1) dma(1, 1);

  // Per row: 3 x 6 x 2 x NT MFMAs, and between them - pinned by sched_barrier, one item behind every second MFMA so
  // that each runs in the shadow of the matrix pipe - the work that prepares row it + 1 and fetches row it + 2:
  //   items 0 .. 8     channel i: bpermute issue (item i) / selects into raw + the loads of row it + 2 (item i + 1)
  //   items 9 ..       the pair splits, (pair q, column c, tile j): 11 VALU each
  float tperm[NT];  // bpermute results in flight (channel i, consumed one item later)
  float pc1[NT], pc2[NT];
  auto body = [&](u32x4 (&cur)[3][NT][3], u32x4 (&nxt)[3][NT][3], int it) {
    const u32x4* __restrict__ sl0 = slab + (c0 & 1) * SLAB_U4 + lane;
    const u32x4* __restrict__ sl = sl0 + (kh0 + r0) * (3 * 2 * 3 * 64);
    const bool more = it + 2 < total;  // (past the end the loads redo the last row: harmless)
    const unsigned s_next = 4u * (unsigned)(((more ? c2 : c1) * CK) * HW + (hi0 + kh0 + (more ? r2 : r1)) * W);
    auto perm_issue = [&](int i) {
#if BF3_EXP & 4
      const float esc = g_bf3_exp_scale[c1 * CK + 8 * half + i], esh = g_bf3_exp_shift[c1 * CK + 8 * half + i];
      ldl[i] = fmaxf(fmaf(ldl[i], esc, esh), 0.0f);
#endif
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float c1v = ld2[j][i].x, c2v = ld2[j][i].y;
#if BF3_EXP & 4
        c1v = fmaxf(fmaf(c1v, esc, esh), 0.0f);
        c2v = fmaxf(fmaf(c2v, esc, esh), 0.0f);
#endif
        if (odd_w) {
          c1v = fix_last[j] ? c2v : c1v;
          c2v = fix_last[j] ? 0.0f : c2v;
        }
        pc1[j] = c1v;
        pc2[j] = c2v;
        tperm[j] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, c2v)));
      }
    };
    auto perm_take = [&](int i) {
      float prev = left_pad ? 0.0f : ldl[i];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        raw[j][0][i] = l31 > 0 ? tperm[j] : prev;
        raw[j][1][i] = pc1[j];
        raw[j][2][i] = pc2[j];
        prev = tperm[j];
      }
#if !(BF3_EXP & 1)
      const unsigned so = s_next + 4u * (unsigned)(i * HW);  // the loads of row it + 2, channel i
#pragma unroll
      for (int j = 0; j < NT; ++j)
        ld2[j][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, voff2[j], so, 0));
      ldl[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, voffl, so, 0));
#endif
    };
    auto pair_split = [&](int k) {  // k = (q * 3 + c) * NT + j: pair q needs channels 2 q, 2 q + 1 (taken by item 2 q + 2)
      const int j = k % NT, c = (k / NT) % 3, q = k / (3 * NT);
      const float av = raw[j][c][2 * q], bv = raw[j][c][2 * q + 1];
#if BF3_EXP & 2
      nxt[c][j][0][q] = __builtin_bit_cast(unsigned, av);
      nxt[c][j][1][q] = __builtin_bit_cast(unsigned, bv);
      nxt[c][j][2][q] = __builtin_bit_cast(unsigned, av) ^ __builtin_bit_cast(unsigned, bv);
#else
      const unsigned h = pack2(av, bv);
      const float ra = av - __builtin_bit_cast(float, h << 16), rb = bv - __builtin_bit_cast(float, h & 0xffff0000u);
      const unsigned m = pack2(ra, rb);
      const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
      unsigned hh = h, mm = m, ll = pack2(sa, sb);
      // (pins the split HERE: without it the compiler sinks the whole computation to its first use, the next row's
      // MFMAs, i.e. out of the shadow of this row's)
      asm volatile("" : "+v"(hh), "+v"(mm), "+v"(ll));
      nxt[c][j][0][q] = hh;
      nxt[c][j][1][q] = mm;
      nxt[c][j][2][q] = ll;
#endif
    };
    constexpr int NITEM = 9 + 4 * 3 * NT;
    auto item = [&](int k) {
      if (k < 9) {
        if (k >= 1) perm_take(k - 1);
        if (k < 8) perm_issue(k);
      } else {
        pair_split(k - 9);
      }
    };
    u32x4 ap[2][3], apn[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) ap[m][pl] = sl[(m * 3 + pl) * 64];
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};  // six products, small terms first
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      if (kw < 2) {  // the next tap's weight fragments, read under this tap's MFMAs
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) apn[m][pl] = sl[(((kw + 1) * 2 + m) * 3 + pl) * 64];
      }
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j) {  // (consecutive MFMAs go to different accumulators)
            const int n = ((kw * 6 + p) * 2 + m) * NT + j;
            acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[m][PA[p]]),
                                                                __builtin_bit_cast(bf16x8, cur[kw][j][PB[p]]), acc[m][j],
                                                                0, 0, 0);
#if BF3_SCHED
            if ((n & 1) == 0 && n / 2 < NITEM) item(n / 2);
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ap[m][pl] = apn[m][pl];
    }
#if !BF3_SCHED
#pragma unroll
    for (int k = 0; k < NITEM; ++k) item(k);
#endif
    if (SC && kh0 + r0 == 1) {  // the centre row (never padding): its middle column is the shortcut's input pixel
      u32x4 as[2][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) as[m][pl] = sl0[((9 * 2 + m) * 3 + pl) * 64];
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            accs[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, as[m][PA[p]]),
                                                                 __builtin_bit_cast(bf16x8, cur[1][j][PB[p]]), accs[m][j],
                                                                 0, 0, 0);
    }
    if (r0 == nv - 1) {  // last row of a chunk: the next chunk's slab has landed; this one's buffer is free for the one after
      dma_wait();
      __syncthreads();
      if (c0 + 2 < nchunk) dma(c0 + 2, c0 & 1);
    }
    c0 = c1; r0 = r1; c1 = c2; r1 = r2;
    advance(c1, r1, c2, r2);
  };
  for (int it = 0; it < total; it += 2) {
    body(bpA, bpB, it);
    if (it + 1 < total) body(bpB, bpA, it + 1);
  }

  // epilogue: D row i = (r & 3) + 8 (r >> 2) + 4 half -> output channel, column l31 -> pixel
  if (!grp_ok) return;
  const size_t plane = (size_t)a.Ho * a.Wo;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int wo = wo0 + 32 * j + l31;
      if (wo >= a.Wo) continue;
      const size_t o0 = ((size_t)b * a.Cout + cot * 64 + m * 32 + 4 * half) * plane + (size_t)ho * a.Wo + wo;
#pragma unroll
      for (int r = 0; r < 16; ++r) a.y[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane] = acc[m][j][r];
      if (SC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a.ysc[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane] = accs[m][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Data gradient of the same layer, joined with the data gradient of the block's 1x1 / stride 2 shortcut
// (resnet.py:56-66), in one pass over dy - conv_s2_dgrad_kernel's formulation (conv2d.hip) on the six-product MFMAs:
// a wave owns 32 consecutive columns j of one dy row i and keeps the four parity classes of dx rows 2 i, 2 i + 1 /
// columns 2 j, 2 j + 1 as 64-channel x 32-column tiles (8 accumulators).  Input pixel (2 i + a, 2 j + b) sees the taps
// kh = 1 (a = 0, dy row i) or kh = 2, 0 (a = 1, dy rows i, i + 1), the same along w.  K = the layer's OUTPUT channels, 16
// per chunk; per chunk a lane loads dy at (i, j .. j + 1), (i + 1, j .. j + 1) and the shortcut's dy at (i, j) for its
// 8 channels (five B fragments, split in registers), the slab holds 10 taps (3 x 3 + the shortcut) x 64 dx channels.
constexpr int DTAPS = 10;
constexpr int DSLAB_U4 = DTAPS * 2 * 3 * 64;
constexpr int DSLAB_BYTES = DSLAB_U4 * 16;  // 61,440

// weights (Cout, Cin, 3, 3) [+ (Cout, Cin, 1, 1)] -> planes, roles swapped: rows = dx channels ci, K = co
//   word((((cot * nchunk + chunk) * 10 + tap) * 2 + mt) * 3 + plane) * 64 + lane): ci = cot * 64 + mt * 32 + (lane & 31),
//   co = chunk * 16 + 8 * (lane >> 5) + 0 .. 7; tap 9 = the shortcut (zeros without one)
__global__ __launch_bounds__(256) void bf3_pack_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ wsc,
                                                             u32x4* __restrict__ wp, int Cout, int Cin, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = e & 63;
  int blk = e >> 6;
  const int mt = blk & 1;
  blk >>= 1;
  const int tap = blk % DTAPS;
  blk /= DTAPS;
  const int nchunk = Cout / CK;
  const int chunk = blk % nchunk, cot = blk / nchunk;
  const int ci = cot * 64 + mt * 32 + (lane & 31);
  const int co0 = chunk * CK + 8 * (lane >> 5);
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    v[i] = tap < 9 ? w[((size_t)(co0 + i) * Cin + ci) * 9 + tap] : (wsc ? wsc[(size_t)(co0 + i) * Cin + ci] : 0.0f);
  u32x4 hi, mid, lo;
  split3(v, hi, mid, lo);
  u32x4* dst = wp + ((size_t)(e >> 6) * 3) * 64 + lane;
  dst[0] = hi;
  dst[64] = mid;
  dst[128] = lo;
}

struct Bf3dArgs {
  const float* dy;          // (B, K, Ho, Wo)
  const float* dysc;        // (B, K, Ho, Wo) gradient of the shortcut's output (SC)
  const u32x4* wp;
  float* dx;                // (B, M, H, W)
  const float* accumulate;  // like dx, may alias it (may be null)
  int B, K, M, H, W, Ho, Wo;
  int WT, ntiles, ncot;
  int pair;                 // W even and dx / accumulate 8-byte aligned: classes (a, 0), (a, 1) leave as one float2
};

template <bool SC>
__global__ __launch_bounds__(NWAVE * 64, 1) void conv_s2d_bf3_kernel(const Bf3dArgs a) {
  __shared__ __attribute__((aligned(16))) u32x4 slab[2 * DSLAB_U4];
  constexpr int NPOS = SC ? 5 : 4;   // B fragments per chunk: dy at (i, j), (i, j + 1), (i + 1, j), (i + 1, j + 1), shortcut dy
  constexpr int NTAP = SC ? 10 : 9;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lb % a.ncot;
  const int pxg = lb / a.ncot;
  const int nt = pxg * NWAVE + wave;
  const bool tile_ok = nt < a.ntiles;
  const int ntc = min(nt, a.ntiles - 1);
  const int wt = ntc % a.WT;
  const int rowid = ntc / a.WT;
  const int i = rowid % a.Ho, b = rowid / a.Ho;
  const int j0 = wt * 32;
  const int j = j0 + l31;
  const int Wo = a.Wo;
  const int HWo = a.Ho * Wo;
  const int nchunk = a.K / CK;

  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned nrec = (unsigned)a.K * (unsigned)HWo * 4u;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.dy + (size_t)b * a.K * HWo), (short)0, (int)nrec, 0x00020000);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((SC ? a.dysc : a.dy) + (size_t)b * a.K * HWo), (short)0, (int)nrec, 0x00020000);
  // the lane's column pair (j, j + 1), clamped into the row; what the clamp changed is fixed when the values are taken
  const bool shifted = Wo >= 2 && j == Wo - 1;   // pair loaded one column early: element 1 is column j
  const bool ok_c1 = j + 1 < Wo;
  const unsigned voff2 = 4u * (unsigned)(8 * half * HWo + max(min(j, Wo - 2), 0));
  const unsigned voff1 = 4u * (unsigned)(8 * half * HWo + min(j, Wo - 1));
  const bool ok_r1 = i + 1 < a.Ho;  // wave-uniform: dy row i + 1 exists (else its loads are sent out of range: zeros)

  f32x2 ld0[8], ld1[8];  // loads in flight: rows i, i + 1, columns (j, j + 1)
  float lds_[8];         // the shortcut's dy at (i, j)
  auto load_chunk = [&](int chunk) {
    const unsigned s0 = 4u * (unsigned)((chunk * CK) * HWo + i * Wo);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const unsigned so = s0 + 4u * (unsigned)(c * HWo);
      ld0[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrs, voff2, so, 0));
      ld1[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrs, voff2, ok_r1 ? so + 4u * (unsigned)Wo : nrec, 0));
      if (SC) lds_[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, voff1, so, 0));
    }
  };
  float raw[NPOS][8];
  auto take = [&](int c) {
    const float x0 = ld0[c].x, y0 = ld0[c].y, x1 = ld1[c].x, y1 = ld1[c].y;
    raw[0][c] = shifted ? y0 : x0;
    raw[1][c] = ok_c1 ? y0 : 0.0f;
    raw[2][c] = shifted ? y1 : x1;
    raw[3][c] = ok_c1 ? y1 : 0.0f;
    if (SC) raw[4][c] = lds_[c];
  };

  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(slab)));
  const u32x4* __restrict__ wsrc = a.wp + (size_t)cot * nchunk * DSLAB_U4;
  auto dma = [&](int chunk, int buf) {
    const u32x4* __restrict__ src = wsrc + (size_t)chunk * DSLAB_U4;
#pragma unroll
    for (int n = 0; n < (DSLAB_U4 / 64 + NWAVE - 1) / NWAVE; ++n) {
      const int blk = wave + NWAVE * n;
      if (blk < NTAP * 6)
        dma16(reinterpret_cast<const float*>(src + blk * 64 + lane), lds0 + (unsigned)(buf * DSLAB_BYTES + blk * 1024));
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][m][r] = 0.0f;

  auto pair_split = [&](u32x4 (&dst)[NPOS][3], int k) {  // k = q * NPOS + pos
    const int pos = k % NPOS, q = k / NPOS;
    const float av = raw[pos][2 * q], bv = raw[pos][2 * q + 1];
    const unsigned h = pack2(av, bv);
    const float ra = av - __builtin_bit_cast(float, h << 16), rb = bv - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = pack2(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    unsigned hh = h, mm = m, ll = pack2(sa, sb);
    asm volatile("" : "+v"(hh), "+v"(mm), "+v"(ll));  // (pins the split here: see conv_s2_bf3_kernel)
    dst[pos][0][q] = hh;
    dst[pos][1][q] = mm;
    dst[pos][2][q] = ll;
  };
  u32x4 bpA[NPOS][3], bpB[NPOS][3];

  dma(0, 0);
  load_chunk(0);
#pragma unroll
  for (int c = 0; c < 8; ++c) take(c);
#pragma unroll
  for (int k = 0; k < 4 * NPOS; ++k) pair_split(bpA, k);
  load_chunk(nchunk > 1 ? 1 : 0);
  dma_wait();
  __syncthreads();
  if (nchunk > 1) dma(1, 1);

  // tap t < 9 = (kh, kw) = (t / 3, t % 3): kh = 0 reads dy row i + 1, kw = 0 column j + 1; tap 9 = the shortcut's dy
  constexpr int POS[10] = {3, 2, 2, 1, 0, 0, 1, 0, 0, 4};
  constexpr int CLS[10] = {3, 2, 3, 1, 0, 1, 3, 2, 3, 0};
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};  // six products, small terms first
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  constexpr int NITEM = 8 + 4 * NPOS;
  auto body = [&](u32x4 (&cur)[NPOS][3], u32x4 (&nxt)[NPOS][3], int chunk) {
    const u32x4* __restrict__ sl = slab + (chunk & 1) * DSLAB_U4 + lane;
    const int cn = min(chunk + 2, nchunk - 1);  // (past the end the loads redo the last chunk: harmless)
    const unsigned s_next = 4u * (unsigned)((cn * CK) * HWo + i * Wo);
    auto item = [&](int k) {
      if (k < 8) {
        take(k);  // chunk + 1, channel k; then its registers take the loads of chunk + 2
        const unsigned so = s_next + 4u * (unsigned)(k * HWo);
        ld0[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrs, voff2, so, 0));
        ld1[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrs, voff2, ok_r1 ? so + 4u * (unsigned)Wo : nrec, 0));
        if (SC) lds_[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, voff1, so, 0));
      } else {
        pair_split(nxt, k - 8);
      }
    };
    // taps two at a time (their classes differ, so consecutive MFMAs never share an accumulator)
    u32x4 ap[2][2][3], apn[2][2][3];
    auto rd = [&](u32x4 (&d)[2][2][3], int tp) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const int tap = 2 * tp + t;
            if (tap < NTAP) d[t][m][pl] = sl[((tap * 2 + m) * 3 + pl) * 64];
          }
    };
    rd(ap, 0);
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) {
      if (tp < 4) rd(apn, tp + 1);
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int tap = 2 * tp + t;
            const int n = ((tp * 6 + p) * 2 + t) * 2 + m;
            if (tap < NTAP)
              acc[CLS[tap]][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, ap[t][m][PA[p]]), __builtin_bit_cast(bf16x8, cur[POS[tap]][PB[p]]),
                  acc[CLS[tap]][m], 0, 0, 0);
            if ((n & 1) == 0 && n / 2 < NITEM) item(n / 2);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) ap[t][m][pl] = apn[t][m][pl];
    }
    dma_wait();
    __syncthreads();
    if (chunk + 2 < nchunk) dma(chunk + 2, chunk & 1);
  };
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
    body(bpA, bpB, chunk);
    if (chunk + 1 < nchunk) body(bpB, bpA, chunk + 1);
  }

  // epilogue (conv_s2_dgrad_kernel's): D row (r & 3) + 8 (r >> 2) + 4 half -> dx channel, column l31 -> dy column j;
  // classes (a, 0) and (a, 1) of a lane are neighbours in dx row 2 i + a
  if (!tile_ok || j >= Wo) return;
  const size_t plane = (size_t)a.H * a.W;
  const bool ok_b1 = 2 * j + 1 < a.W;
  const bool pair = a.pair != 0;
#pragma unroll
  for (int pa = 0; pa < 2; ++pa) {
    const int h = 2 * i + pa;
    if (h >= a.H) continue;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int cb = cot * 64 + 32 * m + 4 * half;
      const size_t o0 = ((size_t)b * a.M + cb) * plane + (size_t)h * a.W + 2 * j;
      float v0[16], v1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v0[r] = acc[2 * pa][m][r];
        v1[r] = acc[2 * pa + 1][m][r];
      }
      if (a.accumulate != nullptr) {  // every load of the tile before its first store
        float t0[16], t1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const size_t o = o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane;
          if (pair) {
            const float2 t = *reinterpret_cast<const float2*>(a.accumulate + o);
            t0[r] = t.x;
            t1[r] = t.y;
          } else {
            t0[r] = a.accumulate[o];
            t1[r] = ok_b1 ? a.accumulate[o + 1] : 0.0f;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v0[r] += t0[r];
          v1[r] += t1[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* __restrict__ o = a.dx + o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane;
        if (pair) {
          *reinterpret_cast<float2*>(o) = make_float2(v0[r], v1[r]);
        } else {
          o[0] = v0[r];
          if (ok_b1) o[1] = v1[r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layer: dW[co][ci][kh][kw] = sum over (b, ho, wo) of dy[b][co][ho][wo] x[b][ci][2 ho + kh - 1]
// [2 wo + kw - 1].  K = pixels, 16 per MFMA (a run of 16 consecutive wo of one output row), BOTH operands are activations:
// A = dy (a lane = one output channel, 8 consecutive wo = 32 contiguous bytes), B = x (a lane = one input channel, the 17
// consecutive columns 2 wo0 - 1 .. 2 wo0 + 15 behind its 8 pixels: the even / odd / shifted-even elements are the three
// kw fragments).  No LDS, no barriers: a wave is on its own - it owns ONE kernel row kh, 128 output channels x 32 input
// channels x 3 kw (12 accumulators) and a segment of the output rows, splits 4 + 3 fragments per 72 MFMAs, and writes
// its sums into the slice of its segment; reduce_partials_kernel (conv2d.hip) adds the slices in order.
struct Bf3wArgs {
  const float* x;    // (B, Cin, H, W)
  const float* dy;   // (B, Cout, Ho, Wo)
  float* partial;    // [nseg][9][Cout][Cin]
  int B, Cin, H, W, Cout, Ho, Wo;
  int ncog, ncit;    // Cout / 128, Cin / 32
  int nseg, rps;     // segments of rps output rows (b, ho)
  int KS;            // 16-pixel K steps per output row
};

template <class F, int... Ns>
__device__ __forceinline__ void bf3_seq(F&& f, std::integer_sequence<int, Ns...>) {
  (f(std::integral_constant<int, Ns>{}), ...);
}

// acc[N / 3][N % 3] += A B as ONE v_mfma_f32_32x32x16_bf16 in place on the fixed AGPR tuple a[16 N : 16 N + 15].
// Through the builtin (and through a plain "+a" constraint alike) the register allocator moves whole accumulator tuples
// between AGPR ranges over the weight-gradient kernel's two-step loop body and copies them back at the loop header:
// 452 v_accvgpr moves per 144 MFMAs, which made that loop VALU-bound (9 VALU instructions per MFMA; 6 without them).
// hipcc pads no hazards for an asm statement: the s_nop covers an operand the allocator brings back with a
// v_accvgpr_read right in front of the MFMA (conv_wino4.hip, round 3).
#if BF3W_ASM_MFMA
#define BF3W_PIN(N, LO, HI)                                                              \
  if constexpr (NI == N)                                                                  \
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"                     \
                 : "+{a[" #LO ":" #HI "]}"(acc[N / 3][N % 3])                             \
                 : "v"(a), "v"(b));
template <int NI>
__device__ __forceinline__ void bf3w_mfma(f32x16 (&acc)[4][3], const u32x4& a, const u32x4& b) {
  BF3W_PIN(0, 0, 15) BF3W_PIN(1, 16, 31) BF3W_PIN(2, 32, 47) BF3W_PIN(3, 48, 63) BF3W_PIN(4, 64, 79) BF3W_PIN(5, 80, 95)
  BF3W_PIN(6, 96, 111) BF3W_PIN(7, 112, 127) BF3W_PIN(8, 128, 143) BF3W_PIN(9, 144, 159) BF3W_PIN(10, 160, 175)
  BF3W_PIN(11, 176, 191)
}
#undef BF3W_PIN
#else
template <int NI>
__device__ __forceinline__ void bf3w_mfma(f32x16 (&acc)[4][3], const u32x4& a, const u32x4& b) {
  acc[NI / 3][NI % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                                acc[NI / 3][NI % 3], 0, 0, 0);
}
#endif

__global__ __launch_bounds__(NWAVE * 64, 1) void conv_s2w_bf3_kernel(const Bf3wArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  // work unit = (segment, kh, channel-group pair); the pair index is fastest so that the waves of a workgroup and
  // neighbouring workgroups stream the same rows (L2)
  int u = blockIdx.x * NWAVE + wave;  // (an XCD-contiguous order measured no better: l2s same, l4s 0.29 -> 0.36 ms)
  const int npair = a.ncog * a.ncit;
  const int nunit = a.nseg * 3 * npair;
  if (u >= nunit) return;
  const int pr = u % npair;
  u /= npair;
  const int kh = u % 3, seg = u / 3;
  const int cog = pr / a.ncit, cit = pr % a.ncit;
  const int H = a.H, W = a.W, Wo = a.Wo, Ho = a.Ho;
  const int HW = H * W, HWo = Ho * Wo;
  const int KS = a.KS;
  const int nrows = a.B * Ho;
  const int row0 = seg * a.rps, row1 = min(row0 + a.rps, nrows);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), (short)0, (int)((unsigned)a.B * (unsigned)a.Cin * (unsigned)HW * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.dy), (short)0, (int)((unsigned)a.B * (unsigned)a.Cout * (unsigned)HWo * 4u), 0x00020000);
  unsigned voffA[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) voffA[m] = 4u * (unsigned)((cog * 128 + m * 32 + l31) * HWo + 8 * half);
  const unsigned voffB = 4u * (unsigned)((cit * 32 + l31) * HW + 16 * half);

  typedef float f32x4u __attribute__((ext_vector_type(4)));
  // step = (row, ks) over the rows of the segment whose input row 2 ho + kh - 1 exists
  auto row_ok = [&](int rid) { const int hi = 2 * (rid % Ho) + kh - 1; return hi >= 0 && hi < H; };
  auto next_step = [&](int& rid, int& ks) {  // advance to the next valid step (rid == row1: past the end)
    if (++ks < KS) return;
    ks = 0;
    ++rid;
    while (rid < row1 && !row_ok(rid)) ++rid;
  };
  int rid0 = row0, ks0 = 0;
  while (rid0 < row1 && !row_ok(rid0)) ++rid0;

  f32x4u la[4][2];   // loads in flight / landed: dy, 8 pixels per channel tile
  f32x4u lb[4];      // x: columns 2 wo0' .. 2 wo0' + 15 of this lane's 8 pixels (wo0' = 16 ks + 8 half)
  float xm1 = 0.0f;  //    column 2 wo0' - 1: the left neighbour's last column (below)
  float carry = 0.0f;
  auto offs = [&](int rid, int ks, unsigned& sa, unsigned& sb) {
    const int ridc = min(rid, row1 - 1);  // (past the end: redo the last row - harmless)
    const int b = ridc / Ho, ho = ridc - b * Ho;
    const int wo0 = 16 * ks;
    sa = 4u * (unsigned)(b * a.Cout * HWo + ho * Wo + wo0);
    sb = 4u * (unsigned)(b * a.Cin * HW + (2 * ho + kh - 1) * W + 2 * wo0);
  };
  auto issue_a = [&](int m, unsigned sa) {
    la[m][0] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(yrs, voffA[m], sa, 0));
    la[m][1] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(yrs, voffA[m] + 16u, sa, 0));
  };
  auto issue_b = [&](unsigned sb) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      lb[q] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(xrs, voffB + 16u * q, sb, 0));
  };
  // Column 2 wo0' - 1 is never loaded (for the first step of the first row it would lie in front of the tensor): for the
  // upper half-wave it is the last column of the lower half's span (lane - 32), for the lower half the last column the
  // upper half held one step ago (the previous 16 pixels of the row).  One ds_bpermute per step: the upper half sends what
  // it kept from the previous step, the lower half its current last column; at the first step of a row the lower half's
  // value is column -1 = padding (masked in xval).
  auto left_col = [&]() {
    const float last = lb[3][3];
    const float send = half ? carry : last;
    xm1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __builtin_bit_cast(int, send)));
    carry = last;
  };

  f32x16 acc[4][3];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][kw][r] = 0.0f;

  // What lies outside the image / the row must enter the sums as zero.  dy of pixels wo >= Wo (the ragged last step of
  // a row) is selected to zero - then whatever x holds behind the row's end meets a zero; x itself needs only column -1
  // (first step) and, for odd W, the kw = 2 column of the last valid pixel.  The per-lane conditions are the same for
  // every row (launch constants, kept as lane masks); a step ORs in "this is not the first / last step" (scalar), so the
  // price is one v_cndmask per dy value and nine per x fragment set, no compares - selecting in place under a uniform
  // branch, or branching between a masked and an unmasked copy of the step, spilled ~500 registers.
  bool mkA[8], mkB2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int wo = 16 * (KS - 1) + 8 * half + i;
    mkA[i] = wo < Wo;
    mkB2[i] = 2 * wo + 1 < W;
  }
  const bool mkB0 = half != 0;
  auto aval = [&](int m, int i, int ks) -> float {  // dy of tile m, pixel i of this lane
    const float v = la[m][i >> 2][i & 3];
    return (ks != KS - 1 || mkA[i]) ? v : 0.0f;
  };
  auto xval = [&](int i, int ks) -> float {         // x at column 2 (16 ks + 8 half) - 1 + i
    const float v = i == 0 ? xm1 : lb[(i - 1) >> 2][(i - 1) & 3];
    if (i == 0) return (ks != 0 || mkB0) ? v : 0.0f;
    if (i >= 2 && (i & 1) == 0) return (ks != KS - 1 || mkB2[i / 2 - 1]) ? v : 0.0f;  // element 2 p + 2 = pixel p's kw = 2
    return v;
  };
  auto pair_split = [&](float av, float bv, u32x4 (&dst)[3], int q) {
#if BF3_EXP & 2
    dst[0][q] = __builtin_bit_cast(unsigned, av);
    dst[1][q] = __builtin_bit_cast(unsigned, bv);
    dst[2][q] = __builtin_bit_cast(unsigned, av) ^ __builtin_bit_cast(unsigned, bv);
    return;
#endif
    const unsigned h = pack2(av, bv);
    const float r0 = av - __builtin_bit_cast(float, h << 16), r1 = bv - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = pack2(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    unsigned hh = h, mm = m, ll = pack2(s0, s1);
    asm volatile("" : "+v"(hh), "+v"(mm), "+v"(ll));  // (pins the split here: see conv_s2_bf3_kernel)
    dst[0][q] = hh;
    dst[1][q] = mm;
    dst[2][q] = ll;
  };
  // Register budget: VALU operands live in 256 VGPRs (the 12 accumulators take 192 AGPRs), so only the x planes are
  // double buffered; the dy planes ROLL: the MFMAs run tile-major (18 per dy tile m), tile m's planes are dead behind its
  // block and take the NEXT step's tile m at once, and tile 3 - alive to the end of the step - is split at the start of
  // the step that consumes it.  Per step s (every item behind every second MFMA):
  //   block m = 0:  dy tile 3 of step s (4 pair splits; its loads for step s + 1 behind them), x of step s + 1 (5 of 12)
  //   block m = 1:  dy tile 0 of step s + 1 (4; its loads for step s + 2 behind them),        x of step s + 1 (4)
  //   block m = 2:  dy tile 1 of step s + 1 (4; loads),                                        x of step s + 1 (3; x loads)
  //   block m = 3:  dy tile 2 of step s + 1 (4; loads)
  u32x4 ap[4][3], bpA[3][3], bpB[3][3];

  if (rid0 < row1) {
    int r1 = rid0, k1 = ks0;   // step s + 1 while step s computes (here: step 0)
    int kc;                    // ks of the step being computed (the masks of its tile 3)
    unsigned sa, sb;
    offs(r1, k1, sa, sb);
#pragma unroll
    for (int m = 0; m < 4; ++m) issue_a(m, sa);
    issue_b(sb);
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) pair_split(aval(m, 2 * q, k1), aval(m, 2 * q + 1, k1), ap[m], q);
    left_col();
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int q = 0; q < 4; ++q) pair_split(xval(4 * q + kw, k1), xval(4 * q + 2 + kw, k1), bpA[kw], q);
    kc = k1;
    next_step(r1, k1);   // (r1, k1) = step 1: tiles 0 .. 2 and x go out now; la[3] still holds step 0's tile 3
    offs(r1, k1, sa, sb);
#pragma unroll
    for (int m = 0; m < 3; ++m) issue_a(m, sa);
    issue_b(sb);

    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};  // six products, small terms first
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    auto body = [&](u32x4 (&bcur)[3][3], u32x4 (&bnxt)[3][3]) {
      // offsets of step s + 1 (tile 3's reload) and of step s + 2 (everything else)
      unsigned sa1, sb1, sa2, sb2;
      offs(r1, k1, sa1, sb1);
      int r2 = r1, k2 = k1;
      if (r2 < row1) next_step(r2, k2);
      offs(r2, k2, sa2, sb2);
      auto item = [&](int k) {  // slot k of 36
        if (k < 4) {            // dy tile 3 of THIS step
          pair_split(aval(3, 2 * k, kc), aval(3, 2 * k + 1, kc), ap[3], k);
          if (k == 3 && !(BF3_EXP & 1)) issue_a(3, sa1);
          if (k == 2) left_col();  // (x of step s + 1 has landed; the bpermute is consumed two slots later)
        } else if (k < 9) {     // x of step s + 1, pairs 0 .. 4
          const int e = k - 4;
          pair_split(xval(4 * (e & 3) + (e >> 2), k1), xval(4 * (e & 3) + 2 + (e >> 2), k1), bnxt[e >> 2], e & 3);
        } else if (k < 13 || (k >= 18 && k < 22) || (k >= 27 && k < 31)) {  // dy tile 0 / 1 / 2 of step s + 1
          const int m = k < 13 ? 0 : (k < 22 ? 1 : 2), q = k - (k < 13 ? 9 : (k < 22 ? 18 : 27));
          pair_split(aval(m, 2 * q, k1), aval(m, 2 * q + 1, k1), ap[m], q);
          if (q == 3 && !(BF3_EXP & 1)) issue_a(m, sa2);
        } else if ((k >= 13 && k < 17) || (k >= 22 && k < 25)) {            // x of step s + 1, pairs 5 .. 11
          const int e = k < 17 ? k - 13 + 5 : k - 22 + 9;
          pair_split(xval(4 * (e & 3) + (e >> 2), k1), xval(4 * (e & 3) + 2 + (e >> 2), k1), bnxt[e >> 2], e & 3);
          if (e == 11 && !(BF3_EXP & 1)) issue_b(sb2);
        }
      };
      // 72 MFMAs, tile-major: n = (6 m + p) 3 + kw - consecutive MFMAs cycle through the tile's three accumulators.  The
      // indices are compile-time (integer sequence): every MFMA names the fixed AGPR tuple of its accumulator.
      auto one = [&](auto nc) {
        constexpr int n = decltype(nc)::value, m = n / 18, p = (n / 3) % 6, kw = n % 3;
        if (m == 3 && p == 0 && kw == 0) __builtin_amdgcn_sched_barrier(0);
        bf3w_mfma<m * 3 + kw>(acc, ap[m][PA[p]], bcur[kw][PB[p]]);
        if ((n & 1) == 0) item(n / 2);
        __builtin_amdgcn_sched_barrier(0);
      };
      bf3_seq(one, std::make_integer_sequence<int, 72>{});
      kc = k1;
      r1 = r2; k1 = k2;
    };
    // (r1 is the step AFTER the one being computed; the loop ends when the computed step was the last valid one)
    for (;;) {
      const bool more = r1 < row1;
      body(bpA, bpB);
      if (!more) break;
      const bool more2 = r1 < row1;
      body(bpB, bpA);
      if (!more2) break;
    }
  }

  // sums -> this segment's slice, [tap][co][ci]: D row (r & 3) + 8 (r >> 2) + 4 half -> co, column l31 -> ci
  float* __restrict__ out = a.partial + (size_t)seg * 9 * a.Cout * a.Cin;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cog * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[((size_t)(kh * 3 + kw) * a.Cout + co) * a.Cin + cit * 32 + l31] = acc[m][kw][r];
      }
}

int pick_nt(int Wo) {
  // pixel tiles of 32 per wave: the split that wastes the fewest pixel columns (Wo = 375: 6 x 64 = 384; 188: 3 x 64 = 192;
  // 94: 1 x 96)
  const int w2 = (Wo + 63) / 64 * 64, w3 = (Wo + 95) / 96 * 96;
  return w3 < w2 ? 3 : 2;
}

}  // namespace

bool air_bf3_s2_ok(int B, int Cin, int H, int W, int Cout) {
  // (a buffer descriptor per utterance: one utterance's (Cin, H, W) block below 4 GB)
  return (air_opt(AIR_OPT_CONV_S2) & 4) != 0 && B > 0 && Cin % CK == 0 && Cout % 64 == 0 && H >= 2 && W >= 2 &&
         4ull * (unsigned long long)Cin * H * W < (1ull << 32);
}

size_t air_bf3_s2_packed_bytes(int Cout, int Cin, bool with_shortcut) {
  return (size_t)(Cout / 64) * (Cin / CK) * (with_shortcut ? 10 : TAPS) * 2 * 3 * 1024;
}

int air_bf3_s2_weights(const float* w, const float* w_sc, void* packed, int Cout, int Cin, hipStream_t st) {
  const int ntaps = w_sc ? 10 : TAPS;
  const int total = (Cout / 64) * (Cin / CK) * ntaps * 2 * 64;
  hipLaunchKernelGGL(bf3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, w_sc, reinterpret_cast<u32x4*>(packed),
                     Cout, Cin, ntaps, total);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_bf3_s2_fwd(const float* x, const void* packed, float* y, float* y_sc, int B, int Cin, int H, int W, int Cout, int Ho,
                   int Wo, double flops, hipStream_t st) {
  Bf3Args a;
  a.x = x; a.wp = reinterpret_cast<const u32x4*>(packed); a.y = y; a.ysc = y_sc;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Ho = Ho; a.Wo = Wo;
  const int nt = y_sc ? 2 : pick_nt(Wo);  // (three pixel tiles + the shortcut's accumulators do not fit the register file)
  a.WT = (Wo + 32 * nt - 1) / (32 * nt);
  a.ngroups = B * Ho * a.WT;
  a.ncot = Cout / 64;
  const int nblk = ((a.ngroups + NWAVE - 1) / NWAVE) * a.ncot;
  // MFMA FLOPs issued: 6 products per algorithmic multiply-add
  AirProfScope ps(AIR_K_CONV_S2_BF3, flops, st, 6.0 * flops);
  if (y_sc)
    hipLaunchKernelGGL((conv_s2_bf3_kernel<2, true>), dim3(nblk), dim3(NWAVE * 64), 0, st, a);
  else if (nt == 3)
    hipLaunchKernelGGL((conv_s2_bf3_kernel<3, false>), dim3(nblk), dim3(NWAVE * 64), 0, st, a);
  else
    hipLaunchKernelGGL((conv_s2_bf3_kernel<2, false>), dim3(nblk), dim3(NWAVE * 64), 0, st, a);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

bool air_bf3_s2d_ok(int B, int Cin, int H, int W, int Cout) {
  return (air_opt(AIR_OPT_CONV_S2) & 8) != 0 && B > 0 && Cout % CK == 0 && Cin % 64 == 0 && H >= 2 && W >= 2 &&
         4ull * (unsigned long long)Cout * H * W < (1ull << 32);  // (dy block of one utterance, generously)
}

size_t air_bf3_s2d_packed_bytes(int Cout, int Cin) { return (size_t)(Cin / 64) * (Cout / CK) * DSLAB_BYTES; }

int air_bf3_s2d_weights(const float* w, const float* w_sc, void* packed, int Cout, int Cin, hipStream_t st) {
  const int total = (Cin / 64) * (Cout / CK) * DTAPS * 2 * 64;
  hipLaunchKernelGGL(bf3_pack_dgrad_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, w_sc,
                     reinterpret_cast<u32x4*>(packed), Cout, Cin, total);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_bf3_s2d_dgrad(const float* dy, const float* dy_sc, const void* packed, float* dx, const float* accumulate, int B,
                      int Cin, int H, int W, int Cout, int Ho, int Wo, double flops, hipStream_t st) {
  Bf3dArgs a;
  a.dy = dy; a.dysc = dy_sc; a.wp = reinterpret_cast<const u32x4*>(packed); a.dx = dx; a.accumulate = accumulate;
  a.B = B; a.K = Cout; a.M = Cin; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.WT = (Wo + 31) / 32;
  a.ntiles = B * Ho * a.WT;
  a.ncot = Cin / 64;
  a.pair = (W % 2 == 0) && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(accumulate)) & 7) == 0;
  const int nblk = ((a.ntiles + NWAVE - 1) / NWAVE) * a.ncot;
  AirProfScope ps(AIR_K_CONV_S2D_BF3, flops, st, 6.0 * flops);
  if (dy_sc)
    hipLaunchKernelGGL(conv_s2d_bf3_kernel<true>, dim3(nblk), dim3(NWAVE * 64), 0, st, a);
  else
    hipLaunchKernelGGL(conv_s2d_bf3_kernel<false>, dim3(nblk), dim3(NWAVE * 64), 0, st, a);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

bool air_bf3_s2w_ok(int B, int Cin, int H, int W, int Cout) {
  // (one buffer descriptor over the WHOLE x and dy tensors, 32-bit byte offsets: both must stay below 4 GB - B = 64 at
  // the bench shape is 0.7 GB; beyond that the f32 kernel, which addresses with 64 bits, takes the layer)
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const unsigned long long xb = 4ull * (unsigned long long)B * Cin * H * W, yb = 4ull * (unsigned long long)B * Cout * Ho * Wo;
  return (air_opt(AIR_OPT_CONV_S2) & 16) != 0 && B > 0 && Cout % 128 == 0 && Cin % 32 == 0 && H >= 2 && W >= 2 &&
         xb < (1ull << 32) && yb < (1ull << 32);
}

// segments of output rows: enough waves for ~2 per SIMD-slot of the chip, at least 8 K steps each
int air_bf3_s2w_nseg(int B, int Cin, int Ho, int Wo, int Cout) {
  const int per_seg = 3 * (Cout / 128) * (Cin / 32);
  const int nrows = B * Ho, KS = (Wo + 15) / 16;
  int nseg = (2048 + per_seg - 1) / per_seg;
  int rps = (nrows + nseg - 1) / nseg;
  while (rps * KS < 8 && rps < nrows) ++rps;
  if (rps < 1) rps = 1;
  return (nrows + rps - 1) / rps;
}

int air_bf3_s2w_partials(const float* x, const float* dy, float* partial, int B, int Cin, int H, int W, int Cout, int Ho,
                         int Wo, double flops, hipStream_t st) {
  Bf3wArgs a;
  a.x = x; a.dy = dy; a.partial = partial;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Ho = Ho; a.Wo = Wo;
  a.ncog = Cout / 128; a.ncit = Cin / 32;
  a.nseg = air_bf3_s2w_nseg(B, Cin, Ho, Wo, Cout);
  const int nrows = B * Ho;
  a.rps = (nrows + a.nseg - 1) / a.nseg;
  a.KS = (Wo + 15) / 16;
  const int nunit = a.nseg * 3 * a.ncog * a.ncit;
  AirProfScope ps(AIR_K_CONV_S2W_BF3, flops, st, 6.0 * flops);
  hipLaunchKernelGGL(conv_s2w_bf3_kernel, dim3((nunit + NWAVE - 1) / NWAVE), dim3(NWAVE * 64), 0, st, a);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}
