// 3x3 / stride 1 / pad 1 convolution as Winograd F(4x4,3x3) on the f32 MFMA of gfx950.
//
// Replaces the 3x3 nn.Conv2d of PreActBlock (resnet.py:56-61) in forward and data-gradient direction.
// Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A with 6x6 transformed tiles: 36 multiplies per 4x4 output
// tile instead of 144 (F(2x2,3x3) in conv_wino.hip: 16 per 2x2 = 64 per 4x4), i.e. 36 independent GEMMs
//   M_p[co][tile] = sum_ci U_p[co][ci] V_p[ci][tile]
// run as v_mfma_f32_16x16x4_f32 (16 co x 16 tiles x 4 ci, 32 cycles):
//   * a wave owns 32 output channels (2 MFMA row blocks) x 16 tiles: 72 accumulators x 4 registers = 288,
//     of which 64 are pinned to the AGPR half of the register file and 8 to VGPRs through inline-asm
//     constraints (left to the compiler, 288 accumulators get shuffled between the two halves);
//   * the f32 MFMA shares the issue port with the VALU on this part (tools/ubench/mfma16_rot.hip: 32 cycles
//     per MFMA + 4 per VALU instruction, +10 for every MFMA -> VALU -> MFMA switch), so nothing is gained
//     by interleaving: a k-step is [72 MFMAs with the 18 ds_read_b128 of their A operands running three
//     groups ahead] then [the 6x6 input transform of the next step as one batch of 144 VALU operations]:
//     3250 cycles per k-step in isolation (tools/ubench/wino4_loop.hip) against 2304 for the bare MFMAs;
//   * a lane transforms the patch of ITS tile and ITS input channel (the B operand layout of the MFMA:
//     lane = 16 ci + tile), read from LDS-staged image rows; transformed weights come pre-packed from a
//     small transform kernel; nothing transformed ever touches HBM; the output transform is per lane.
//   * staging is hand-issued LDS-DMA (buffer_load_dwordx4 ... lds): 16-byte chunks aligned to image columns
//     that are multiples of 4, so a chunk is inside or outside the row as a whole (outside = the buffer's
//     out-of-range rule = zero padding); only the chunk that straddles the right edge when W % 4 != 0
//     brings the next row's first pixels, and the reading wave zeroes those cells in LDS.
//   * weights (U) and patches run through separate 3-deep LDS rings: the patch of step s+1 is read during
//     step s (transform) and U(s) during step s (MFMA operands), so at step s the DMAs of U(s+2) and
//     patch(s+3) are issued and everything lands two steps (about 3 us) before it is read.
// A workgroup = 4 waves = 4 tile groups sharing one 32-channel weight slab; persistent workgroups walk
// contiguous work items (32 channels x 4 groups) and the k-step stream runs on across item boundaries.
// Tile groups are 1 x 16 or 2 x 8 tiles of the image rows STACKED over the batch (tile row = b * TH + th),
// so short images (layer4: 3 x 94) fill the 16-tile MFMA width with tiles of two images.
//
// Round 3.  (a) Output tiles are MH x 4 with MH = 4 or 3 (template): F(3x4,3x3) has 5 x 6 = 30 positions for
// 12 outputs (2.5 multiplies per output against 2.25), but the ResNet's feature maps are 18 / 9 / 5 / 3 rows
// high: 4-row tiles compute 20 / 12 / 8 / 4 rows for them, 3-row tiles 18 / 9 / 6 / 3, so layers 2-4 issue
// 60 MFMAs per k-step instead of 72 for the same tile count (and all 60 accumulators fit the AGPR half).
// (b) Work items are dealt to the workgroups XCD by XCD: an XCD owns a contiguous range of the item list,
// numbered so that 32 consecutive items are (up to) 8 channel slabs x 4 tile quads, and its workgroups take
// items round-robin - at any time the workgroups of one XCD read the same patches (the slabs of a quad) and the
// same transformed weights (the quads of a slab), which then hit that XCD's L2 instead of each workgroup
// fetching its own copy through the fabric (round 2: 3.8x the algorithmic bytes).  (c) When the last round of
// an XCD is at most half full its items are cut in two k-halves: the workgroups that take the upper halves do
// so FIRST, store partial sums into y and raise a flag; the owners of the lower halves meet them LAST, add the
// stored sums and the residual.  A workgroup publishes before it ever waits.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "air_common.h"
#include "air_options.h"
#include "air_lds_dma.h"
#include "air_prof.h"
#include "conv_wino.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int W4_CO = 32;                       // output channels per workgroup
constexpr int W4_CK = 4;                        // input channels per k-step (MFMA K)
constexpr int W4_NBUF = 3;
// Partial sums of a cut item travel between two workgroups that may sit on different XCDs, whose L2s are not
// coherent with each other: stored and loaded with the sc1 (agent-scope) cache policy they go through to the
// memory side, and the flag (an agent-scope atomic, sc1 too) follows the stores' acknowledgement.  Round 2
// used an agent-scope RELEASE fence instead, which writes back the publishing XCD's whole L2: affordable for the
// one layer that cut items then, a 7 % loss on layer1 when every layer's last round is cut.
constexpr int W4_SC1 = 16;
#ifndef W4_DMA_PLACE
#define W4_DMA_PLACE 0  // A/B: 0 = one DMA per MFMA group boundary; 1 = inside the transform; 2 = two per boundary
#endif
// TIMING-ONLY experiment builds (results are garbage; python -m asvspoof2021_air_amd.build --variant ...; never the
// default library).  W4_EXP_RETILE = R emulates the staging stream of a workgroup tile of 32 R output channels x
// 64 / R tiles on this kernel's schedule: every weight-slab DMA is issued R times (from R different channel slabs),
// only NI / R (rounded up) of the patch DMAs go out.  W4_EXP_NOXF additionally drops the input transform - the
// upper bound of what sharing V between the waves of such a tile could save.  profiles/r04_wino4_retile.md.
#ifndef W4_EXP_RETILE
#define W4_EXP_RETILE 1
#endif
#ifndef W4_PATCH_FIRST
#define W4_PATCH_FIRST 0  // A/B: 1 = round 2's order (patch reads, then the first operand reads)
#endif
constexpr unsigned W4_OOB = 0x80000000u;        // byte offset beyond any tensor we accept: reads as zero

// Positions and the layout of the transformed weights for MH x 4 output tiles.
//   MH = 4: 36 positions, slab [ci 4][co 32][36] (a lane's 36 floats are contiguous: 144-byte lane stride,
//           conflict-free 16-byte reads), 4608 floats = 4.5 DMAs of 1024 floats, LDS pitch 5120;
//   MH = 3: 30 positions in 8 quads (the last one half used), slab [ci 4][quad 8][co 32][4] (the 16 lanes of
//           an MFMA row block read consecutive 16-byte words), 4096 floats = exactly 4 DMAs.
template <int MH>
struct W4Pos {
  static constexpr int NPR = MH + 2;            // patch rows = positions along H
  static constexpr int NP = NPR * 6;            // Winograd positions
  static constexpr int NQ = (NP + 3) / 4;       // position quads per MFMA row block (A operands are read 4 at a time)
  static constexpr int USLAB = MH == 4 ? W4_CK * W4_CO * NP : W4_CK * NQ * W4_CO * 4;  // floats per k-step
  static constexpr int NU = (USLAB + 1023) / 1024;  // DMAs (256 threads x 16 bytes) per slab
  static constexpr int ULDS = NU * 1024;        // LDS pitch of a slab
  static constexpr int NACC = 2 * NP;           // accumulators per wave: 2 row blocks x positions
  // float offset of (ci k, channel col, position p) inside a slab
  __host__ __device__ static constexpr int uoff(int k, int col, int p) {
    return MH == 4 ? (k * W4_CO + col) * NP + p : ((k * NQ + (p >> 2)) * W4_CO + col) * 4 + (p & 3);
  }
};

template <int TRG, int MH>
struct W4Cfg {
  static constexpr int TCG = 16 / TRG;          // tile columns of a group
  static constexpr int RC = TCG + 2;            // 16-byte chunks per staged row: columns 4 TCG twg - 4 ...
  static constexpr int ROWF = 4 * RC;           // floats per staged row
  static constexpr int NPR = MH + 2;            // input rows of one tile row
  static constexpr int BANDC = NPR * RC;        // chunks per band
  static constexpr int BANDF = 4 * BANDC;
  // A (group, ci) plane = TRG bands, rounded up to 16 chunks (64 floats: planes stay bank-aligned for the 16-byte
  // reads).  Round 3 gave every plane 128 chunks: 8 DMAs per thread and k-step of which the 1 x 16 groups of three
  // of the ResNet's four layers used 90 / 128 - and an LDS-DMA costs the issuing wave ~50 - 60 cycles between its
  // MFMAs whether its lanes fetch or not.  Now the 16 planes are packed back to back and the 256 threads walk the
  // flat chunk index: 6 DMAs (MH = 3, 1 x 16), 7 (MH = 3, 2 x 8 and MH = 4, 1 x 16) or 8 (MH = 4, 2 x 8).
  static constexpr int PLC = (TRG * BANDC + 15) / 16 * 16;  // chunks per plane
  static constexpr int PLF = 4 * PLC;           // floats per (group, ci) plane
  static constexpr int GRPF = W4_CK * PLF;      // floats per tile group
  static constexpr int PATCHF = 4 * GRPF;       // floats per patch buffer
  static constexpr int NI = 16 * PLC / 256;     // patch DMAs per thread and k-step
  static_assert(16 * PLC % 256 == 0 && PLC <= 128, "16 planes = a whole number of 256-thread DMAs");
};

// Textbook interpolation points (0, +-1, +-2, inf):
//   G   = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
//   B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
//   A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
// Error budget (tests/golden/wino4_points.py: numpy emulation in fp32 against fp64 on the ResNet's layer
// shapes; GPU measurements agree): 4e-6 .. 1.05e-5 of the output scale, against 1e-6 .. 3e-6 for the direct
// f32 convolution and 3e-7 .. 9e-7 for F(2x2,3x3); inside the 2e-5 every conv test allows.  The point set
// (0, 1, -1, 2, -1/2, inf) measured 2e-6 .. 5e-6 but costs 16 instead of 12 operations per 1-D input
// transform (+8 % kernel time) and did not move the model-level gradient errors, which are set by ReLU /
// sign flips of the ill-conditioned filler-initialised net, not by the convolutions' rounding: not used.
__device__ __forceinline__ void w4_g6(double g0, double g1, double g2, double& u0, double& u1, double& u2,
                                      double& u3, double& u4, double& u5) {
  u0 = g0 / 4.0;
  u1 = -(g0 + g1 + g2) / 6.0;
  u2 = -(g0 - g1 + g2) / 6.0;
  u3 = g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
  u4 = g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
  u5 = g2;
}

// F(3,3), points (0, 1, -1, 2, inf):
//   G   = [[1/2,0,0],[-1/2,-1/2,-1/2],[-1/6,1/6,-1/6],[1/6,1/3,2/3],[0,0,1]]
//   B^T = [[2,-1,-2,1,0],[0,-2,-1,1,0],[0,2,-3,1,0],[0,-1,0,1,0],[0,2,-1,-2,1]]
//   A^T = [[1,1,1,1,0],[0,1,-1,2,0],[0,1,1,4,1]]
__device__ __forceinline__ void w4_g5(double g0, double g1, double g2, double& u0, double& u1, double& u2,
                                      double& u3, double& u4) {
  u0 = g0 / 2.0;
  u1 = -(g0 + g1 + g2) / 2.0;
  u2 = -(g0 - g1 + g2) / 6.0;
  u3 = g0 / 6.0 + g1 / 3.0 + g2 * (2.0 / 3.0);
  u4 = g2;
}

// U = G_h g G_w^T in double, rounded once (G_h: 6 or 5 points along H, G_w: 6 points along W).
// Packed [cot][chunk][slab of W4Pos<MH>].
// (bid of nb blocks: the launch's own grid, or this layer's share of a batched launch)
template <int MH>
__device__ __forceinline__ void w4_weights_body(const float* __restrict__ w, float* __restrict__ up, int M, int Kc,
                                                int dgrad, int bid, int nb) {
  using P = W4Pos<MH>;
  const int nchunk = Kc / W4_CK;
  const int Mpad = (M + W4_CO - 1) / W4_CO * W4_CO;  // a 16-channel tail runs as a slab whose upper rows are zero
  const size_t total = (size_t)Mpad * Kc;
  for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < total;
       e += (size_t)nb * blockDim.x) {
    const int col = (int)(e % W4_CO);
    size_t r = e / W4_CO;
    const int k = (int)(r % Kc);
    const int cot = (int)(r / Kc);
    const int m = cot * W4_CO + col;
    double g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
      g[t] = m >= M ? 0.0 : (dgrad ? w[((size_t)k * M + m) * 9 + (8 - t)] : w[((size_t)m * Kc + k) * 9 + t]);
    double tmp[6][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (MH == 4) w4_g6(g[c], g[3 + c], g[6 + c], tmp[0][c], tmp[1][c], tmp[2][c], tmp[3][c], tmp[4][c], tmp[5][c]);
      else w4_g5(g[c], g[3 + c], g[6 + c], tmp[0][c], tmp[1][c], tmp[2][c], tmp[3][c], tmp[4][c]);
    }
    float* o = up + ((size_t)cot * nchunk + k / W4_CK) * P::USLAB;
    const int kk = k % W4_CK;
#pragma unroll
    for (int i = 0; i < P::NPR; ++i) {
      double u[6];
      w4_g6(tmp[i][0], tmp[i][1], tmp[i][2], u[0], u[1], u[2], u[3], u[4], u[5]);
#pragma unroll
      for (int j = 0; j < 6; ++j) o[P::uoff(kk, col, 6 * i + j)] = (float)u[j];
    }
    if (P::NP % 4 != 0) {  // the unused half of the last position quad: multiplied into nothing, kept finite
      for (int p = P::NP; p < 4 * P::NQ; ++p) o[P::uoff(kk, col, p)] = 0.0f;
    }
  }
}

template <int MH>
__global__ void wino4_weights_kernel(const float* __restrict__ w, float* __restrict__ up, int M, int Kc,
                                     int dgrad) {
  w4_weights_body<MH>(w, up, M, Kc, dgrad, (int)blockIdx.x, (int)gridDim.x);
}

// The transforms of SEVERAL layers in one launch (air_conv2d_prepack_begin / _flush, conv2d.hip): the ResNet's 26
// per-step launches of 5 - 25 us each sit between the persistent convolution kernels of the other stream, which leave
// no CU for them until a workgroup retires - each costs its own gap.  Job j owns blocks [blk0[j], blk0[j + 1]).
constexpr int W4_WJOBS = 32;
struct W4WeightJobs {
  const float* w[W4_WJOBS];
  float* up[W4_WJOBS];
  int M[W4_WJOBS], Kc[W4_WJOBS];
  unsigned char dgrad[W4_WJOBS], mh[W4_WJOBS];
  int blk0[W4_WJOBS + 1];
  int n;
};
__global__ void wino4_weights_batch_kernel(const W4WeightJobs jb) {
  int j = 0;
  while (j + 1 < jb.n && (int)blockIdx.x >= jb.blk0[j + 1]) ++j;
  const int bid = (int)blockIdx.x - jb.blk0[j], nb = jb.blk0[j + 1] - jb.blk0[j];
  if (jb.mh[j] == 3) w4_weights_body<3>(jb.w[j], jb.up[j], jb.M[j], jb.Kc[j], jb.dgrad[j], bid, nb);
  else w4_weights_body<4>(jb.w[j], jb.up[j], jb.M[j], jb.Kc[j], jb.dgrad[j], bid, nb);
}

struct W4Args {
  const float* x;         // (B, Cin, H, W)
  const float* up;        // packed transformed weights
  float* y;               // (B, Cout, H, W)
  const float* residual;  // same shape as y (may be null)
  int B, Cin, H, W, Cout;
  int TH, TW;             // MH x 4 output tiles per image
  int SR;                 // stacked tile rows: B * TH
  int GRR, TWG;           // group rows (of TRG stacked tile rows), group columns
  int ngroups;            // GRR * TWG
  int ncot;               // Cout / 32
  int nquad;              // ceil(ngroups / 4)
  int cotb;               // channel slabs per numbering block: min(ncot, 8)
  int nitems;             // work items: nquad * ncot; id = ((cot / cotb) * nquad + quad) * cotb + cot % cotb
  int nxg;                // item dealing groups (XCDs): workgroup b belongs to group b % nxg
  int xmode;              // 1: a group owns a contiguous range of the item list; 2: the tile quads = group mod nxg
  int split;              // cut the items of a half-empty last round in two k-halves (see the kernel)
  unsigned* flags;        // split: one word per cut item, raised to `epoch` when its upper-half sums are in y
  unsigned epoch;
  long long* trace;       // debug: cycle totals of workgroup 0 (null in production)
  int dephase;            // option WINO4_DEPHASE: every second workgroup of a dealing group starts this many x 4096 cycles late
  // BatchNorm statistics of y from the epilogue (forward only; null = none): a header {G, Cout, 0, 0} and then one
  // record {n, K, sum(y - K), sum((y - K)^2)} per (channel c, tile group g) at [c * G + g], G = 4 * nquad: the
  // (n, mean, M2) of the group's valid outputs in shifted form, merged in fp64 by bn_finalize_records_kernel
  // (norm_act.hip) - the pass over y that air_bn_stats otherwise makes.  air_wino4_stats_bytes() sizes it.
  float* stats;
  // stats_mode 2 (data-gradient launches; same buffer layout): y is dA, the gradient with respect to the OUTPUT of
  // relu(batchnorm(bnx)); the records are {sum g, sum g * xhat, 0, 0} with g = dA where the ReLU passed and
  // xhat = (bnx - mean) * invstd - the two sums of the BatchNorm backward (bn_bwd_partial_kernel, norm_act.hip,
  // same mask and xhat arithmetic), taken while dA is written instead of in a pass that re-reads dA and bnx.
  int stats_mode;          // 0 none, 1 forward statistics, 2 BatchNorm-backward sums
  const float* bnx;        // (B, Cout, H, W): the BatchNorm's input
  const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta;  // (Cout,)
};

__device__ __forceinline__ i32x4 w4_rsrc(const void* base, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32));  // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// One DMA = 3 instructions: M0 = (wave's LDS base in the target buffer) + MIMM, wait state, load.
template <int MIMM>
__device__ __forceinline__ void w4_dma16(i32x4 rsrc, unsigned soff, unsigned mbase, unsigned v0) {
  asm volatile("s_add_i32 m0, %1, %4\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %0, %2 offen lds"
               :: "s"(rsrc), "s"(mbase), "s"(soff), "v"(v0), "n"(MIMM) : "memory", "m0", "scc");
}
template <int N>
__device__ __forceinline__ void w4_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// accumulate / start (C = 0) forms, accumulator pinned to the AGPR ("a") or VGPR ("v") half
#define W4_MFMA_A(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define W4_MFMA_V(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
// The start forms open with two wait states: V[] is live across the epilogue in front of an item's first k-step,
// and when the allocator parks it there (MH = 3 leaves 16 AGPRs free: it uses them as spill space) it brings each
// value back with a v_accvgpr_read right in front of the MFMA that reads it - a VALU write -> MFMA operand read,
// which needs the states and which hipcc does not pad for an asm statement (round 3: NaNs in row block 0).
// tools/audit_asm_hazards.py checks the build's ISA for the same pattern in front of the accumulate forms.
#define W4_MFMA_A0(ACC, A, B) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(ACC) : "v"(A), "v"(B))
// (early clobber: a fresh VGPR destination must not share registers with the A / B operands)
#define W4_MFMA_V0(ACC, A, B) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A), "v"(B))

// one row / column of B^T d, 6 points: 12 operations
__device__ __forceinline__ void w4_bt6(float d0, float d1, float d2, float d3, float d4, float d5, float& t0,
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
// one column of B^T d, 5 points (F(3,3)): 9 operations
__device__ __forceinline__ void w4_bt5(float d0, float d1, float d2, float d3, float d4, float& t0, float& t1,
                                       float& t2, float& t3, float& t4) {
  const float e = d3 - d1;
  t0 = __builtin_fmaf(2.0f, d0 - d2, e);
  t1 = __builtin_fmaf(-2.0f, d1, d3 - d2);
  t2 = __builtin_fmaf(2.0f, d1, __builtin_fmaf(-3.0f, d2, d3));
  t3 = e;
  t4 = __builtin_fmaf(-2.0f, e, d4 - d2);
}
// two columns of B^T d at once, 5 points: the same 9 operations as v_pk_add_f32 / v_pk_fma_f32 (the f32 MFMA and the
// VALU share one issue port - every VALU instruction of the input transform is 4+ cycles the matrix pipe idles)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 w4_fma2(float k, f32x2 a, f32x2 b) {
  return __builtin_elementwise_fma(f32x2{k, k}, a, b);
}
__device__ __forceinline__ void w4_bt5x2(f32x2 d0, f32x2 d1, f32x2 d2, f32x2 d3, f32x2 d4, f32x2& t0, f32x2& t1,
                                         f32x2& t2, f32x2& t3, f32x2& t4) {
  const f32x2 e = d3 - d1;
  t0 = w4_fma2(2.0f, d0 - d2, e);
  t1 = w4_fma2(-2.0f, d1, d3 - d2);
  t2 = w4_fma2(2.0f, d1, w4_fma2(-3.0f, d2, d3));
  t3 = e;
  t4 = w4_fma2(-2.0f, e, d4 - d2);
}
// one row of B^T d, 6 points, on the pairs the packed column pass leaves: E = {d0, d5}, R0 = {d1, d2}, R1 = {d3, d4}.
// 4 packed + 4 scalar instructions for w4_bt6's 12 (the halves are picked with op_sel, no moves):
//   {b, a} = -4 R0 + R1;  {t1, t2} = {a + b, a - b};  {e, c} = R1 - R0;  {t3, t4} = {c + 2 e, c - 2 e}
__device__ __forceinline__ void w4_bt6p(f32x2 E, f32x2 R0, f32x2 R1, float& t0, float& t1, float& t2, float& t3,
                                        float& t4, float& t5) {
  const f32x2 X = __builtin_elementwise_fma(f32x2{-4.0f, -4.0f}, R0, R1);
  const f32x2 t12 = __builtin_elementwise_fma(f32x2{1.0f, -1.0f}, __builtin_shufflevector(X, X, 0, 0),
                                              __builtin_shufflevector(X, X, 1, 1));
  const f32x2 Y = R1 - R0;
  const f32x2 t34 = __builtin_elementwise_fma(f32x2{2.0f, -2.0f}, __builtin_shufflevector(Y, Y, 0, 0),
                                              __builtin_shufflevector(Y, Y, 1, 1));
  t0 = __builtin_fmaf(4.0f, E[0], __builtin_fmaf(-5.0f, R0[1], R1[1]));
  t5 = __builtin_fmaf(4.0f, R0[0], __builtin_fmaf(-5.0f, R1[0], E[1]));
  t1 = t12[0]; t2 = t12[1]; t3 = t34[0]; t4 = t34[1];
}
#ifndef W4_PK
// A/B: 0 = scalar transform; 1 = rows scalar, then two columns at a time - bit-identical to 0, 3.3 % off the kernel
// (tools/exp_wino4_pk.sh: layer1-4 forward 0.289 / 0.243 / 0.288 / 0.306 -> 0.280 / 0.235 / 0.278 / 0.297 ms);
// 2 = columns first (pairs as the 16-byte patch reads deliver them), then the rows on those pairs: 1-2 % more on
// layers 3-4, nothing on 1-2 (the 30 packed intermediates cost 17 more AGPR parkings per transform), and the 6-point
// pass - constants 4 and 5 - then amplifies the 5-point pass's rounding: 1.54e-5 of the output scale at K = 512
// against 1.05e-5, outside the bound tests/_budget.py derives from the emulation.  Not the default.
#define W4_PK 1
#endif
// one row / column of A^T m, 6 points -> 4 outputs: 10 operations
__device__ __forceinline__ void w4_at6(float m0, float m1, float m2, float m3, float m4, float m5, float& y0,
                                       float& y1, float& y2, float& y3) {
  const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  y0 = m0 + s1 + s2;
  y1 = __builtin_fmaf(2.0f, d2, d1);
  y2 = __builtin_fmaf(4.0f, s2, s1);
  y3 = __builtin_fmaf(8.0f, d2, d1) + m5;
}
// one column of A^T m, 5 points -> 3 outputs: 7 operations
__device__ __forceinline__ void w4_at5(float m0, float m1, float m2, float m3, float m4, float& y0, float& y1,
                                       float& y2) {
  const float s1 = m1 + m2, d1 = m1 - m2;
  y0 = m0 + s1 + m3;
  y1 = __builtin_fmaf(2.0f, m3, d1);
  y2 = __builtin_fmaf(4.0f, m3, s1) + m4;
}

template <int TRG, int MH, bool TRACE = false, bool BST = false>
__global__ __launch_bounds__(256) void wino4_conv_kernel(W4Args a) {
  using C = W4Cfg<TRG, MH>;
  using P = W4Pos<MH>;
  constexpr int NPR = P::NPR, NP = P::NP, NQ = P::NQ, NU = P::NU, NI = C::NI;
  constexpr int NIX = (NI + W4_EXP_RETILE - 1) / W4_EXP_RETILE;  // (= NI in every product build)
  constexpr int ND = W4_EXP_RETILE * NU + NIX;  // DMAs per thread and k-step
  constexpr int W4_PLF = C::PLF, W4_GRPF = C::GRPF, W4_PATCHF = C::PATCHF;
  constexpr int ULDS = P::ULDS, USLAB = P::USLAB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(2 * ND <= 63, "vmcnt is a 6-bit counter");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = tile group of the item's 4
  const int HWi = a.H * a.W;
  const int nchunk = a.Cin / W4_CK;
  const int wrem = a.W & 3;  // pixels of the chunk that straddles the right edge

  // ---- which items.  Dealing group xg = blockIdx % nxg (the XCD of this workgroup when nxg == 8: speed only,
  // nothing below depends on where a workgroup really runs); it owns items [R0, R1) of the list and its Wx
  // workgroups take them round-robin: `nfull` whole rounds, then `rem` items for the last one.  rem <= Wx / 2
  // and a.split: each is cut in two k-halves; workgroups j < rem take the LOWER halves as their last segment,
  // workgroups rem <= j < 2 rem the UPPER halves as their FIRST segment (stored as partial sums into y, flag
  // raised), so a flag is up long before its reader arrives and no workgroup waits on one that waits.
  // (Which j takes which half: see is_upper / is_lower below - the text above names the roles, not the indices.)
  const int xg = (int)blockIdx.x % a.nxg;
  const int Wx = ((int)gridDim.x - xg + a.nxg - 1) / a.nxg;
  // j: this workgroup's index inside its dealing group, counted from the END of the group: the dispatcher hands out
  // workgroups in index order, so were fewer CUs free than the launch has workgroups (another kernel resident, a
  // partitioned device) the owners of the lower halves of cut items - j < rem, the only workgroups that ever wait -
  // start LAST, behind the publishers they wait for (j in [rem, 2 rem)).  Round 4 first swapped the two roles
  // instead; same protection, but layer4 then fetched 838 MB per launch against 670 (tools/exp_l4_traffic.sh:
  // W4_ROLE 2 / 1 / this, 0).
#ifndef W4_ROLE
#define W4_ROLE 0  // A/B: 1 = round 3 (index from the front), 2 = index from the front with the two roles swapped
#endif
  const int j = W4_ROLE == 0 ? Wx - 1 - (int)blockIdx.x / a.nxg : (int)blockIdx.x / a.nxg;
  // VERDICT r4 item 6a (de-phasing): all workgroups walk items of equal length in lockstep, so their store sections
  // collide; started late by a fraction of an item, every second workgroup stores while its neighbours compute
  if (a.dephase > 0 && (j & 1)) {
    for (int d = 0; d < a.dephase; ++d) __builtin_amdgcn_s_sleep(64);
  }
  // xmode 1: group xg owns the contiguous id range [R0, R0 + nx); xmode 2: the quads congruent to xg mod nxg
  // (nqx of them, every channel slab), numbered locally the same way.  Items below are LOCAL ids 0 .. nx - 1.
  const int nqx = (a.nquad - xg + a.nxg - 1) / a.nxg;
  const int R0 = a.xmode == 2 ? 0 : (int)((long long)xg * a.nitems / a.nxg);
  const int nx = a.xmode == 2 ? nqx * a.ncot : (int)((long long)(xg + 1) * a.nitems / a.nxg) - R0;
  const int nround = nx / Wx, rem = nx - nround * Wx;
  const bool cut = a.split && rem > 0 && 2 * rem <= Wx && nchunk >= 2;
  const int half = nchunk >> 1;
  // segments of this workgroup: [upper half of a cut item]? whole items* [lower half | whole tail item]?
  int tail_seg = -1, tail_item = 0, fc = 0, last_end = nchunk, nseg = nround;
  const int full_base = j;
  const bool is_upper = cut && (W4_ROLE == 2 ? j < rem : (j >= rem && j < 2 * rem));
  const bool is_lower = cut && (W4_ROLE == 2 ? (j >= rem && j < 2 * rem) : j < rem);
  const int jt = j >= rem ? j - rem : j;  // index of the cut item this workgroup shares
  if (is_lower) { tail_seg = nround; tail_item = nround * Wx + jt; last_end = half; ++nseg; }
  else if (is_upper) { tail_seg = 0; tail_item = nround * Wx + jt; fc = half; ++nseg; }
  else if (!cut && j < rem) { tail_seg = nround; tail_item = nround * Wx + j; ++nseg; }
  if (a.stats != nullptr && blockIdx.x == 0 && tid == 0) {  // header of the statistics buffer, once per launch
    int* h = reinterpret_cast<int*>(a.stats);
    h[0] = 4 * a.nquad;
    h[1] = a.Cout;
    h[2] = 0;
    h[3] = 0;
  }
  if (nseg == 0) return;
  const int S = nround * nchunk + (tail_seg < 0 ? 0 : (is_upper ? nchunk - half : (is_lower ? half : nchunk)));
  const int flag_idx = xg * 32 + (tail_item - nround * Wx);  // cut items of this launch: < 8 * 32
  long long tk0 = 0, tw0 = 0;
  if (TRACE) { tk0 = clock64(); tw0 = wall_clock64(); }
  const int c0 = fc;
  auto seg_item = [&](int seg) {
    return seg == tail_seg ? tail_item : full_base + (seg - (is_upper ? 1 : 0)) * Wx;
  };
  auto seg_end = [&](int seg) { return seg == nseg - 1 ? last_end : nchunk; };
  // item id -> (tile quad, channel slab): 32 consecutive ids = cotb slabs x (32 / cotb) quads
  const int per_block = (a.xmode == 2 ? nqx : a.nquad) * a.cotb;
  auto item_quad = [&](int item) {
    const int q = ((R0 + item) % per_block) / a.cotb;
    return a.xmode == 2 ? xg + a.nxg * q : q;
  };
  auto item_cot = [&](int item) { return ((R0 + item) / per_block) * a.cotb + (R0 + item) % a.cotb; };
  const int i0 = seg_item(0);

  float* const ldsU = lds;
  float* const ldsP = lds + W4_NBUF * ULDS;
  const i32x4 xrs = w4_rsrc(a.x, (unsigned)a.B * a.Cin * HWi * 4u);
  const i32x4 urs = w4_rsrc(a.up, (unsigned)a.ncot * (unsigned)nchunk * (USLAB * 4u));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  const unsigned mU0 = lds0 + wave * 1024u;
  const unsigned mP0 = lds0 + W4_NBUF * ULDS * 4u + wave * 1024u;

  // ---- staging cursors.  Behind the end of the stream they stay on the last k-step (restaged into free
  // buffers), so issue counts - and with them the vmcnt waits - never vary.
  unsigned voff[NI];  // byte offsets of this thread's patch chunks (out of range = zeros)
  // DMA i, thread t stages flat chunk f = 256 i + t of the buffer: plane f / PLC = 4 group + ci, slot f % PLC =
  // (band, row, chunk) of that plane.  Everything is re-derived from the thread index per item and not kept live
  // (the k-step loop has no register to spare: see conv_wino.hip).
  auto set_voff = [&](int item) {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int quad = item_quad(item);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int f = i * 256 + t;
      const int pl = f / C::PLC, slot = f - pl * C::PLC;
      const int band = slot / C::BANDC, rem3 = slot - band * C::BANDC;
      const int row = rem3 / C::RC, cc = rem3 - row * C::RC;
      const int g = 4 * quad + (pl >> 2);
      const int gr = g % a.GRR, twg = g / a.GRR;
      const int sr = gr * TRG + band;
      const int b = sr / a.TH, th = sr - b * a.TH;
      const int hi = MH * th - 1 + row, col = 4 * C::TCG * twg - 4 + 4 * cc;
      const bool ok = slot < TRG * C::BANDC && g < a.ngroups && sr < a.SR && hi >= 0 && hi < a.H && col >= 0 && col < a.W;
      voff[i] = ok ? (unsigned)((((b * a.Cin + (pl & 3)) * a.H + hi) * a.W + col) * 4) : W4_OOB;
    }
  };
  int pSeg = 0, pChunk = c0, pBuf = 0, pLeft = S;
  int uSeg = 0, uChunk = c0, uBuf = 0, uLeft = S;
  unsigned xso = __builtin_amdgcn_readfirstlane((unsigned)c0 * (unsigned)(W4_CK * HWi * 4)), mP = mP0, mU = mU0;
  auto u_item_base = [&](int item) {
    return (unsigned)item_cot(item) * (unsigned)nchunk * (USLAB * 4u);
  };
  unsigned ubase = u_item_base(i0), uso = __builtin_amdgcn_readfirstlane(ubase + (unsigned)c0 * (USLAB * 4u));
  const unsigned uvoff = (unsigned)tid * 16u;
  set_voff(i0);

  // DMA unit u of the restaging of one k-step: 0 .. NU - 1 = the weight slab (MH = 4: 4608 floats, the fifth
  // DMA's upper half lands in the pad behind it - LDS pitch 5120 - and reads the next slab's head or, behind
  // the last one, out of range; MH = 3: exactly four), NU .. NU + 7 = the patches.
  auto dma_unit = [&](auto unit_tag) {
    constexpr int u = decltype(unit_tag)::value;
#if W4_EXP_RETILE > 1
    // flat unit index: R NU weight DMAs (slab unit u / R, from channel slab + u % R), then NIX patch DMAs
    if constexpr (u < W4_EXP_RETILE * NU)
      w4_dma16<(u / W4_EXP_RETILE) * 4096>(urs, uso + (unsigned)(u / W4_EXP_RETILE) * 4096u +
                                           (unsigned)(u % W4_EXP_RETILE) * (unsigned)nchunk * (USLAB * 4u), mU, uvoff);
    else if constexpr (u < ND) w4_dma16<(u - W4_EXP_RETILE * NU) * 4096>(xrs, xso, mP, voff[u - W4_EXP_RETILE * NU]);
#else
    if constexpr (u < NU) w4_dma16<u * 4096>(urs, uso + (unsigned)u * 4096u, mU, uvoff);
    else if constexpr (u < ND) w4_dma16<(u - NU) * 4096>(xrs, xso, mP, voff[u - NU]);
#endif
  };
  auto adv_patch = [&]() {
    pBuf = pBuf + 1 == W4_NBUF ? 0 : pBuf + 1;
    if (pLeft > 1) {
      --pLeft;
      if (++pChunk == seg_end(pSeg)) {
        pChunk = 0;
        ++pSeg;
        set_voff(seg_item(pSeg));
      }
    }
    xso = __builtin_amdgcn_readfirstlane((unsigned)pChunk * (unsigned)(W4_CK * HWi * 4));
    mP = __builtin_amdgcn_readfirstlane(mP0 + (unsigned)pBuf * (W4_PATCHF * 4u));
  };
  auto adv_u = [&]() {
    uBuf = uBuf + 1 == W4_NBUF ? 0 : uBuf + 1;
    if (uLeft > 1) {
      --uLeft;
      if (++uChunk == seg_end(uSeg)) {
        uChunk = 0;
        ++uSeg;
        ubase = u_item_base(seg_item(uSeg));
      }
    }
    uso = __builtin_amdgcn_readfirstlane(ubase + (unsigned)uChunk * (USLAB * 4u));
    mU = __builtin_amdgcn_readfirstlane(mU0 + (unsigned)uBuf * (ULDS * 4u));
  };
#define W4_UNIT(U_) dma_unit(std::integral_constant<int, U_>{})
  constexpr int NUX = W4_EXP_RETILE * NU;  // (= NU in every product build)
  auto dma_u = [&]() {
    W4_UNIT(0); W4_UNIT(1); W4_UNIT(2); W4_UNIT(3);
    if constexpr (NUX > 4) W4_UNIT(4);
#if W4_EXP_RETILE > 1
    W4_UNIT(5); W4_UNIT(6); W4_UNIT(7);
    if constexpr (NUX > 8) { W4_UNIT(8); W4_UNIT(9); W4_UNIT(10); W4_UNIT(11); W4_UNIT(12); W4_UNIT(13); W4_UNIT(14); W4_UNIT(15); }
    if constexpr (NUX > 16) { W4_UNIT(16); W4_UNIT(17); W4_UNIT(18); W4_UNIT(19); }
    static_assert(NUX <= 20, "experiment builds: R <= 4");
#endif
    adv_u();
  };
  auto dma_patch = [&]() {
    W4_UNIT(NUX + 0); W4_UNIT(NUX + 1); W4_UNIT(NUX + 2); W4_UNIT(NUX + 3);
    W4_UNIT(NUX + 4); W4_UNIT(NUX + 5); W4_UNIT(NUX + 6); W4_UNIT(NUX + 7);
    adv_patch();
  };

  // ---- compute state
  // accumulator cb * NP + p: the first 64 in AGPRs, the rest (MH = 4: 8) in VGPRs
  f32x4 accA[P::NACC < 64 ? P::NACC : 64], accV[P::NACC > 64 ? P::NACC - 64 : 1];
  float V[NP];         // B^T d B of the step about to be multiplied
  float raw[NPR * 6];  // patch of the next step
  int pb_lane = 0, ub_lane = 0;
  auto lane_consts = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int jl = t & 15, k = (t >> 4) & 3;
    const int tr = jl / C::TCG, tc = jl - tr * C::TCG;
    pb_lane = wave * W4_GRPF + k * W4_PLF + tr * C::BANDF + 4 * tc;
    ub_lane = P::uoff(k, jl, 0);
  };
  // staged-row float offset of image column W - W % 4 for this wave's group of `item`, or -1
  auto edge_of = [&](int item) {
    if (wrem == 0) return -1;
    const int g = 4 * item_quad(item) + wave;
    const int twg = g / a.GRR;
    const int cc = (a.W - wrem - (4 * C::TCG * twg - 4)) >> 2;
    return (g < a.ngroups && cc >= 0 && cc < C::RC) ? 4 * cc + wrem : -1;
  };
  auto fix_edge = [&](float* pbuf, int eoff) {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int l = t & 63;
    if (l < 4 * NPR * TRG) {
      const int ci = l / (NPR * TRG), rb = l - ci * (NPR * TRG);
      float* p = pbuf + wave * W4_GRPF + ci * W4_PLF + rb * C::ROWF + eoff;  // bands are contiguous rows
      p[0] = 0.0f;
      if (wrem < 3) p[1] = 0.0f;
      if (wrem < 2) p[2] = 0.0f;
    }
  };
  auto read_patch = [&](const float* pbuf) {
    const float* p = pbuf + pb_lane;
#pragma unroll
    for (int r = 0; r < NPR; ++r) {
      raw[6 * r] = p[r * C::ROWF + 3];
      const f32x4 m = *reinterpret_cast<const f32x4*>(p + r * C::ROWF + 4);
      raw[6 * r + 1] = m[0];
      raw[6 * r + 2] = m[1];
      raw[6 * r + 3] = m[2];
      raw[6 * r + 4] = m[3];
      raw[6 * r + 5] = p[r * C::ROWF + 8];
    }
  };
  // with_dma (W4_DMA_PLACE == 1): the k-step's restaging DMAs go out between the transform's sub-steps
  auto transform = [&](auto dma_tag) {
    constexpr bool WITH_DMA = decltype(dma_tag)::value;
    float t[NPR * 6];
    auto unit = [&](auto u_tag) {
      if constexpr (WITH_DMA) { __builtin_amdgcn_sched_barrier(0); dma_unit(u_tag); __builtin_amdgcn_sched_barrier(0); }
    };
#define W4_TU(U_) unit(std::integral_constant<int, U_>{})
    if constexpr (MH == 3 && W4_PK == 2 && !WITH_DMA) {
      // down the columns first, two at a time: (0, 5), (1, 2), (3, 4) - the second and third pair are the halves of
      // the row's 16-byte read - then along each row on those pairs: 27 + 40 instructions for 54 + 60.  (Another order
      // of the same sums than rounds 2-3: the rounding differs in the last bits, the error budget does not.)
      f32x2 E[5], R0[5], R1[5];
      w4_bt5x2(f32x2{raw[0], raw[5]}, f32x2{raw[6], raw[11]}, f32x2{raw[12], raw[17]}, f32x2{raw[18], raw[23]},
               f32x2{raw[24], raw[29]}, E[0], E[1], E[2], E[3], E[4]);
      w4_bt5x2(f32x2{raw[1], raw[2]}, f32x2{raw[7], raw[8]}, f32x2{raw[13], raw[14]}, f32x2{raw[19], raw[20]},
               f32x2{raw[25], raw[26]}, R0[0], R0[1], R0[2], R0[3], R0[4]);
      w4_bt5x2(f32x2{raw[3], raw[4]}, f32x2{raw[9], raw[10]}, f32x2{raw[15], raw[16]}, f32x2{raw[21], raw[22]},
               f32x2{raw[27], raw[28]}, R1[0], R1[1], R1[2], R1[3], R1[4]);
#pragma unroll
      for (int i = 0; i < 5; ++i)
        w4_bt6p(E[i], R0[i], R1[i], V[6 * i], V[6 * i + 1], V[6 * i + 2], V[6 * i + 3], V[6 * i + 4], V[6 * i + 5]);
      return;
    }
    W4_TU(0);
#pragma unroll
    for (int r = 0; r < NPR; ++r) {  // along the row: t[r][j] = sum_c BT[j][c] d[r][c]
      w4_bt6(raw[6 * r], raw[6 * r + 1], raw[6 * r + 2], raw[6 * r + 3], raw[6 * r + 4], raw[6 * r + 5],
             t[6 * r], t[6 * r + 1], t[6 * r + 2], t[6 * r + 3], t[6 * r + 4], t[6 * r + 5]);
      switch (r) { case 0: W4_TU(1); break; case 1: W4_TU(2); break; case 2: W4_TU(3); break; case 3: W4_TU(4); break;
                   case 4: W4_TU(5); break; default: W4_TU(6); break; }
    }
    if constexpr (MH == 3 && W4_PK == 1 && !WITH_DMA) {
#pragma unroll
      for (int c = 0; c < 6; c += 2) {  // down two columns at a time
        f32x2 o0, o1, o2, o3, o4;
        w4_bt5x2(f32x2{t[c], t[c + 1]}, f32x2{t[6 + c], t[7 + c]}, f32x2{t[12 + c], t[13 + c]}, f32x2{t[18 + c], t[19 + c]},
                 f32x2{t[24 + c], t[25 + c]}, o0, o1, o2, o3, o4);
        V[c] = o0[0]; V[c + 1] = o0[1];
        V[6 + c] = o1[0]; V[7 + c] = o1[1];
        V[12 + c] = o2[0]; V[13 + c] = o2[1];
        V[18 + c] = o3[0]; V[19 + c] = o3[1];
        V[24 + c] = o4[0]; V[25 + c] = o4[1];
      }
    } else
#pragma unroll
    for (int c = 0; c < 6; ++c) {  // down the column: V[i][j] = sum_r BT_h[i][r] t[r][j]
      if constexpr (MH == 4)
        w4_bt6(t[c], t[6 + c], t[12 + c], t[18 + c], t[24 + c], t[30 + c],
               V[c], V[6 + c], V[12 + c], V[18 + c], V[24 + c], V[30 + c]);
      else
        w4_bt5(t[c], t[6 + c], t[12 + c], t[18 + c], t[24 + c], V[c], V[6 + c], V[12 + c], V[18 + c], V[24 + c]);
      switch (c) { case 0: W4_TU(NPR + 1); break; case 1: W4_TU(NPR + 2); break; case 2: W4_TU(NPR + 3); break;
                   case 3: W4_TU(NPR + 4); break; case 4: W4_TU(NPR + 5); break; default: W4_TU(NPR + 6); break; }
    }
  };

  // v += the same value of the other 15 lanes of its row of 16 (xor 1, xor 2, mirror within 8, mirror within 16)
  auto rowsum16 = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    return v;
  };
  // Y = A_h^T M A_w per lane; D row (l >> 4) * 4 + r -> channel, D column l & 15 -> tile.
  // Output rows go out as 16-byte buffer stores whose offset is pushed out of range for lanes / rows that
  // do not exist (the store is dropped, the residual load returns zero): no branches in the common path.
  // Tiles that straddle the right image edge (W % 4 != 0) store element by element instead.
  const unsigned ybytes = (unsigned)a.B * a.Cout * HWi * 4u;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, (short)0, (int)ybytes, 0x00020000);
  const bool has_res = a.residual != nullptr;
  const bool tracing = TRACE && a.trace != nullptr && blockIdx.x == 0;
  long long tS = 0, tDr = 0, tL = 0;
  // partial: store the sums as they are (upper k-half of a cut item); accum: add what the upper half stored
  auto epilogue = [&](int item, bool partial, bool accum) {
    int t = tid;
    asm volatile("" : "+v"(t));
    const int jl = t & 15, q = (t >> 4) & 3;
    const int tr = jl / C::TCG, tc = jl - tr * C::TCG;
    const int cot = item_cot(item);
    const int g = 4 * item_quad(item) + wave;
    const int gr = g % a.GRR, twg = g / a.GRR;
    int b, th, sr;
    if (TRG == 1) {
      sr = gr;
      b = sr / a.TH;
      th = sr - b * a.TH;
    } else {
      const int sr0 = gr * TRG, sr1 = sr0 + 1;
      const int b0 = sr0 / a.TH, b1 = sr1 / a.TH;
      sr = tr ? sr1 : sr0;
      b = tr ? b1 : b0;
      th = tr ? sr1 - b1 * a.TH : sr0 - b0 * a.TH;
    }
    const int tw = twg * C::TCG + tc;
    const bool valid = g < a.ngroups && sr < a.SR && tw < a.TW;
    const int ho = MH * th, wo = 4 * tw;
    const bool wide = valid && wo + 4 <= a.W;
    const bool part = valid && !wide;
    const int co0 = cot * W4_CO + 4 * q;
    const unsigned obase = (unsigned)(((b * a.Cout + co0) * a.H + ho) * a.W + wo) * 4u;  // bytes
    unsigned orow[MH];
#pragma unroll
    for (int yy = 0; yy < MH; ++yy)
      orow[yy] = (wide && ho + yy < a.H) ? obase + (unsigned)(yy * a.W) * 4u : W4_OOB;
    const unsigned chan = (unsigned)a.H * (unsigned)a.W * 4u;
    // Cout = 32 n + 16 (the ResNet's 64 -> 16 data gradient, resnet.py:56): the last slab's upper 16 rows multiply
    // zero weights and have no channel to land in - their offsets are out of range (wave-uniform test)
    const bool tail16 = cot * W4_CO + 16 >= a.Cout;
    asm volatile("s_nop 15");  // the last MFMAs' results (inline asm: no compiler-inserted wait states)
    // The residual and the partner's partial sums come through descriptors that are empty (every offset out of
    // range: zeros, no memory access) when that operand does not exist: no per-load branches.
    const bool use_res = has_res && !partial;
    const __amdgpu_buffer_rsrc_t rrs_e = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_res ? a.residual : a.y), (short)0, use_res ? (int)ybytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars_e = __builtin_amdgcn_make_buffer_rsrc(a.y, (short)0, accum ? (int)ybytes : 0, 0x00020000);
    // LOADS FIRST, STORES AFTER.  vmcnt retires in order, so a residual load issued behind an earlier channel pair's
    // stores is only known to have landed when those stores have been acknowledged by the memory side: round 3
    // loaded pair cr + 1 next to the stores of pair cr and every pair waited a write round trip (5.5 k of the
    // epilogue's 10 k cycles, tools/wino4_trace.py; tiles on the right image edge loaded and stored element by
    // element, 72 dependent round trips per item for those workgroups).  Now every load of the item goes out before
    // its first store - 16-byte loads for the edge tiles too (the bytes behind the row's end are the next row's, or
    // out of range = zero, and are not used) - and the stores that follow wait for nothing.  96 registers: V[], raw[]
    // and the operand registers are dead here.  The branches are wave-uniform; a launch without residual and
    // without a cut item loads nothing.
    unsigned lrow[MH];
#pragma unroll
    for (int yy = 0; yy < MH; ++yy)
      lrow[yy] = (valid && ho + yy < a.H) ? obase + (unsigned)(yy * a.W) * 4u : W4_OOB;
    // statistics records of this wave's tile group (see W4Args::stats)
    const bool do_stats = a.stats != nullptr && a.stats_mode == 1 && !partial;
    const bool do_bst = a.stats != nullptr && a.stats_mode == 2 && !partial;
    const bool st_ok = valid;
    const int st_G = 4 * a.nquad;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        a.stats, (short)0, (do_stats || do_bst) ? (int)(16u + (unsigned)a.Cout * (unsigned)st_G * 16u) : 0, 0x00020000);
    const unsigned st_chan = (unsigned)st_G * 16u;  // bytes between the records of consecutive channels
    const unsigned st_base = 16u + ((unsigned)co0 * (unsigned)st_G + (unsigned)g) * 16u;
    float st_cnt = 0.0f;
    bool all_in = false;
    if (do_stats || do_bst) {
      int n_in = 0;
#pragma unroll
      for (int yy = 0; yy < MH; ++yy)
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) n_in += (st_ok && ho + yy < a.H && wo + xx < a.W) ? 1 : 0;
      all_in = __builtin_amdgcn_ballot_w64(n_in != MH * 4) == 0;  // wave-uniform: no masks needed below
      float c = (float)n_in;
      st_cnt = rowsum16(c);
    }
    if constexpr (BST) {
      // Data-gradient launch that also takes the BatchNorm-backward sums (W4Args::stats_mode 2).  Per channel pair:
      // load the residual / partner sums, the BatchNorm input at the same positions and the channel's constants
      // (one pair ahead), form dA and the two sums - and keep dA in registers: NOTHING is stored until every load of
      // the item has been issued (a load behind a store waits for the store's acknowledgement, see below), then the
      // 8 x MH stores go out back to back.  96 registers of dA + two pairs of operands in flight; the instances
      // without the sums keep the round-3 register budget.
      const __amdgpu_buffer_rsrc_t xbrs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(do_bst ? a.bnx : a.y), (short)0, do_bst ? (int)ybytes : 0, 0x00020000);
      if (do_bst && !(use_res || accum)) {
        // The common case - no residual, not the lower half of a cut item (every eligible launch of the ResNet's
        // backward pass but those halves): the 8 x MH loads of the BatchNorm input and the 32 channel constants all
        // go out first (the residual's 96 registers are free), then each channel pair is formed, summed and STORED
        // at once - no load follows a store.  (With one pair in flight the loads' latency was exposed eight times
        // per item and the stores went out as one burst: +45 us per launch on layer1.)
        f32x4 xa[8][MH];
        float kmu[8], kis[8], ksc[8], ksh[8];
#pragma unroll
        for (int cr = 0; cr < 8; ++cr) {
          const unsigned coff = (unsigned)((cr >> 2) * 16 + (cr & 3)) * chan;
          const bool gone = (cr >> 2) == 1 && tail16;
          int c = co0 + (cr >> 2) * 16 + (cr & 3);
          c = c < a.Cout ? c : a.Cout - 1;
          kmu[cr] = a.bn_mean[c]; kis[cr] = a.bn_invstd[c]; ksc[cr] = a.bn_gamma[c]; ksh[cr] = a.bn_beta[c];
#pragma unroll
          for (int yy = 0; yy < MH; ++yy)
            xa[cr][yy] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xbrs, gone ? W4_OOB : lrow[yy] + coff, 0, 0));
        }
#pragma unroll
        for (int cr = 0; cr < 8; ++cr) {
          const int cb = cr >> 2, r = cr & 3;
          const unsigned coff = (unsigned)(cb * 16 + r) * chan;
          const bool gone = cb == 1 && tail16;
          const float mu = kmu[cr], is = kis[cr];
          const float sc = ksc[cr] * is;
          const float shf = ksh[cr] - mu * sc;
          float T[MH][6];
#pragma unroll
          for (int jj = 0; jj < 6; ++jj) {
            float m[NPR];
#pragma unroll
            for (int i = 0; i < NPR; ++i) {
              const int acc = cb * NP + 6 * i + jj;
              m[i] = acc < 64 ? accA[acc < 64 ? acc : 0][r] : accV[acc >= 64 ? acc - 64 : 0][r];
            }
            if constexpr (MH == 4) w4_at6(m[0], m[1], m[2], m[3], m[4], m[5], T[0][jj], T[1][jj], T[2][jj], T[3][jj]);
            else w4_at5(m[0], m[1], m[2], m[3], m[4], T[0][jj], T[1][jj], T[2][jj]);
          }
          float s1 = 0.0f, s2 = 0.0f;
          f32x4 Yv[MH];
#pragma unroll
          for (int yy = 0; yy < MH; ++yy) {
            float v0, v1, v2, v3;
            w4_at6(T[yy][0], T[yy][1], T[yy][2], T[yy][3], T[yy][4], T[yy][5], v0, v1, v2, v3);
            Yv[yy] = (f32x4){v0, v1, v2, v3};
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
              const float xe = xa[cr][yy][xx];
              float gg = Yv[yy][xx];
              if (!(xe * sc + shf > 0.0f)) gg = 0.0f;  // (the mask of bn_bwd_partial / bn_bwd_apply, verbatim)
              if (!all_in) gg = (st_ok && ho + yy < a.H && wo + xx < a.W) ? gg : 0.0f;
              const float xh = (xe - mu) * is;
              s1 += gg;
              s2 += gg * xh;
            }
          }
          s1 = rowsum16(s1);
          s2 = rowsum16(s2);
          const unsigned so = (jl == 0 && !gone) ? st_base + (unsigned)(cb * 16 + r) * st_chan : W4_OOB;
          const f32x4 rec = {s1, s2, st_cnt, 0.0f};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rec), srs, so, 0, 0);
#pragma unroll
          for (int yy = 0; yy < MH; ++yy)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Yv[yy]), yrs, gone ? W4_OOB : orow[yy] + coff, 0, 0);
          if (part) {
#pragma unroll
            for (int yy = 0; yy < MH; ++yy)
#pragma unroll
              for (int xx = 0; xx < 3; ++xx) {
                const unsigned o = (ho + yy < a.H && wo + xx < a.W && !gone) ? obase + (unsigned)(yy * a.W + xx) * 4u + coff : W4_OOB;
                const float v = Yv[yy][xx];  // (bit_cast of the element expression itself reads element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, o, 0, 0);
              }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        return;
      }
      f32x4 Yall[8][MH];
      f32x4 xn[MH], rn[MH];
      float cmu = 0.0f, cis = 0.0f, cga = 0.0f, cbe = 0.0f;
      auto fetch = [&](int cr) {
        const unsigned coff = (unsigned)((cr >> 2) * 16 + (cr & 3)) * chan;
        const bool gone = (cr >> 2) == 1 && tail16;
        int c = co0 + (cr >> 2) * 16 + (cr & 3);
        c = c < a.Cout ? c : a.Cout - 1;
        cmu = a.bn_mean[c]; cis = a.bn_invstd[c]; cga = a.bn_gamma[c]; cbe = a.bn_beta[c];
#pragma unroll
        for (int yy = 0; yy < MH; ++yy) {
          const unsigned o = gone ? W4_OOB : lrow[yy] + coff;
          xn[yy] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xbrs, o, 0, 0));
          rn[yy] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs_e, o, 0, 0)) +
                   __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars_e, o, 0, W4_SC1));
        }
      };
      fetch(0);
#pragma unroll
      for (int cr = 0; cr < 8; ++cr) {
        const int cb = cr >> 2, r = cr & 3;
        const bool gone = cb == 1 && tail16;
        f32x4 xc[MH], rc[MH];
#pragma unroll
        for (int yy = 0; yy < MH; ++yy) { xc[yy] = xn[yy]; rc[yy] = rn[yy]; }
        const float mu = cmu, is = cis;
        const float sc = cga * is;
        const float shf = cbe - mu * sc;
        if (cr + 1 < 8) fetch(cr + 1);
        float T[MH][6];
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
          float m[NPR];
#pragma unroll
          for (int i = 0; i < NPR; ++i) {
            const int acc = cb * NP + 6 * i + jj;
            m[i] = acc < 64 ? accA[acc < 64 ? acc : 0][r] : accV[acc >= 64 ? acc - 64 : 0][r];
          }
          if constexpr (MH == 4) w4_at6(m[0], m[1], m[2], m[3], m[4], m[5], T[0][jj], T[1][jj], T[2][jj], T[3][jj]);
          else w4_at5(m[0], m[1], m[2], m[3], m[4], T[0][jj], T[1][jj], T[2][jj]);
        }
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int yy = 0; yy < MH; ++yy) {
          float v0, v1, v2, v3;
          w4_at6(T[yy][0], T[yy][1], T[yy][2], T[yy][3], T[yy][4], T[yy][5], v0, v1, v2, v3);
          const f32x4 Yv = (f32x4){v0, v1, v2, v3} + rc[yy];
          Yall[cr][yy] = Yv;
          if (do_bst) {
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
              const float xe = xc[yy][xx];
              float gg = Yv[xx];
              if (!(xe * sc + shf > 0.0f)) gg = 0.0f;  // (the mask of bn_bwd_partial / bn_bwd_apply, verbatim)
              if (!all_in) gg = (st_ok && ho + yy < a.H && wo + xx < a.W) ? gg : 0.0f;
              const float xh = (xe - mu) * is;
              s1 += gg;
              s2 += gg * xh;
            }
          }
        }
        if (do_bst) {
          s1 = rowsum16(s1);
          s2 = rowsum16(s2);
          // (records are 16-byte stores as well - but nothing is loaded behind them: the operands of pair cr + 1
          // were requested above)
          const unsigned so = (jl == 0 && !gone) ? st_base + (unsigned)(cb * 16 + r) * st_chan : W4_OOB;
          const f32x4 rec = {s1, s2, st_cnt, 0.0f};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rec), srs, so, 0, 0);
        }
      }
#pragma unroll
      for (int cr = 0; cr < 8; ++cr) {
        const int cb = cr >> 2, r = cr & 3;
        const unsigned coff = (unsigned)(cb * 16 + r) * chan;
        const bool gone = cb == 1 && tail16;
#pragma unroll
        for (int yy = 0; yy < MH; ++yy) {
          const unsigned o = gone ? W4_OOB : orow[yy] + coff;
          if (partial) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Yall[cr][yy]), yrs, o, 0, W4_SC1);
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Yall[cr][yy]), yrs, o, 0, 0);
        }
        if (part) {
#pragma unroll
          for (int yy = 0; yy < MH; ++yy)
#pragma unroll
            for (int xx = 0; xx < 3; ++xx) {
              const unsigned o = (ho + yy < a.H && wo + xx < a.W && !gone) ? obase + (unsigned)(yy * a.W + xx) * 4u + coff : W4_OOB;
              const float v = Yall[cr][yy][xx];
              if (partial) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, o, 0, W4_SC1);
              else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, o, 0, 0);
            }
        }
      }
      return;
    }
    f32x4 res[8][MH];
    if (use_res || accum) {
#pragma unroll
      for (int cr = 0; cr < 8; ++cr) {
        const unsigned coff = (unsigned)((cr >> 2) * 16 + (cr & 3)) * chan;
        const bool gone = (cr >> 2) == 1 && tail16;
#pragma unroll
        for (int yy = 0; yy < MH; ++yy) {
          const unsigned o = gone ? W4_OOB : lrow[yy] + coff;
          res[cr][yy] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs_e, o, 0, 0));
        }
      }
      if (accum) {  // (a cut item's lower half: rare - two batches of 4 x MH loads in flight)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 ps[4][MH];
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const int cr = 4 * h + c4;
            const unsigned coff = (unsigned)((cr >> 2) * 16 + (cr & 3)) * chan;
            const bool gone = (cr >> 2) == 1 && tail16;
#pragma unroll
            for (int yy = 0; yy < MH; ++yy)
              ps[c4][yy] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                  ars_e, gone ? W4_OOB : lrow[yy] + coff, 0, W4_SC1));
          }
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
            for (int yy = 0; yy < MH; ++yy) res[4 * h + c4][yy] += ps[c4][yy];
        }
      }
    } else {
#pragma unroll
      for (int cr = 0; cr < 8; ++cr)
#pragma unroll
        for (int yy = 0; yy < MH; ++yy) res[cr][yy] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int cr = 0; cr < 8; ++cr) {
      const int cb = cr >> 2, r = cr & 3;
      // (added to the per-lane offset: as the soffset operand of the buffer stores, a non-zero channel offset
      // gave wrong second dwords in lanes 12-15 of every row of 16 on gfx950 - not understood, avoided)
      const unsigned coff = (unsigned)(cb * 16 + r) * chan;
      const bool gone = cb == 1 && tail16;
      float T[MH][6];
#pragma unroll
      for (int jj = 0; jj < 6; ++jj) {
        float m[NPR];
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
          const int acc = cb * NP + 6 * i + jj;
          m[i] = acc < 64 ? accA[acc < 64 ? acc : 0][r] : accV[acc >= 64 ? acc - 64 : 0][r];
        }
        if constexpr (MH == 4) w4_at6(m[0], m[1], m[2], m[3], m[4], m[5], T[0][jj], T[1][jj], T[2][jj], T[3][jj]);
        else w4_at5(m[0], m[1], m[2], m[3], m[4], T[0][jj], T[1][jj], T[2][jj]);
      }
      f32x4 Y[MH];
#pragma unroll
      for (int yy = 0; yy < MH; ++yy) {
        float v0, v1, v2, v3;
        w4_at6(T[yy][0], T[yy][1], T[yy][2], T[yy][3], T[yy][4], T[yy][5], v0, v1, v2, v3);
        Y[yy] = (f32x4){v0, v1, v2, v3} + res[cr][yy];
      }
      if (do_stats) {  // wave-uniform: forward launch with a statistics buffer, not the upper half of a cut item
        // channel (cb, r) of this lane's row of 16 tiles; K = the row's first tile's first output (any finite value
        // serves as the shift; one near the data keeps (y - K)^2 from cancelling in the merge)
        const float K = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
            0, __builtin_bit_cast(int, Y[0][0]), 0x150 /* row_newbcast:0 */, 0xf, 0xf, false));
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int yy = 0; yy < MH; ++yy)
#pragma unroll
          for (int xx = 0; xx < 4; ++xx) {
            float d = Y[yy][xx] - K;
            if (!all_in) d = (st_ok && ho + yy < a.H && wo + xx < a.W) ? d : 0.0f;
            s1 += d;
            s2 = __builtin_fmaf(d, d, s2);
          }
        s1 = rowsum16(s1);
        s2 = rowsum16(s2);
        const unsigned so = (jl == 0 && !gone) ? st_base + (unsigned)(cb * 16 + r) * st_chan : W4_OOB;
        const f32x4 rec = {st_cnt, K, s1, s2};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rec), srs, so, 0, 0);
      }
      long long cs = 0;
      if (tracing) { __builtin_amdgcn_sched_barrier(0); cs = clock64(); }
#pragma unroll
      for (int yy = 0; yy < MH; ++yy) {
        const unsigned o = gone ? W4_OOB : orow[yy] + coff;
        if (partial) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Y[yy]), yrs, o, 0, W4_SC1);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Y[yy]), yrs, o, 0, 0);
      }
      if (tracing) { __builtin_amdgcn_sched_barrier(0); tS += clock64() - cs; }
      if (part) {  // lanes of the tile column that straddles the right edge: W % 4 element stores per row
#pragma unroll
        for (int yy = 0; yy < MH; ++yy)
#pragma unroll
          for (int xx = 0; xx < 3; ++xx) {
            const unsigned o = (ho + yy < a.H && wo + xx < a.W && !gone) ? obase + (unsigned)(yy * a.W + xx) * 4u + coff : W4_OOB;
            const float v = Y[yy][xx];
            if (partial) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, o, 0, W4_SC1);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, o, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: patch(0) | U(0) patch(1) | U(1) patch(2); wait for the first, transform it
  dma_patch();
  dma_u(); dma_patch();
  dma_u(); dma_patch();
  w4_wait<2 * ND>();
  __syncthreads();
  lane_consts();
  int e_cur = edge_of(i0), e_nxt = -1;
  if (e_cur >= 0) fix_edge(ldsP, e_cur);
  read_patch(ldsP);
  transform(std::false_type{});

  int cur = 0;
  long long tW = 0, tB = 0, tD = 0, tM = 0, tT = 0, tE = 0;
  // FIRST: the item's first k-step (accumulators start from zero).  LAST: its last one - the patch of the NEXT
  // item's first step is not read and transformed here but behind the epilogue (next_patch below): V[] and
  // raw[] are then dead across the epilogue, which otherwise spills them to scratch and brings them back inside
  // the next k-step behind `s_waitcnt vmcnt(0)` - draining the DMA queue once per item (round 3: 10 % on layer1,
  // whose items are 16 k-steps long).
  auto kstep = [&](auto first_tag, auto last_tag, int e_next) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;
    const int nxt = cur + 1 == W4_NBUF ? 0 : cur + 1;
    long long c0_ = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    if (tracing) c0_ = clock64();
    // U(s) and patch(s+1) have landed (loads retire in order; the group issued last step may still fly).
    // (Round 4, measured and dropped: the first AHEAD operand reads of step s + 1 issued at the END of step s, across
    // the barrier - needs U(s+1) landed one step earlier, vmcnt(NI) here - layer3 0.281 -> 0.285 ms: the LDS round
    // trip behind the barrier is not what the k-step waits for.)
    w4_wait<ND>();
    if (tracing) c1 = clock64();
    __syncthreads();
    if (tracing) c2 = clock64();
    if (tracing) c3 = clock64();
    float* pn = ldsP + nxt * W4_PATCHF;
    const float* ub = ldsU + cur * ULDS + ub_lane;
    constexpr int NG = 2 * NQ;  // operand groups: row block g / NQ, position quad g % NQ
    f32x4 u[NG];
    constexpr int AHEAD = 3;
    auto ldu = [&](int g) {
      const int rb = g / NQ, q = g % NQ;
      u[g] = *reinterpret_cast<const f32x4*>(ub + P::uoff(0, 16 * rb, 4 * q));
    };
#if W4_PATCH_FIRST
    if (e_next >= 0) fix_edge(pn, e_next);
    if constexpr (!LAST) read_patch(pn);
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) ldu(g);
#else
    // the first operand reads go out BEFORE the 3 NPR patch reads: LDS returns in order, and the first MFMA then
    // waits for its own operand only while the patch (needed by the transform at the end) lands under the MFMAs
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) ldu(g);
    __builtin_amdgcn_sched_barrier(0);
    if (e_next >= 0) fix_edge(pn, e_next);
    if constexpr (!LAST) read_patch(pn);
#endif
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + AHEAD < NG) ldu(g + AHEAD);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = 4 * (g % NQ) + q, acc = (g / NQ) * NP + p;
        if (p < NP) {
          if (acc < 64) {
            if (FIRST) W4_MFMA_A0(accA[acc < 64 ? acc : 0], u[g][q], V[p < NP ? p : 0]);
            else W4_MFMA_A(accA[acc < 64 ? acc : 0], u[g][q], V[p < NP ? p : 0]);
          } else {
            if (FIRST) W4_MFMA_V0(accV[acc >= 64 ? acc - 64 : 0], u[g][q], V[p < NP ? p : 0]);
            else W4_MFMA_V(accV[acc >= 64 ? acc - 64 : 0], u[g][q], V[p < NP ? p : 0]);
          }
        }
      }
      // restaging, one DMA per group boundary so the memory pipeline takes them one at a time (issued back
      // to back by all four waves, 52 KB queue up in front of it and every issue stalls): U(s+2) into the
      // buffer U(s-1) left, patch(s+3) into the buffer patch(s) left
#if W4_EXP_RETILE > 1
      // ND units spread evenly over the NG group boundaries
#define W4_FLAT(K_) if ((K_) * NG / ND == g || (ND > NG && ((K_) * NG) / ND == g)) W4_UNIT(K_)
      W4_FLAT(0); W4_FLAT(1); W4_FLAT(2); W4_FLAT(3); W4_FLAT(4); W4_FLAT(5); W4_FLAT(6); W4_FLAT(7); W4_FLAT(8);
      W4_FLAT(9); W4_FLAT(10); W4_FLAT(11); W4_FLAT(12); W4_FLAT(13); W4_FLAT(14); W4_FLAT(15); W4_FLAT(16);
      W4_FLAT(17); W4_FLAT(18); W4_FLAT(19); W4_FLAT(20); W4_FLAT(21); W4_FLAT(22); W4_FLAT(23);
#undef W4_FLAT
#elif W4_DMA_PLACE == 0
      switch (g) {
        case 0: W4_UNIT(0); break;
        case 1: W4_UNIT(1); break;
        case 2: W4_UNIT(2); break;
        case 3: W4_UNIT(3); break;
        case 4: W4_UNIT(4); break;
        case 5: W4_UNIT(5); break;
        case 6: W4_UNIT(6); break;
        case 7: W4_UNIT(7); break;
        case 8: W4_UNIT(8); break;
        case 9: W4_UNIT(9); break;
        case 10: W4_UNIT(10); break;
        case 11: W4_UNIT(11); break;
        case 12: W4_UNIT(12); break;
        default: break;
      }
#elif W4_DMA_PLACE == 2
      switch (g) {
        case 0: W4_UNIT(0); W4_UNIT(1); break;
        case 1: W4_UNIT(2); W4_UNIT(3); break;
        case 2: W4_UNIT(4); W4_UNIT(5); break;
        case 3: W4_UNIT(6); W4_UNIT(7); break;
        case 4: W4_UNIT(8); W4_UNIT(9); break;
        case 5: W4_UNIT(10); W4_UNIT(11); break;
        case 6: W4_UNIT(12); break;
        default: break;
      }
#else
      if (LAST) {  // no transform in this k-step: restage here
        switch (g) {
          case 0: W4_UNIT(0); break; case 1: W4_UNIT(1); break; case 2: W4_UNIT(2); break; case 3: W4_UNIT(3); break;
          case 4: W4_UNIT(4); break; case 5: W4_UNIT(5); break; case 6: W4_UNIT(6); break; case 7: W4_UNIT(7); break;
          case 8: W4_UNIT(8); break; case 9: W4_UNIT(9); break; case 10: W4_UNIT(10); break; case 11: W4_UNIT(11); break;
          case 12: W4_UNIT(12); break; default: break;
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    // the last MFMAs write (MH = 4) the VGPR-resident accumulators: inline asm gets no compiler-inserted wait
    // states, and a register copy the allocator places right behind them would read half-written results
    asm volatile("s_nop 15");
#if W4_DMA_PLACE == 1
    if (tracing) c4 = clock64();
    if constexpr (!LAST) transform(std::true_type{});
    adv_u();
    adv_patch();
#else
    adv_u();
    adv_patch();
    if (tracing) c4 = clock64();
#ifndef W4_EXP_NOXF
    if constexpr (!LAST) transform(std::false_type{});
#else
    if constexpr (!LAST) {
#pragma unroll
      for (int p = 0; p < NP; ++p) V[p] = raw[p];
    }
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (tracing) {
      c5 = clock64();
      tW += c1 - c0_; tB += c2 - c1; tD += c3 - c2; tM += c4 - c3; tT += c5 - c4;
    }
    cur = nxt;
  };
  for (int seg = 0; seg < nseg; ++seg) {
    const int item = seg_item(seg);
    const int len = seg_end(seg) - (seg == 0 ? c0 : 0);
    e_nxt = seg + 1 < nseg ? edge_of(seg_item(seg + 1)) : -1;
    long long cl = 0;
    if (tracing) cl = clock64();
    if (len == 1) {
      kstep(std::true_type{}, std::true_type{}, e_nxt);
    } else {
      kstep(std::true_type{}, std::false_type{}, e_cur);
      for (int chunk = 1; chunk < len - 1; ++chunk) kstep(std::false_type{}, std::false_type{}, e_cur);
      kstep(std::false_type{}, std::true_type{}, e_nxt);
    }
    e_cur = e_nxt;
    long long ce = 0;
    if (tracing) { ce = clock64(); tL += ce - cl; }
    const bool upper_part = is_upper && seg == tail_seg;        // -> partial sums into y
    const bool lower_part = is_lower && seg == tail_seg;        // <- the partner's partial sums
    if (lower_part) {
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(a.flags + flag_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
          __builtin_amdgcn_s_sleep(8);
          // the partner never ran (it publishes before anything else): fail the launch loudly instead of adding
          // sums that are not there
          if (++spins == (1u << 26)) __builtin_trap();
        }
        // consumed: back to zero, so that a REPLAY of this launch from a captured hipGraph (same epoch, same
        // flag word) waits for its own upper half again instead of finding last replay's flag
        __hip_atomic_store(a.flags + flag_idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
    epilogue(item, upper_part, lower_part);
    if (tracing) tE += clock64() - ce;
    // Compiler-visible vmcnt(0): whatever it spilled around the epilogue has come back, so it puts no
    // vmcnt waits (which would also drain the DMAs in flight) into the k-step loop; the tail sums are out.
    long long cd = 0;
    if (tracing) cd = clock64();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (tracing) tDr += clock64() - cd;
    if (seg + 1 < nseg) {  // the next item's first patch (buffer `cur` after the last k-step's swap; its right edge
      lane_consts();       // was fixed there): the reads and the transform the LAST k-step left out
      read_patch(ldsP + cur * W4_PATCHF);
      transform(std::false_type{});
    }
    if (upper_part) {  // publish: every wave's sc1 stores have been acknowledged (vmcnt(0) above), then the flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.flags + flag_idx, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tracing && (tid & 63) == 0) {
    long long* o = a.trace + wave * 8;
    o[0] = tW; o[1] = tB; o[2] = tD; o[3] = tM; o[4] = tT; o[5] = tE; o[6] = nseg; o[7] = S; o[32] = tS; o[33] = clock64() - tk0; o[34] = wall_clock64() - tw0; o[35] = tDr; o[36] = tL;
  }
}

int w4_grid_for(size_t n) {
  size_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

// output-tile height for an H-row image: the one that issues fewer positions (ties: 4)
int w4_tile_rows(int H) {
  const int opt = air_opt(AIR_OPT_WINO4_TH3);  // 0: always 4; 1: fewer positions, ties -> 4; 2: ties -> 3
  if (!opt) return 4;
  const int c4 = (H + 3) / 4 * 36, c3 = (H + 2) / 3 * 30;
  return (c3 < c4 || (c3 == c4 && opt == 2)) ? 3 : 4;
}

template <int MH>
int w4_launch(W4Args& a, int trg, int nblk, hipStream_t st) {
  // (sized for the 2 x 8 groups' planes, the larger of the two layouts: one attribute for all four instances)
  const size_t ldsb = (size_t)W4_NBUF * (W4Pos<MH>::ULDS + W4Cfg<2, MH>::PATCHF) * sizeof(float);
  static_assert(W4Cfg<2, MH>::PATCHF >= W4Cfg<1, MH>::PATCHF, "the 2 x 8 layout is the larger one");
  static const bool attr_ok = [=] {  // > 64 KB of dynamic LDS needs the opt-in, once per kernel
    const void* ks[6] = {reinterpret_cast<const void*>(wino4_conv_kernel<1, MH, false>),
                         reinterpret_cast<const void*>(wino4_conv_kernel<2, MH, false>),
                         reinterpret_cast<const void*>(wino4_conv_kernel<1, MH, true>),
                         reinterpret_cast<const void*>(wino4_conv_kernel<2, MH, true>),
                         reinterpret_cast<const void*>(wino4_conv_kernel<1, MH, false, true>),
                         reinterpret_cast<const void*>(wino4_conv_kernel<2, MH, false, true>)};
    bool ok = true;
    for (int i = 0; i < 6; ++i)
      ok = ok && hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) == hipSuccess;
    return ok;
  }();
  if (!attr_ok) return AIR_ELAUNCH;
  if (a.stats_mode == 2) {  // data gradient + BatchNorm-backward sums: its own instances (deferred stores)
    if (trg == 2) hipLaunchKernelGGL((wino4_conv_kernel<2, MH, false, true>), dim3(nblk), dim3(256), ldsb, st, a);
    else hipLaunchKernelGGL((wino4_conv_kernel<1, MH, false, true>), dim3(nblk), dim3(256), ldsb, st, a);
  } else if (a.trace != nullptr) {
    if (trg == 2) hipLaunchKernelGGL((wino4_conv_kernel<2, MH, true>), dim3(nblk), dim3(256), ldsb, st, a);
    else hipLaunchKernelGGL((wino4_conv_kernel<1, MH, true>), dim3(nblk), dim3(256), ldsb, st, a);
  } else if (trg == 2) {
    hipLaunchKernelGGL((wino4_conv_kernel<2, MH, false>), dim3(nblk), dim3(256), ldsb, st, a);
  } else {
    hipLaunchKernelGGL((wino4_conv_kernel<1, MH, false>), dim3(nblk), dim3(256), ldsb, st, a);
  }
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

// workgroups of wino4_conv_kernel<*, MH> a CU holds (occupancy query, cached per tile height)
int w4_resident_per_cu(int mh) {
  static std::atomic<int> cached[2] = {{-1}, {-1}};
  std::atomic<int>& c = cached[mh == 3 ? 0 : 1];
  int v = c.load(std::memory_order_relaxed);
  if (v >= 0) return v;
  int n = 0;
  hipError_t e;
  if (mh == 3) {
    const size_t ldsb = (size_t)W4_NBUF * (W4Pos<3>::ULDS + W4Cfg<2, 3>::PATCHF) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_conv_kernel<1, 3, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wino4_conv_kernel<1, 3, false>, 256, ldsb);
  } else {
    const size_t ldsb = (size_t)W4_NBUF * (W4Pos<4>::ULDS + W4Cfg<2, 4>::PATCHF) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_conv_kernel<1, 4, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wino4_conv_kernel<1, 4, false>, 256, ldsb);
  }
  if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
  c.store(n, std::memory_order_relaxed);
  return n;
}

}  // namespace

static long long* g_wino4_trace = nullptr;
extern "C" void air_dbg_wino4_trace(long long* p) { g_wino4_trace = p; }

bool air_wino4_ok(int B, int Kc, int H, int W, int M) {
  if (air_opt(AIR_OPT_NO_WINO4) || (air_opt(AIR_OPT_NO_WINOGRAD) & 1)) return false;
  if (M < 16 || M % 16 != 0 || Kc < W4_CK || Kc % W4_CK != 0) return false;  // (M = 32 n + 16: a half-used last slab)
  // buffer-descriptor staging: byte offsets stay below the out-of-range marker (2 GiB)
  const double ein = (double)B * Kc * H * W, eout = (double)B * M * H * W;
  return ein * 4.0 + 8192.0 < 2147483648.0 && eout * 4.0 + 8192.0 < 2147483648.0 &&
         (double)air_wino4_packed_elems(M, Kc) * 4.0 < 4294967296.0 && H >= 1 && W >= 4;
}

// (either tile height: 36 floats per (co, ci) cover the 32 of the 3-row layout)
size_t air_wino4_packed_elems(int M, int Kc) { return (size_t)((M + W4_CO - 1) / W4_CO * W4_CO) * Kc * 36 + 1024; }

// deferred transforms of this thread (air_wino4_weights_defer): recorded here, run by air_wino4_weights_flush
namespace {
thread_local bool g_w4_defer = false;
thread_local W4WeightJobs g_w4_jobs;
int w4_run_jobs(hipStream_t st) {
  if (g_w4_jobs.n == 0) return AIR_OK;
  hipLaunchKernelGGL(wino4_weights_batch_kernel, dim3(g_w4_jobs.blk0[g_w4_jobs.n]), dim3(256), 0, st, g_w4_jobs);
  g_w4_jobs.n = 0;
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}
}  // namespace

void air_wino4_weights_defer(bool on) {
  g_w4_defer = on;
  g_w4_jobs.n = 0;
  g_w4_jobs.blk0[0] = 0;
}
int air_wino4_weights_flush(hipStream_t st) { return w4_run_jobs(st); }

int air_wino4_weights(const float* w, float* up, int M, int Kc, int H, int dgrad, hipStream_t st, bool may_defer) {
  const size_t n = (size_t)((M + W4_CO - 1) / W4_CO * W4_CO) * Kc;
  if (g_w4_defer && may_defer) {
    if (g_w4_jobs.n == W4_WJOBS) {
      const int rc = w4_run_jobs(st);
      if (rc != AIR_OK) return rc;
      g_w4_jobs.blk0[0] = 0;
    }
    const int j = g_w4_jobs.n++;
    g_w4_jobs.w[j] = w; g_w4_jobs.up[j] = up; g_w4_jobs.M[j] = M; g_w4_jobs.Kc[j] = Kc;
    g_w4_jobs.dgrad[j] = (unsigned char)dgrad; g_w4_jobs.mh[j] = (unsigned char)w4_tile_rows(H);
    g_w4_jobs.blk0[j + 1] = g_w4_jobs.blk0[j] + w4_grid_for(n);
    return AIR_OK;
  }
  if (w4_tile_rows(H) == 3)
    hipLaunchKernelGGL(wino4_weights_kernel<3>, dim3(w4_grid_for(n)), dim3(256), 0, st, w, up, M, Kc, dgrad);
  else
    hipLaunchKernelGGL(wino4_weights_kernel<4>, dim3(w4_grid_for(n)), dim3(256), 0, st, w, up, M, Kc, dgrad);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

// cut-item flags: per device, 64 launches' worth of 256 words, zeroed once; epochs never repeat
static unsigned* w4_flag_ring(hipStream_t st) {
  static std::mutex mu;
  static unsigned* rings[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (rings[dev] == nullptr) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    unsigned* p = nullptr;
    if (hipMalloc(&p, 64 * 256 * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 64 * 256 * sizeof(unsigned)) != hipSuccess) return nullptr;
    rings[dev] = p;
  }
  return rings[dev];
}

// tile geometry of a launch (shared by the launch and the statistics-buffer size query); returns TRG
static int w4_geometry(W4Args& a, int B, int H, int W, int M, int mh) {
  a.B = B; a.H = H; a.W = W; a.Cout = M;
  a.TH = (H + mh - 1) / mh; a.TW = (W + 3) / 4;
  a.SR = B * a.TH;
  // 1 x 16 or 2 x 8 tiles per group: fewer groups = less padding waste
  const long g1 = (long)a.SR * ((a.TW + 15) / 16), g2 = (long)((a.SR + 1) / 2) * ((a.TW + 7) / 8);
  const int trg = g2 < g1 ? 2 : 1;
  a.GRR = (a.SR + trg - 1) / trg;
  a.TWG = (a.TW + 16 / trg - 1) / (16 / trg);
  a.ngroups = a.GRR * a.TWG;
  a.ncot = (M + W4_CO - 1) / W4_CO;
  a.nquad = (a.ngroups + 3) / 4;
  return trg;
}

size_t air_wino4_stats_bytes(int B, int H, int W, int M) {
  W4Args a;
  w4_geometry(a, B, H, W, M, w4_tile_rows(H));
  return 16 + (size_t)M * (size_t)(4 * a.nquad) * 16;
}

int air_wino4_conv(const float* x, const float* w, float* y, const float* residual, int B, int Kc, int H,
                   int W, int M, int dgrad, float* up, double flops, hipStream_t st, float* stats,
                   const float* const* bn) {
  if (w != nullptr) {  // w == nullptr: `up` already holds the transformed weights (air_wino4_weights)
    const int rc = air_wino4_weights(w, up, M, Kc, H, dgrad, st, false);  // consumed by the launch below: never deferred
    if (rc != AIR_OK) return rc;
  }
  // forward launches take forward statistics (bn == nullptr), data-gradient launches BatchNorm-backward sums
  // (bn = {bnx, mean, invstd, gamma, beta})
  if (stats != nullptr && ((dgrad != 0) != (bn != nullptr) || (reinterpret_cast<size_t>(stats) & 15) ||
                           air_wino4_stats_bytes(B, H, W, M) >= 0x80000000ull))
    return AIR_EINVAL;
  if (stats == nullptr && bn != nullptr) return AIR_EINVAL;
  const int mh = w4_tile_rows(H);
  W4Args a;
  a.x = x; a.up = up; a.y = y; a.residual = residual; a.stats = stats;
  a.stats_mode = stats == nullptr ? 0 : (bn != nullptr ? 2 : 1);
  a.bnx = bn ? bn[0] : nullptr; a.bn_mean = bn ? bn[1] : nullptr; a.bn_invstd = bn ? bn[2] : nullptr;
  a.bn_gamma = bn ? bn[3] : nullptr; a.bn_beta = bn ? bn[4] : nullptr;
  a.Cin = Kc;
  const int trg = w4_geometry(a, B, H, W, M, mh);
  a.cotb = a.ncot < 8 ? a.ncot : 8;
  while (a.ncot % a.cotb != 0) --a.cotb;  // (channel counts here are powers of two times 32: a no-op)
  a.nitems = a.nquad * a.ncot;
  a.trace = g_wino4_trace;
  a.dephase = air_opt(AIR_OPT_WINO4_DEPHASE);
  // one persistent workgroup per CU the stream can use (144 KB of LDS and 4 x 512 registers: one fits)
  const int ncu = air_stream_cus(st);
  const int nblk = a.nitems < ncu ? a.nitems : ncu;
  // dealing groups = XCDs (workgroup b runs on XCD b % 8); WINO4_XCD = 0: one group, i.e. round-robin over
  // the whole chip (for A/B measurements of the locality)
  a.xmode = air_opt(AIR_OPT_WINO4_XCD) == 2 ? 2 : 1;
  a.nxg = air_opt(AIR_OPT_WINO4_XCD) ? (nblk < 8 ? nblk : 8) : 1;
  if (a.nquad < a.nxg) a.xmode = 1;  // (xmode 2 wants at least one quad per group)
  // cut items.  Under stream capture the per-launch epoch and flag slot are frozen into the graph: the reader of
  // a flag resets it, so every replay starts from zero; the ring must exist before the capture (hipMalloc)
  a.split = 0; a.flags = nullptr; a.epoch = 0;
  // Only when every workgroup of the launch is resident at once (nblk <= the CUs behind this stream, which holds
  // by construction above unless the occupancy query says a workgroup does not fit a CU at all): an owner that
  // spins on a publisher still waiting for a CU would otherwise hold that CU from it.
  if (air_opt(AIR_OPT_WINO4_SPLIT) && w4_resident_per_cu(mh) >= 1) {
    unsigned* flag_ring = w4_flag_ring(st);
    static std::atomic<unsigned> next_epoch{1};
    if (flag_ring != nullptr) {
      a.epoch = next_epoch.fetch_add(1);
      if (a.epoch == 0) a.epoch = next_epoch.fetch_add(1);
      a.flags = flag_ring + (a.epoch % 64) * 256;
      a.split = 1;
    }
  }
  // MFMA FLOPs the launch issues: every item runs Cin / 4 k-steps of NACC MFMAs (16 x 16 x 4) in 4 waves
  const double issued = (double)a.nitems * (Kc / W4_CK) * 4.0 * (mh == 3 ? W4Pos<3>::NACC : W4Pos<4>::NACC) * 2048.0;
  // algorithmic bytes: x and the transformed weights read once, y written once, the residual read once
  const double abytes = 4.0 * ((double)B * Kc * H * W + (double)B * M * H * W * (residual ? 2.0 : 1.0) +
                               (double)M * Kc * (mh == 3 ? 32.0 : 36.0));
  // (launches that also do a BatchNorm reduction's work are timed under their own id: bench.py reports the family's
  // issued-FLOP fraction over all launches and the plain launches' beside it)
  AirProfScope ps(stats ? AIR_K_CONV_WINO4_BN : AIR_K_CONV_WINO4, flops, st, issued,
                  abytes + (bn ? 4.0 * (double)B * M * H * W : 0.0));
  return mh == 3 ? w4_launch<3>(a, trg, nblk, st) : w4_launch<4>(a, trg, nblk, st);
}
