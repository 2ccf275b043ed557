// conv2d forward / dgrad / wgrad for gfx950 as implicit GEMMs on the exact-f32
// MFMA (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TFLOP/s peak).
//
// Replaces nn.Conv2d (bias=False) as used by resnet.py:131 (conv1), :56-61
// (PreActBlock 3x3 / 1x1 shortcut) and :140 (conv5).  NCHW fp32.
//
// Forward GEMM:  D[co][px] = sum_k Wp[k][co] * Patch[k][px],  k = (tap, ci).
//   * M = output channels: a workgroup owns 64 of them (2 MFMA row tiles).
//   * N = output pixels in "pixel tiles" of 32 consecutive columns of one
//     output row; pixel tiles are numbered linearly over (b, ho, wo/32) so
//     ragged widths (750, 375, 188, 94) waste <= 2.4 %.  One wave per pixel
//     tile, 4 waves per workgroup; MFMA D columns = pixels, so stores are
//     128-byte coalesced rows of NCHW.
//   * K is walked in chunks of 8 input channels.  Per chunk the workgroup
//     stages (a) the packed weight slab [tap][ci][64 co] (shared by its 4
//     waves) and (b) per wave the input patch [ci][KH rows][31*S+KW cols] in
//     LDS, so each staged input element feeds all KH*KW taps and all 64
//     output channels: global/L2 traffic is ~1/9 of an im2col GEMM.
//   * LDS is double buffered; the next chunk's global loads are issued before
//     the current chunk's 72 MFMAs and written to LDS after them.
//   * Optional fused prologue while staging: y = max(0, x*scale[ci]+shift[ci])
//     (the BatchNorm-apply + ReLU in front of every PreActBlock conv), with
//     zero padding applied AFTER the activation like F.conv2d on the
//     activated tensor; optional epilogue residual add (resnet.py:68).
// dgrad (stride 1) is the same kernel on flipped/transposed packed weights;
// stride-2 dgrad first zero-upsamples dy.  wgrad is a second implicit GEMM
// with K = pixels, split over workgroups, reduced deterministically.
#include <stdlib.h>

#include <type_traits>

#include "air_common.h"
#include "air_options.h"
#include "air_prof.h"
#include "air_lds_dma.h"
#include "conv_wino.h"
#include "conv_bf3.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64;       // output channels per workgroup
constexpr int NWAVE = 4;     // waves (= pixel tiles) per workgroup
constexpr int PXT = 32;      // output pixels per pixel tile

// ------------------------------------------------------------------ packing
// forward:  Wp[cot][chunk][tap][cil][col] = W[cot*64+col][chunk*ck+cil][sel[tap]]
// dgrad:    roles swapped:               = W[chunk*ck+cil][cot*64+col][sel[tap]]
// sel lists the source taps (all taps in order for a forward conv, flipped for a stride-1
// dgrad, one parity class for a stride-2 dgrad).
struct TapSel {
  int n;
  int idx[9];
};

// (bid of nb blocks: the launch's own grid, or this slab's share of a batched launch)
__device__ __forceinline__ void pack_weights_body(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                                  int Cin, int taps_full, int transpose, int ck, int bm,
                                                  const TapSel& sel, int bid, int nb) {
  // logical (M = "out" role, Kc = "in" role)
  const int M = transpose ? Cin : Cout;
  const int Kc = transpose ? Cout : Cin;
  const int Mpad = (M + bm - 1) / bm * bm;  // channel tiles (bm = 64 or 32) are zero-padded
  const size_t total = (size_t)Mpad * ((Kc + ck - 1) / ck * ck) * sel.n;
  for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < total;
       e += (size_t)nb * blockDim.x) {
    const int col = (int)(e % bm);
    size_t r = e / bm;
    const int cil = (int)(r % ck);
    r /= ck;
    const int tap = (int)(r % sel.n);
    r /= sel.n;
    const int nchunk = (Kc + ck - 1) / ck;
    const int chunk = (int)(r % nchunk);
    const int cot = (int)(r / nchunk);
    // ragged Kc: the last chunk covers [Kc-ck, Kc); channels an earlier chunk already
    // covers get zero weights there
    const bool last_ragged = chunk == nchunk - 1 && Kc % ck != 0;
    const int k = last_ragged ? Kc - ck + cil : chunk * ck + cil;
    const int m = cot * bm + col;
    float v = 0.0f;
    if (m < M && !(last_ragged && k < chunk * ck)) {
      const int src = sel.idx[tap];
      v = transpose ? w[((size_t)k * Cin + m) * taps_full + src]
                    : w[((size_t)m * Cin + k) * taps_full + src];
    }
    wp[e] = v;
  }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                    int Cin, int taps_full, int transpose, int ck, int bm,
                                    TapSel sel) {
  pack_weights_body(w, wp, Cout, Cin, taps_full, transpose, ck, bm, sel, (int)blockIdx.x, (int)gridDim.x);
}

// the slabs of several layers in one launch (air_conv2d_prepack_begin / _flush): job j owns blocks [blk0[j], blk0[j+1])
constexpr int PK_JOBS = 32;
struct PackJobs {
  const float* w[PK_JOBS];
  float* wp[PK_JOBS];
  int Cout[PK_JOBS], Cin[PK_JOBS];
  unsigned char taps_full[PK_JOBS], transpose[PK_JOBS], ck[PK_JOBS], bm[PK_JOBS];
  unsigned char seln[PK_JOBS], selidx[PK_JOBS][9];
  int blk0[PK_JOBS + 1];
  int n;
};
__global__ void pack_weights_batch_kernel(const PackJobs jb) {
  int j = 0;
  while (j + 1 < jb.n && (int)blockIdx.x >= jb.blk0[j + 1]) ++j;
  TapSel sel;
  sel.n = jb.seln[j];
#pragma unroll
  for (int t = 0; t < 9; ++t) sel.idx[t] = jb.selidx[j][t];
  pack_weights_body(jb.w[j], jb.wp[j], jb.Cout[j], jb.Cin[j], jb.taps_full[j], jb.transpose[j], jb.ck[j], jb.bm[j], sel,
                    (int)blockIdx.x - jb.blk0[j], jb.blk0[j + 1] - jb.blk0[j]);
}

// zero-upsample dy (B,C,Ho,Wo) -> (B,C,Hu,Wu) with u[2i][2j] = dy[i][j]
__global__ void upsample2_kernel(const float* __restrict__ dy, float* __restrict__ up, int Ho,
                                 int Wo, int Hu, int Wu, size_t planes) {
  const size_t total = planes * Hu * Wu;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const int wu = (int)(e % Wu);
    const size_t r = e / Wu;
    const int hu = (int)(r % Hu);
    const size_t pl = r / Hu;
    float v = 0.0f;
    if (!(wu & 1) && !(hu & 1) && (hu >> 1) < Ho && (wu >> 1) < Wo)
      v = dy[(pl * Ho + (hu >> 1)) * Wo + (wu >> 1)];
    up[e] = v;
  }
}

struct FwdArgs {
  const float* x;      // (B, Cin, H, W)
  const float* wp;     // packed weights
  float* y;            // (B, Cout, Ho, Wo)
  const float* scale;  // per input channel (may be null)
  const float* shift;
  const float* residual;  // same shape as y (may be null)
  int B, Cin, H, W, Cout, Ho, Wo, ph, pw;
  int relu;
  int WT;        // pixel tiles per output row
  int ntiles;    // B*Ho*WT
  int npxg;      // pixel-tile groups (ntiles / NWAVE, rounded up)
  int ncot;      // Cout / 64
  // output addressing: pixel (ho, wo) -> (ho*oh_mul)*ow_row + wo*ow_mul + o_off in a plane of
  // oplane floats (a plain conv: 1, Wo, 1, 0, Ho*Wo)
  int oh_mul, ow_row, ow_mul, o_off;
  size_t oplane;
  size_t x_bstride, y_bstride;  // batch strides in floats (channel-slice views of wider tensors)
  const float* bias;            // per output channel, added in the epilogue (may be null)
  const float* bias_bc;         // per (batch, output channel) bias (B, Cout) (may be null)
  int out_relu;                 // epilogue ReLU (conv -> ReLU -> BN ordering of ecapa_tdnn.py)
  int last_cbase;               // first channel of the last K chunk (Cin - CK when Cin % CK != 0)
  // K split (blockIdx.y): this launch slice runs chunks [y * kchunks, min(nchunk, (y + 1) * kchunks)) and writes its
  // partial sums at y + blockIdx.y * ksplit_stride (kchunks >= nchunk, stride 0: the whole sum, the plain case)
  int kchunks;
  size_t ksplit_stride;
};

// CKT = input channels per K chunk (8 for 3x3; more for 1-4 tap kernels so a chunk still
// holds >= 16 k-steps between barriers)
// MT = 32-channel MFMA row tiles per workgroup: 2 (64 output channels), or 1 where that fills
// the chip's rounds better (small layers) or the layer has <= 32 output channels
template <int KH, int KW, int S, int CKT, int DIL = 1, int MT = 2>
struct FwdCfg {
  static constexpr int TAPS = KH * KW;
  static constexpr int BMT = 32 * MT;                  // output channels per workgroup
  static constexpr int PW = (PXT - 1) * S + (KW - 1) * DIL + 1;  // patch columns (DIL: width dilation)
  static constexpr int CHS = KH * PW;                 // channel pitch (dense: LDS-DMA is lane-linear)
  static constexpr int NE = CKT * CHS;                 // patch elements per wave
  static constexpr int NI = (NE + 63) / 64;           // DMA instructions (elements per lane)
  static constexpr int PATCHP = NI * 64;              // padded so the last DMA stays in the wave's region
  static constexpr int WSLAB = TAPS * CKT * BMT;       // floats per weight slab
  static constexpr int NWV = (WSLAB / 4 + NWAVE * 64 - 1) / (NWAVE * 64);  // 16-byte DMAs per thread
  static constexpr int BUF = WSLAB + NWAVE * PATCHP;  // floats per LDS buffer
};

constexpr int MAXC = 512;  // channels whose BN scale/shift fit the LDS table

// MODE 0: plain input.  MODE 1: input = max(0, x*scale[ci]+shift[ci]) applied when the
// operand is read from LDS (the VALU is idle under the MFMAs), so staging is a pure copy.
template <int KH, int KW, int S, int MODE, int CKT, int DIL = 1, int MT = 2>
__global__ __launch_bounds__(NWAVE * 64) void conv_fwd_kernel(FwdArgs a) {
  using C = FwdCfg<KH, KW, S, CKT, DIL, MT>;
  __shared__ __attribute__((aligned(16))) float lds[2 * C::BUF];
  __shared__ float s_scale[MODE == 1 ? MAXC : 1], s_shift[MODE == 1 ? MAXC : 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lb % a.ncot;
  const int pxg = lb / a.ncot;
  const int nt = pxg * NWAVE + wave;  // this wave's pixel tile
  const bool tile_ok = nt < a.ntiles;
  const int wt = nt % a.WT;
  const int rowid = nt / a.WT;  // b*Ho + ho
  const int ho = rowid % a.Ho;
  const int b = rowid / a.Ho;
  const int wo0 = wt * PXT;
  const int hi0 = ho * S - a.ph;
  const int wi0 = wo0 * S - a.pw;
  const size_t HW = (size_t)a.H * a.W;
  const int HWi = a.H * a.W;
  const int nchunk = (a.Cin + CKT - 1) / CKT;
  const float* __restrict__ wslab0 = a.wp + (size_t)cot * nchunk * C::WSLAB;

  // Staging = LDS-DMA (global_load_lds): no VGPR round trip, no ds_write pass, no branch.
  // The patch is copied from CLAMPED (always valid) addresses; zero padding and the
  // activation are applied when the operand is read.  Per-lane source offsets are
  // chunk-invariant and live in registers.
  const int rowc = min(rowid, a.B * a.Ho - 1);  // keep address maths inside the tensor
  const float* __restrict__ xbc = a.x + (size_t)(rowc / a.Ho) * a.x_bstride;
  int goff[C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = lane + 64 * i;
    const int cil = min(e / C::CHS, CKT - 1);
    const int rem = e % C::CHS;
    const int r = rem / C::PW;
    const int c = rem - r * C::PW;
    const int hi = min(max(hi0 + r, 0), a.H - 1), wi = min(max(wi0 + c, 0), a.W - 1);
    goff[i] = cil * HWi + hi * a.W + wi;
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  auto dma = [&](int chunk, int buf) {
    const unsigned base = lds0 + 4u * (buf * C::BUF);
    const unsigned pl = base + 4u * (C::WSLAB + wave * C::PATCHP);
    // the last chunk of a ragged Cin slides back to [Cin-CK, Cin); its already-covered
    // channels carry zero weights (pack_weights_kernel)
    const int cbase = (chunk == nchunk - 1) ? a.last_cbase : chunk * CKT;
    const float* __restrict__ xc = xbc + (size_t)cbase * HW;
#pragma unroll
    for (int i = 0; i < C::NI; ++i)
      dma4(xc + goff[i], pl + 256u * i);
    const float* __restrict__ ws = wslab0 + (size_t)chunk * C::WSLAB;
#pragma unroll
    for (int i = 0; i < C::NWV; ++i) {
      const int e0 = (wave + NWAVE * i) * 64;  // first float4 of this wave's DMA (wave-uniform)
      if (e0 < C::WSLAB / 4) {
        // (a slab that is not a whole number of 1 KB wave DMAs - 4-channel chunks of 32 output channels - must not
        // run into the first wave's patch behind it: the tail lanes sit out)
        if ((C::WSLAB / 4) % 64 == 0 || e0 + lane < C::WSLAB / 4)
          dma16(ws + 4 * (e0 + lane), base + 16u * e0);
      }
    }
  };

  // padding masks: row validity per kh (wave-uniform), column validity per (kw, lane)
  bool okm[C::TAPS];
#pragma unroll
  for (int kh = 0; kh < KH; ++kh)
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
      const int hi = hi0 + kh, wi = wi0 + l31 * S + kw * DIL;
      okm[kh * KW + kw] = hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
    }

  // wave-uniform: the whole patch of this pixel tile lies inside the image
  const bool interior = hi0 >= 0 && hi0 + KH - 1 < a.H && wi0 >= 0 && wi0 + C::PW - 1 < a.W;

  f32x16 acc0 = {0}, acc1 = {0};

  if (MODE == 1) {
    for (int e = tid; e < a.Cin; e += NWAVE * 64) {
      s_scale[e] = a.scale[e];
      s_shift[e] = a.shift[e];
    }
  }
  const int kc0 = (int)blockIdx.y * a.kchunks, kc1 = min(nchunk, kc0 + a.kchunks);  // this slice of the K loop
  dma(kc0, 0);
  dma_wait();
  __syncthreads();
  float scn[CKT / 2], shn[CKT / 2];
#pragma unroll
  for (int st = 0; st < CKT / 2; ++st) {
    scn[st] = MODE == 1 ? s_scale[kc0 * CKT + 2 * st + half] : 1.0f;
    shn[st] = MODE == 1 ? s_shift[kc0 * CKT + 2 * st + half] : 0.0f;
  }
  for (int chunk = kc0; chunk < kc1; ++chunk) {
    const int cur = (chunk - kc0) & 1;
    if (chunk + 1 < kc1) dma(chunk + 1, cur ^ 1);
    const float* __restrict__ wl = lds + cur * C::BUF;
    const float* __restrict__ pl = wl + C::WSLAB + wave * C::PATCHP;
    // BN scale/shift of this chunk's channels were fetched during the previous chunk
    float sc[CKT / 2], sh[CKT / 2];
#pragma unroll
    for (int st = 0; st < CKT / 2; ++st) {
      sc[st] = scn[st];
      sh[st] = shn[st];
    }
    if (MODE == 1 && chunk + 1 < kc1) {
#pragma unroll
      for (int st = 0; st < CKT / 2; ++st) {
        scn[st] = s_scale[(chunk + 1) * CKT + 2 * st + half];
        shn[st] = s_shift[(chunk + 1) * CKT + 2 * st + half];
      }
    }
    // k-steps of this chunk: (tap, channel pair).  Operands for step n+1 are read from
    // LDS before the MFMAs of step n are issued.
    constexpr int NSTEP = C::TAPS * (CKT / 2);
    auto ld = [&](int n, float& a0, float& a1, float& bv) {
      const int tap = n / (CKT / 2), st = n % (CKT / 2);
      const int kh = tap / KW, kw = tap % KW;
      const int cil = 2 * st + half;
      bv = pl[cil * C::CHS + kh * C::PW + kw * DIL + l31 * S];
      a0 = wl[(tap * CKT + cil) * C::BMT + l31];
      a1 = MT == 2 ? wl[(tap * CKT + cil) * C::BMT + 32 + l31] : 0.0f;
    };
    // 3-stage software pipeline per k-step: LDS read (n+2) | activate + mask (n+1) | MFMA (n),
    // so neither the LDS round trip nor the VALU chain sits between two MFMAs.  A wave issues
    // about one instruction per 4 cycles, so the per-MFMA instruction count is what bounds
    // the matrix pipe: interior tiles (no padding in reach) run a mask-free loop.
    auto steps = [&](auto masked_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      auto act = [&](int n, float bv) -> float {
        if (MODE == 1) bv = fmaxf(bv * sc[n % (CKT / 2)] + sh[n % (CKT / 2)], 0.0f);
        if (MASKED) bv = okm[n / (CKT / 2)] ? bv : 0.0f;  // zero padding of the activated tensor
        return bv;
      };
      float ra0[3], ra1[3], rb[3];
      ld(0, ra0[0], ra1[0], rb[0]);
      if (NSTEP > 1) ld(1, ra0[1], ra1[1], rb[1]);
      rb[0] = act(0, rb[0]);
#pragma unroll
      for (int n = 0; n < NSTEP; ++n) {
        if (n + 2 < NSTEP) ld(n + 2, ra0[(n + 2) % 3], ra1[(n + 2) % 3], rb[(n + 2) % 3]);
        if (n + 1 < NSTEP) rb[(n + 1) % 3] = act(n + 1, rb[(n + 1) % 3]);
        __builtin_amdgcn_sched_barrier(0);  // keep reads / VALU ahead of the MFMAs
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[n % 3], rb[n % 3], acc0, 0, 0, 0);
        if (MT == 2)
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[n % 3], rb[n % 3], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (interior)
      steps(std::false_type{});
    else
      steps(std::true_type{});
    dma_wait();
    __syncthreads();  // fences the buffer swap
  }

  // epilogue: D row i = (r&3) + 8*(r>>2) + 4*half -> output channel, col = l31 -> pixel.
  // Output pixel (ho, wo) lands at y[(ho*oh_mul)*ow_row + wo*ow_mul + o_off] of its channel
  // plane (identity for a plain conv; a parity class of a stride-2 dgrad otherwise).
  // One uniform branch per OPERAND around all 16 rows of an accumulator (not one per element: those kept every
  // residual load behind its own s_waitcnt - 32 dependent round trips per workgroup - and made the epilogue
  // 1500 instructions of a kernel whose 1-tap instances issue 128 - 512 MFMAs per wave).
  if (!tile_ok) return;
  const int wo = wo0 + l31;
  if (wo >= a.Wo) return;
  const size_t oplane = a.oplane;
  const size_t obase = (size_t)blockIdx.y * a.ksplit_stride + (size_t)b * a.y_bstride + (size_t)cot * C::BMT * oplane +
                       (size_t)(ho * a.oh_mul) * a.ow_row + (size_t)wo * a.ow_mul + a.o_off;
  const bool full = (cot + 1) * C::BMT <= a.Cout;  // every row of this channel tile is a real channel
  auto emit = [&](const f32x16& acc, int cofs) {
    const int cb = cot * C::BMT + cofs + 4 * half;
    const size_t o0 = obase + (size_t)(cofs + 4 * half) * oplane;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    float t[16];  // the residual's 16 loads go out together, ahead of the bias terms (added in the old order)
    if (a.residual != nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2);
        t[r] = (full || cb + i < a.Cout) ? a.residual[o0 + (size_t)i * oplane] : 0.0f;
      }
    }
    if (a.bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2);
        v[r] += a.bias[min(cb + i, a.Cout - 1)];
      }
    }
    if (a.bias_bc != nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2);
        v[r] += a.bias_bc[(size_t)b * a.Cout + min(cb + i, a.Cout - 1)];
      }
    }
    if (a.residual != nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += t[r];
    }
    if (a.out_relu) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
    }
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) a.y[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * oplane] = v[r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2);
        if (cb + i < a.Cout) a.y[o0 + (size_t)i * oplane] = v[r];
      }
    }
  };
  emit(acc0, 0);
  if (MT == 2) emit(acc1, 32);
}

// ------------------------------------------------- stride-2 3x3 data gradient
// dx of a 3x3 / stride 2 / pad 1 convolution (resnet.py:56 conv1 of layer2-4's first block), optionally joined with
// the data gradient of the block's 1x1 / stride 2 shortcut (resnet.py:61-66), in ONE pass over dy.
// Input pixel (2i + a, 2j + b) sees the taps kh = 1 (a = 0, dy row i) or kh = 2, 0 (a = 1, dy rows i, i + 1), same
// along w: four parity classes with 1, 2, 2 and 4 taps.  Rounds 1-3 ran each class as its own dense launch of
// conv_fwd_kernel (dy read four times, every class writing every other float of dx) and the shortcut as a fifth
// read-modify-write launch.  Here a wave owns 32 consecutive columns j of one dy row i and keeps the four classes'
// 64 x 32 tiles in registers (8 accumulators): every staged dy element feeds all its taps, the K loop is one stream
// of 9 (10 with the shortcut: its dy is a third patch row, its weights a tenth tap that lands in class (0, 0))
// taps x 4 channel pairs x 2 MFMAs per 8-channel chunk, and the epilogue writes rows 2i and 2i + 1 of dx as
// consecutive floats.  The sums are the same fmaf chains in a different order (class by class before, tap-major now).
struct S2dArgs {
  const float* dy;          // (B, K, Ho, Wo)
  const float* dysc;        // (B, K, Ho, Wo) gradient of the shortcut's output (SC)
  const float* wp;          // packed [cot][chunk][9][CKT][BMT] (pack_weights_kernel, roles swapped)
  const float* wpsc;        // packed [cot][chunk][1][CKT][BMT]
  float* dx;                // (B, M, H, W)
  const float* accumulate;  // like dx, may alias it (may be null)
  int B, K, M, H, W, Ho, Wo;
  int WT, ntiles, npxg, ncot;
  int pair;                 // W even and dx / accumulate 8-byte aligned: classes (a, 0), (a, 1) leave as one float2
};

template <int CKT, int MT, bool SC>
struct S2dCfg {
  static constexpr int TAPS = SC ? 10 : 9;
  static constexpr int BMT = 32 * MT;
  static constexpr int PW = PXT + 1;                    // dy columns j0 .. j0 + 32
  static constexpr int ROWS = SC ? 3 : 2;               // dy rows i, i + 1 (+ the shortcut's dy row i)
  static constexpr int CHS = ROWS * PW;
  static constexpr int NE = CKT * CHS;
  static constexpr int NI = (NE + 63) / 64;
  static constexpr int PATCHP = NI * 64;
  static constexpr int WSLAB = TAPS * CKT * BMT;
  static constexpr int W9 = 9 * CKT * BMT;              // the 3x3 part of a slab
  static constexpr int NWV = (WSLAB / 4 + NWAVE * 64 - 1) / (NWAVE * 64);
  static constexpr int BUF = WSLAB + NWAVE * PATCHP;
};

// (two resident workgroups per CU at 64 channels - 128 accumulator registers of <= 256 - and three at 32)
template <int CKT, int MT, bool SC>
__global__ __launch_bounds__(NWAVE * 64, MT == 2 ? 2 : 3) void conv_s2_dgrad_kernel(S2dArgs a) {
  using C = S2dCfg<CKT, MT, SC>;
  __shared__ __attribute__((aligned(16))) float lds[2 * C::BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = lb % a.ncot;
  const int pxg = lb / a.ncot;
  const int nt = pxg * NWAVE + wave;
  const bool tile_ok = nt < a.ntiles;
  const int ntc = min(nt, a.ntiles - 1);  // (a wave past the last tile stages that tile again and writes nothing)
  const int wt = ntc % a.WT;
  const int rowid = ntc / a.WT;
  const int i = rowid % a.Ho;
  const int b = rowid / a.Ho;
  const int j0 = wt * PXT;
  const size_t HWo = (size_t)a.Ho * a.Wo;
  const int nchunk = a.K / CKT;

  // staging: LDS-DMA from clamped addresses (conv_fwd_kernel's scheme); what lies outside dy is masked at operand read
  const float* gp[C::NI];
#pragma unroll
  for (int n = 0; n < C::NI; ++n) {
    const int e = lane + 64 * n;
    const int cil = min(e / C::CHS, CKT - 1);
    const int rem = e % C::CHS;
    const int r = rem / C::PW;
    const int c = rem - r * C::PW;
    const int row = min(i + (r == 1 ? 1 : 0), a.Ho - 1), col = min(j0 + c, a.Wo - 1);
    const float* __restrict__ src = (SC && r == 2) ? a.dysc : a.dy;
    gp[n] = src + ((size_t)b * a.K + cil) * HWo + (size_t)row * a.Wo + col;
  }
  const float* __restrict__ w9 = a.wp + (size_t)cot * nchunk * C::W9;
  const float* __restrict__ w1 = SC ? a.wpsc + (size_t)cot * nchunk * (CKT * C::BMT) : nullptr;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  auto dma = [&](int chunk, int buf) {
    const unsigned base = lds0 + 4u * (buf * C::BUF);
    const unsigned pl = base + 4u * (C::WSLAB + wave * C::PATCHP);
    const size_t coff = (size_t)chunk * CKT * HWo;
#pragma unroll
    for (int n = 0; n < C::NI; ++n) dma4(gp[n] + coff, pl + 256u * n);
    const float* __restrict__ ws9 = w9 + (size_t)chunk * C::W9;
#pragma unroll
    for (int n = 0; n < C::NWV; ++n) {
      const int e0 = (wave + NWAVE * n) * 64;  // first float4 of this wave's DMA (wave-uniform)
      if (e0 < C::W9 / 4) {
        dma16(ws9 + 4 * (e0 + lane), base + 16u * e0);
      } else if (SC && e0 < C::WSLAB / 4) {
        dma16(w1 + (size_t)chunk * (CKT * C::BMT) + 4 * (e0 - C::W9 / 4 + lane), base + 16u * e0);
      }
    }
  };
  static_assert((C::W9 / 4) % 64 == 0 && (C::WSLAB / 4) % 64 == 0, "slabs are whole wave DMAs");

  // masks per patch position (r, c): dy row i + 1 / column j + 1 may lie outside
  const bool ok_r1 = i + 1 < a.Ho, ok_c1 = j0 + l31 + 1 < a.Wo;

  f32x16 acc[4][MT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][m][r] = 0.0f;

  dma(0, 0);
  dma_wait();
  __syncthreads();
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int cur = chunk & 1;
    if (chunk + 1 < nchunk) dma(chunk + 1, cur ^ 1);
    const float* __restrict__ wl = lds + cur * C::BUF;
    const float* __restrict__ pl = wl + C::WSLAB + wave * C::PATCHP;
    constexpr int NSTEP = C::TAPS * (CKT / 2);
    // k-step n = (tap, channel pair): tap t < 9 is (kh, kw) = (t / 3, t % 3); kh = 0 reads dy row i + 1, kw = 0
    // column j + 1; tap 9 is the shortcut's dy (patch row 2)
    auto ld = [&](int n, float& a0, float& a1, float& bv) {
      const int tap = n / (CKT / 2), st = n % (CKT / 2);
      const int r = tap == 9 ? 2 : (tap / 3 == 0 ? 1 : 0);
      const int c = tap == 9 ? 0 : (tap % 3 == 0 ? 1 : 0);
      const int cil = 2 * st + half;
      bv = pl[cil * C::CHS + r * C::PW + c + l31];
      a0 = wl[(tap * CKT + cil) * C::BMT + l31];
      a1 = MT == 2 ? wl[(tap * CKT + cil) * C::BMT + 32 + l31] : 0.0f;
    };
    // (one code path: the select that zeroes what lies outside dy costs one VALU instruction per 2 MFMAs of 64 cycles;
    // a mask-free copy of the 80-MFMA body for interior tiles made the register allocator shuttle the accumulators)
    auto msk = [&](int n, float bv) -> float {
      const int tap = n / (CKT / 2);
      if (tap == 9) return bv;
      const bool r1 = tap / 3 == 0, c1 = tap % 3 == 0;
      if (!r1 && !c1) return bv;
      const bool ok = (!r1 || ok_r1) && (!c1 || ok_c1);
      return ok ? bv : 0.0f;
    };
    float ra0[3], ra1[3], rb[3];
    ld(0, ra0[0], ra1[0], rb[0]);
    ld(1, ra0[1], ra1[1], rb[1]);
    rb[0] = msk(0, rb[0]);
#pragma unroll
    for (int n = 0; n < NSTEP; ++n) {
      if (n + 2 < NSTEP) ld(n + 2, ra0[(n + 2) % 3], ra1[(n + 2) % 3], rb[(n + 2) % 3]);
      if (n + 1 < NSTEP) rb[(n + 1) % 3] = msk(n + 1, rb[(n + 1) % 3]);
      __builtin_amdgcn_sched_barrier(0);
      const int tap = n / (CKT / 2);
      const int cls = tap == 9 ? 0 : ((tap / 3 != 1 ? 2 : 0) + (tap % 3 != 1 ? 1 : 0));
      acc[cls][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[n % 3], rb[n % 3], acc[cls][0], 0, 0, 0);
      if (MT == 2)
        acc[cls][MT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[n % 3], rb[n % 3], acc[cls][MT - 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    dma_wait();
    __syncthreads();
  }

  // epilogue: D row (r & 3) + 8 * (r >> 2) + 4 * half -> dx channel, column l31 -> dy column j; classes (a, 0) and
  // (a, 1) of a lane are neighbours in dx row 2i + a
  if (!tile_ok) return;
  const int j = j0 + l31;
  if (j >= a.Wo) return;
  const size_t plane = (size_t)a.H * a.W;
  const bool full = (cot + 1) * C::BMT <= a.M;
  const bool ok_b1 = 2 * j + 1 < a.W;
  const bool pair = a.pair != 0;  // rows start on even floats: the two classes go out as one 8-byte store
#pragma unroll
  for (int pa = 0; pa < 2; ++pa) {
    const int h = 2 * i + pa;
    if (h >= a.H) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int cb = cot * C::BMT + 32 * m + 4 * half;
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
          const int ch = (r & 3) + 8 * (r >> 2);
          const bool okc = full || cb + ch < a.M;
          if (pair) {
            const float2 t = okc ? *reinterpret_cast<const float2*>(a.accumulate + o0 + (size_t)ch * plane)
                                 : make_float2(0.0f, 0.0f);
            t0[r] = t.x;
            t1[r] = t.y;
          } else {
            t0[r] = okc ? a.accumulate[o0 + (size_t)ch * plane] : 0.0f;
            t1[r] = (okc && ok_b1) ? a.accumulate[o0 + (size_t)ch * plane + 1] : 0.0f;
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
        const int ch = (r & 3) + 8 * (r >> 2);
        if (!full && cb + ch >= a.M) continue;
        float* __restrict__ o = a.dx + o0 + (size_t)ch * plane;
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

// ------------------------------------------------------------------- wgrad
// dW[co][ci][tap] = sum_px dy[co][px] * act(x)[ci][px shifted by tap]
// GEMM M = co (64 per workgroup), N = (ci, tap) with 32-channel MFMA column
// tiles per tap, K = pixels (pixel tiles of 32).  4 waves = (co half) x (ci
// half of a 64-channel tile); each wave keeps TAPS accumulators.
struct WgradArgs {
  const float* x;
  const float* dy;
  float* partial;  // [nsplit][Cout][Cin][taps]
  const float* scale;
  const float* shift;
  int B, Cin, H, W, Cout, Ho, Wo, ph, pw;
  int relu;
  int WT, ntiles;
  int ncot, ncit;  // co tiles (BMW), ci tiles (CT)
  int nsplit;
  int tiles_per_split;
  size_t x_bstride, dy_bstride;  // batch strides in floats (channel-slice views)
};

// CT = input channels per workgroup (64: waves = 2 co halves x 2 ci halves, 64 co;
//                                     32: waves = 4 co quarters x 1 ci tile, 128 co)
template <int KH, int KW, int S, int CT_, int DIL = 1>
struct WgCfg {
  static constexpr int TAPS = KH * KW;
  static constexpr int CT = CT_;
  static constexpr int NWCI = CT / 32;                 // waves along ci
  static constexpr int NWCO = 4 / NWCI;                // waves along co
  static constexpr int BMW = 32 * NWCO;                // co per workgroup
  static constexpr int PW = (PXT - 1) * S + (KW - 1) * DIL + 1;
  static constexpr int CHS = KH * PW;                  // dense channel pitch (LDS-DMA is lane-linear)
  static constexpr int NE = CT * CHS;                  // patch elements per workgroup
  static constexpr int NI = (NE + 255) / 256;          // patch DMA instructions per wave
  static constexpr int PATCHP = NI * 256;              // padded to whole instructions
  static constexpr int DPAIR = 65;                     // pitch of a 2-row (2 x 32 px) dy DMA: odd
  static constexpr int ND = BMW / 2 / 4;               // dy DMA instructions per wave
  static constexpr int DYT = (BMW / 2) * DPAIR;
  static constexpr int BUF = PATCHP + DYT;
};

// MODE 0: plain input.  MODE 1: input = max(0, x*scale[ci]+shift[ci]), applied to the B
// operand after it is read from LDS (lanes = channels, so scale/shift are two registers).
// Staging is LDS-DMA from clamped addresses; padding is masked at operand-read time.
template <int KH, int KW, int S, int CT_, int MODE, int DIL = 1>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  using C = WgCfg<KH, KW, S, CT_, DIL>;
  __shared__ __attribute__((aligned(16))) float lds[2 * C::BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = wave % C::NWCO;  // co sub-tile
  const int ch = wave / C::NWCO;  // ci sub-tile

  int lb = blockIdx.x;
  const int split = lb % a.nsplit;
  lb /= a.nsplit;
  const int cit = lb % a.ncit;
  const int cot = lb / a.ncit;
  const int ci0 = cit * C::CT;
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
  const int HWi = a.H * a.W;
  const int HoWoi = a.Ho * a.Wo;
  const size_t HoWo = (size_t)a.Ho * a.Wo;

  // this lane's channel for the B operand
  const int ci = ci0 + ch * 32 + l31;
  const bool ci_ok = ci < a.Cin;
  float sc = 1.0f, sh = 0.0f;
  if (MODE == 1) {
    sc = a.scale[min(ci, a.Cin - 1)];
    sh = a.shift[min(ci, a.Cin - 1)];
  }

  // Tile-invariant part of every lane's patch source offset: element e -> (cil, r, c).
  // For interior tiles the address is just tile_base + eoff[i] (one SGPR base, no VALU).
  int eoff[C::NI];
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int e = 256 * i + 64 * wave + lane;
    const int cil = min(e / C::CHS, C::CT - 1);
    const int rem = e % C::CHS;
    const int r = rem / C::PW;
    eoff[i] = min(ci0 + cil, a.Cin - 1) * HWi + r * a.W + (rem - r * C::PW);
  }
  auto dma = [&](int nt, int buf) {
    const int wt = nt % a.WT;
    const int rowid = nt / a.WT;
    const int ho = rowid % a.Ho;
    const int b = rowid / a.Ho;
    const int wo0 = wt * PXT;
    const int hi0 = ho * S - a.ph, wi0 = wo0 * S - a.pw;
    float* pl = lds + buf * C::BUF;
    float* dl = pl + C::PATCHP;
    const float* __restrict__ xb = a.x + (size_t)b * a.x_bstride;
    const bool inside = hi0 >= 0 && hi0 + KH - 1 < a.H && wi0 >= 0 && wi0 + C::PW - 1 < a.W;
    if (inside) {  // wave-uniform
      const float* __restrict__ xt = xb + (hi0 * a.W + wi0);
#pragma unroll
      for (int i = 0; i < C::NI; ++i)
        dma4_auto(xt + eoff[i], pl + 256 * i + 64 * wave);
    } else if (MODE == 0) {
      // plain input: the cells outside the image are ZEROED in LDS once the tile has landed (zero_padding below),
      // so any address inside the tensor will do for them - the interior offset clamped to the utterance, two
      // instructions per element instead of the row / column clamps
      const int tb = hi0 * a.W + wi0, lim = a.Cin * HWi - 1;
#pragma unroll
      for (int i = 0; i < C::NI; ++i)
        dma4_auto(xb + min(max(tb + eoff[i], 0), lim), pl + 256 * i + 64 * wave);
    } else {
#pragma unroll
      for (int i = 0; i < C::NI; ++i) {
        const int e = 256 * i + 64 * wave + lane;
        const int cil = min(e / C::CHS, C::CT - 1);
        const int rem = e % C::CHS;
        const int r = rem / C::PW;
        const int c = rem - r * C::PW;
        const int cc = min(ci0 + cil, a.Cin - 1);
        const int hi = min(max(hi0 + r, 0), a.H - 1), wi = min(max(wi0 + c, 0), a.W - 1);
        dma4_auto(xb + cc * HWi + hi * a.W + wi, pl + 256 * i + 64 * wave);
      }
    }
    // dy rows (co pairs): lane -> (co = 2j + half, px = l31), column clamped to the row
    const float* __restrict__ dyb =
        a.dy + (size_t)b * a.dy_bstride + (size_t)cot * C::BMW * HoWo + (size_t)ho * a.Wo;
    const int dyo = half * HoWoi + min(wo0 + l31, a.Wo - 1);
#pragma unroll
    for (int i = 0; i < C::ND; ++i) {
      const int j = wave + 4 * i;  // row pair
      dma4_auto(dyb + 2 * j * HoWoi + dyo, dl + j * C::DPAIR);
    }
  };

  f32x16 acc[C::TAPS];
#pragma unroll
  for (int t = 0; t < C::TAPS; ++t) acc[t] = (f32x16){0};

  if (t_begin < t_end) dma(t_begin, 0);
  __syncthreads();
  // MODE 0: padding as zeros IN LDS.  A tile that reaches over the image edge (the first and the last tile of every
  // row, every tile of a padded row: 26 / 47 / 78 % of the tiles of the ResNet's three stride-2 layers) used to run
  // a loop with two selects and a compare per MFMA behind 20 address instructions per staged element; now its
  // out-of-image rows and columns and the dy columns behind the row's end are overwritten with zeros after the
  // tile has landed - a few LDS stores per thread - and every tile runs the mask-free loop.  (What a zeroed dy
  // column multiplies is staged from clamped addresses: real, finite values.)
  auto zero_padding = [&](float* plw, int hi0, int wi0, int wo0) {
    float* dlw = plw + C::PATCHP;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
      if (hi0 + kh < 0 || hi0 + kh >= a.H)  // wave-uniform
        for (int e = tid; e < C::CT * C::PW; e += 256) {
          const int cil = e / C::PW;
          plw[cil * C::CHS + kh * C::PW + (e - cil * C::PW)] = 0.0f;
        }
    const int pxv = min(PXT, a.Wo - wo0);                  // real pixels of this tile
    const int clast = (pxv - 1) * S + (KW - 1) * DIL;      // last patch column a real pixel reads
    const int cell = (tid / KH) * C::CHS + (tid % KH) * C::PW;
    if (tid < C::CT * KH) {
      for (int c = 0; c < -wi0; ++c) plw[cell + c] = 0.0f;
      for (int c = max(0, a.W - wi0); c <= clast; ++c) plw[cell + c] = 0.0f;
    }
    if (pxv < PXT)
      for (int e = tid; e < C::BMW * PXT; e += 256) {
        const int co = e / PXT, px = e % PXT;
        if (px >= pxv) dlw[(co >> 1) * C::DPAIR + (co & 1) * 32 + px] = 0.0f;
      }
  };
  for (int nt = t_begin; nt < t_end; ++nt) {
    const int cur = (nt - t_begin) & 1;
    const int wt = nt % a.WT;
    const int ho = (nt / a.WT) % a.Ho;
    const int wo0 = wt * PXT;
    const int hi0 = ho * S - a.ph, wi0 = wo0 * S - a.pw;
    if (MODE == 0) {
      // (before the next tile's DMA goes out: the compiler drains vmcnt in front of LDS stores it cannot tell apart
      // from a DMA's destination, and here nothing is in flight)
      const bool edge = !(hi0 >= 0 && hi0 + KH - 1 < a.H && wi0 >= 0 && wi0 + C::PW - 1 < a.W && wo0 + PXT <= a.Wo);
      if (edge) {  // wave-uniform
        zero_padding(lds + cur * C::BUF, hi0, wi0, wo0);
        __syncthreads();
      }
    }
    if (nt + 1 < t_end) dma(nt + 1, cur ^ 1);
    const float* __restrict__ pl = lds + cur * C::BUF;
    const float* __restrict__ dl = pl + C::PATCHP;
    const float* __restrict__ pbase = pl + (ch * 32 + l31) * C::CHS;
    const int aco = mt * 32 + l31;
    const float* __restrict__ abase = dl + (aco >> 1) * C::DPAIR + (aco & 1) * 32;
    bool rok[KH];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) rok[kh] = ci_ok && hi0 + kh >= 0 && hi0 + kh < a.H;
    // valid patch columns c (= px*S + kw) of this tile: [cmin, cmax]; per lane the test is
    // done on the compile-time part of c, so fold the lane's half*S into the bounds
    const int clo = max(0, -wi0) - half * S;
    const unsigned cspan = (unsigned)(min(C::PW - 1, a.W - 1 - wi0) - half * S - clo);
    const int pxmax = a.Wo - wo0 - half;  // px = 2 st + half is a real pixel iff 2 st < pxmax

    // 16 k-steps (pixel pairs) x TAPS MFMAs.  Software pipeline, interleaved PER MFMA so the
    // VALU/LDS work of the following steps issues in the shadow of each 64-cycle MFMA:
    //   tap t of step n+2: LDS read | tap t of step n+1: activate (+ mask) | tap t of step n: MFMA
    // One wave issues ~1 instruction per 4 cycles, so instructions per MFMA bound the matrix
    // pipe: interior tiles (no padding, full row, real channels) take the mask-free loop.
    const bool interior = hi0 >= 0 && hi0 + KH - 1 < a.H && wi0 >= 0 &&
                          wi0 + C::PW - 1 < a.W && wo0 + PXT <= a.Wo && ci0 + C::CT <= a.Cin;
    auto ldb = [&](int st, int t) -> float {
      const int kh = t / KW, kw = t % KW;
      return pbase[kh * C::PW + (2 * st + half) * S + kw * DIL];
    };
    auto steps = [&](auto masked_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      auto actb = [&](int st, int t, float v) -> float {
        const int kh = t / KW, kw = t % KW;
        if (MODE == 1) v = fmaxf(v * sc + sh, 0.0f);
        if (MASKED) {
          const bool ok = rok[kh] && (unsigned)(2 * st * S + kw * DIL - clo) <= cspan;
          v = ok ? v : 0.0f;
        }
        return v;
      };
      float ra[3], rb[3][C::TAPS];
      ra[0] = abase[half];
      ra[1] = abase[2 + half];
#pragma unroll
      for (int t = 0; t < C::TAPS; ++t) {
        rb[0][t] = actb(0, t, ldb(0, t));
        rb[1][t] = ldb(1, t);
      }
      if (MASKED) ra[0] = (0 < pxmax) ? ra[0] : 0.0f;
#pragma unroll
      for (int st = 0; st < PXT / 2; ++st) {
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) {
          if (st + 2 < PXT / 2) {
            if (t == 0) ra[(st + 2) % 3] = abase[2 * (st + 2) + half];
            rb[(st + 2) % 3][t] = ldb(st + 2, t);
          }
          if (st + 1 < PXT / 2) {
            if (MASKED && t == 0)
              ra[(st + 1) % 3] = (2 * (st + 1) < pxmax) ? ra[(st + 1) % 3] : 0.0f;
            rb[(st + 1) % 3][t] = actb(st + 1, t, rb[(st + 1) % 3][t]);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[st % 3], rb[st % 3][t], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (MODE == 0 || interior)
      steps(std::false_type{});
    else
      steps(std::true_type{});
    __syncthreads();  // drains the DMA (vmcnt) and fences the buffer swap
  }

  // D[i = co][j = ci] per tap -> partial[split][tap][co][ci]: lanes = consecutive ci, so
  // every store is a 128-byte row segment (reduce_partials_kernel restores [co][ci][tap])
  if (!ci_ok) return;
  float* __restrict__ out = a.partial + (size_t)split * a.Cout * a.Cin * C::TAPS;
#pragma unroll
  for (int t = 0; t < C::TAPS; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int co = cot * C::BMW + mt * 32 + i;
      out[((size_t)t * a.Cout + co) * a.Cin + ci] = acc[t][r];
    }
  }
}

// dw[e] = sum_k partial[k][e]; with taps > 1 the partials are [tap][co*ci] and dw is [co*ci][tap].
// A workgroup owns 64 consecutive outputs; its 4 waves take the split index k = w, w + 4, ... (so every load
// instruction is one coalesced 256-byte row segment, 8 in flight per wave) and their 4 sums meet in LDS in a
// fixed order: deterministic, whatever the split count.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial,
                                                              float* __restrict__ dw, size_t n,
                                                              int nsplit, int taps) {
  __shared__ float s_part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (nsplit <= 16) {
    // few splits over many outputs (the wide layers): one output per thread, all its loads in flight at once
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = k < nsplit ? partial[(size_t)k * n + e] : 0.0f;
      float s = v[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) s += v[k];
      if (taps > 1) {
        const size_t plane = n / taps;
        const size_t t = e / plane, cc = e - t * plane;
        dw[cc * taps + t] = s;
      } else {
        dw[e] = s;
      }
    }
    return;
  }
  const size_t nblk = (n + 63) / 64;
  for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const size_t e = blk * 64 + lane;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < n) {
      int k = w;
      for (; k + 28 < nsplit; k += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += partial[(size_t)(k + 4 * u) * n + e];
      }
      for (int u = 0; k < nsplit; k += 4, ++u) acc[u] += partial[(size_t)k * n + e];
    }
    s_part[w][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (w == 0 && e < n) {
      const float s = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
      if (taps > 1) {
        const size_t plane = n / taps;  // Cout*Cin
        const size_t t = e / plane, cc = e - t * plane;
        dw[cc * taps + t] = s;
      } else {
        dw[e] = s;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------- small-Cin direct kernels
// conv1 of the ResNet: Cin = 1, Cout = 16, 9x3, stride (3,1), pad (1,1)
// (resnet.py:131): 0.1 % of the FLOPs, HBM-bound on its 16-channel output.
struct DirectArgs {
  const float* x;
  const float* w;
  float* y;
  const float* dy;
  float* partial;
  int B, Cin, H, W, Cout, KH, KW, sh, sw, ph, pw, Ho, Wo;
};

constexpr int DIRECT_MAX_W = 16 * 27;

__global__ __launch_bounds__(256) void conv_direct_fwd_kernel(DirectArgs a) {
  // weights as [k][16 co]: the 16 output channels of a tap are four 16-byte broadcast reads (one 4-byte read per
  // (co, tap) made the LDS instruction stream the bound: 79 us for a 67 MB layer)
  __shared__ __attribute__((aligned(16))) float sw_[DIRECT_MAX_W];
  const int K = a.Cin * a.KH * a.KW;
  for (int e = threadIdx.x; e < 16 * K; e += 256) {
    const int k = e >> 4, co = e & 15;
    sw_[e] = co < a.Cout ? a.w[co * K + k] : 0.0f;
  }
  __syncthreads();
  const size_t npx = (size_t)a.B * a.Ho * a.Wo;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npx; p += (size_t)gridDim.x * 256) {
    const int wo = (int)(p % a.Wo);
    const size_t r = p / a.Wo;
    const int ho = (int)(r % a.Ho);
    const int b = (int)(r / a.Ho);
    float acc[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) acc[co] = 0.0f;
    for (int ci = 0; ci < a.Cin; ++ci)
      for (int kh = 0; kh < a.KH; ++kh) {
        const int hi = ho * a.sh - a.ph + kh;
        if (hi < 0 || hi >= a.H) continue;
        for (int kw = 0; kw < a.KW; ++kw) {
          const int wi = wo * a.sw - a.pw + kw;
          if (wi < 0 || wi >= a.W) continue;
          const float xv = a.x[(((size_t)b * a.Cin + ci) * a.H + hi) * a.W + wi];
          const int k = (ci * a.KH + kh) * a.KW + kw;
          const float4* __restrict__ wk = reinterpret_cast<const float4*>(&sw_[k * 16]);
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const float4 w4 = wk[c4];
            acc[4 * c4 + 0] = fmaf(xv, w4.x, acc[4 * c4 + 0]);
            acc[4 * c4 + 1] = fmaf(xv, w4.y, acc[4 * c4 + 1]);
            acc[4 * c4 + 2] = fmaf(xv, w4.z, acc[4 * c4 + 2]);
            acc[4 * c4 + 3] = fmaf(xv, w4.w, acc[4 * c4 + 3]);
          }
        }
      }
#pragma unroll
    for (int co = 0; co < 16; ++co)
      if (co < a.Cout) a.y[(((size_t)b * a.Cout + co) * a.Ho + ho) * a.Wo + wo] = acc[co];
  }
}

// wgrad for the same layer: one block per (b, ho) row slice; block-reduces
// dy[co][px] * x[patch k] into partial[block][co][k].
__global__ __launch_bounds__(256) void conv_direct_wgrad_kernel(DirectArgs a) {
  const int K = a.Cin * a.KH * a.KW;  // <= 27
  const int row = blockIdx.x;         // b*Ho + ho
  const int ho = row % a.Ho, b = row / a.Ho;
  __shared__ float red[4][DIRECT_MAX_W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // each wave owns a subset of (co,k) pairs; lanes stride over wo
  for (int e = wave; e < a.Cout * K; e += 4) {
    const int co = e / K, k = e - co * K;
    const int kw = k % a.KW, kh = (k / a.KW) % a.KH, ci = k / (a.KW * a.KH);
    const int hi = ho * a.sh - a.ph + kh;
    float s = 0.0f;
    if (hi >= 0 && hi < a.H) {
      const float* __restrict__ dyr = a.dy + (((size_t)b * a.Cout + co) * a.Ho + ho) * a.Wo;
      const float* __restrict__ xr = a.x + (((size_t)b * a.Cin + ci) * a.H + hi) * a.W;
      for (int wo = lane; wo < a.Wo; wo += 64) {
        const int wi = wo * a.sw - a.pw + kw;
        if (wi >= 0 && wi < a.W) s = fmaf(dyr[wo], xr[wi], s);
      }
    }
    s = air_wave_sum(s);
    if (lane == 0) a.partial[(size_t)row * a.Cout * K + e] = s;
  }
  (void)red;
}

// The same partials, one workgroup per (b, ho) output row with the row's operands in LDS: dy [Cout][Wo] and the
// KH input rows of every input channel, zero-padded [Cin * KH][W + 2 pw].  A thread owns one (co, ci, kh) and a
// segment of the row and keeps KW accumulators (the kernel above re-reads both rows from L1/L2 for each of its
// 432 (co, k) pairs and wave-reduces each: 472 us for a 67 MB problem; this one 69 us).  Segments are folded in
// order, so the summation order is fixed.
constexpr int DIRECT_KW_MAX = 4;
__global__ __launch_bounds__(512) void conv_direct_wgrad_rows_kernel(DirectArgs a, int nseg) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  // pitches: multiples of 4 floats so that a thread can walk its row 4 pixels at a time with 16-byte reads (the
  // padding is zero-filled: it contributes nothing); + 4 on x for the KW - 1 floats behind the last quad
  const int DP = (a.Wo + 3) & ~3;
  const int XW = ((a.W + 2 * a.pw + 3) & ~3) + 4;
  const int nrow = a.Cin * a.KH;
  float* sdy = dsm;                   // [Cout][DP]
  float* sx = dsm + a.Cout * DP;      // [Cin * KH][XW]
  const int row = blockIdx.x;         // b * Ho + ho
  const int ho = row % a.Ho, b = row / a.Ho;
  const int tid = threadIdx.x, nt = blockDim.x;
  // staging with eight loads in flight per thread (a load-then-store loop pays one memory latency per element)
  const int ndy = a.Cout * DP, nx = nrow * XW;
  for (int e0 = tid; e0 < ndy; e0 += 8 * nt) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * nt;
      const int co = e / DP, wo = e - co * DP;
      v[u] = (e < ndy && wo < a.Wo) ? a.dy[(((size_t)b * a.Cout + co) * a.Ho + ho) * a.Wo + wo] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + u * nt < ndy) sdy[e0 + u * nt] = v[u];
  }
  for (int e0 = tid; e0 < nx; e0 += 8 * nt) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * nt;
      const int r = e / XW, c = e - r * XW;
      const int ci = r / a.KH, kh = r - ci * a.KH;
      const int hi = ho * a.sh - a.ph + kh, wi = c - a.pw;
      v[u] = (e < nx && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W)
                 ? a.x[(((size_t)b * a.Cin + ci) * a.H + hi) * a.W + wi] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + u * nt < nx) sx[e0 + u * nt] = v[u];
  }
  __syncthreads();
  const int npair = a.Cout * nrow;
  const int seg = tid / npair, pair = tid - seg * npair;
  float acc[DIRECT_KW_MAX];
#pragma unroll
  for (int k = 0; k < DIRECT_KW_MAX; ++k) acc[k] = 0.0f;
  const bool active = seg < nseg;
  const int co = pair / nrow, r = pair - co * nrow;
  if (active) {
    const float* __restrict__ pd = sdy + co * DP;
    const float* __restrict__ px = sx + r * XW;
    if (a.sw == 1 && a.KW == 3) {
      // four pixels per step: one 16-byte read of dy, one 16-byte + one 8-byte read of x, twelve FMAs
      const int nq = DP / 4;
      const int q_lo = (int)((long long)nq * seg / nseg), q_hi = (int)((long long)nq * (seg + 1) / nseg);
      for (int q = q_lo; q < q_hi; ++q) {
        const float4 d = *reinterpret_cast<const float4*>(pd + 4 * q);
        const float4 x0 = *reinterpret_cast<const float4*>(px + 4 * q);
        const float2 x1 = *reinterpret_cast<const float2*>(px + 4 * q + 4);
        acc[0] = fmaf(d.x, x0.x, acc[0]); acc[1] = fmaf(d.x, x0.y, acc[1]); acc[2] = fmaf(d.x, x0.z, acc[2]);
        acc[0] = fmaf(d.y, x0.y, acc[0]); acc[1] = fmaf(d.y, x0.z, acc[1]); acc[2] = fmaf(d.y, x0.w, acc[2]);
        acc[0] = fmaf(d.z, x0.z, acc[0]); acc[1] = fmaf(d.z, x0.w, acc[1]); acc[2] = fmaf(d.z, x1.x, acc[2]);
        acc[0] = fmaf(d.w, x0.w, acc[0]); acc[1] = fmaf(d.w, x1.x, acc[1]); acc[2] = fmaf(d.w, x1.y, acc[2]);
      }
    } else {
      const int w_lo = (int)((long long)a.Wo * seg / nseg), w_hi = (int)((long long)a.Wo * (seg + 1) / nseg);
      for (int wo = w_lo; wo < w_hi; ++wo) {
        const float d = pd[wo];
#pragma unroll
        for (int k = 0; k < DIRECT_KW_MAX; ++k)
          if (k < a.KW) acc[k] = fmaf(d, px[wo * a.sw + k], acc[k]);
      }
    }
  }
  __syncthreads();  // the staged rows are dead: their space takes the segments' sums
  float* red = dsm;  // [nseg][npair][KW]
  if (active)
    for (int k = 0; k < a.KW; ++k) red[((size_t)seg * npair + pair) * a.KW + k] = acc[k];
  __syncthreads();
  const int K = nrow * a.KW;
  for (int e = tid; e < npair * a.KW; e += nt) {
    float s = 0.0f;
    for (int g = 0; g < nseg; ++g) s += red[(size_t)g * npair * a.KW + e];
    // e = (co * nrow + r) * KW + k = co * K + (r * KW + k): the layout reduce_partials_kernel expects
    a.partial[(size_t)row * a.Cout * K + e] = s;
  }
}

// reduce_partials_kernel: one workgroup per 64 outputs, at most 8 per CU
int reduce_grid(size_t n, int nsplit) {
  const size_t g = nsplit <= 16 ? (n + 255) / 256 : (n + 63) / 64;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

int grid_for(size_t n, int per_block = 256, int cap = 256 * 16) {
  size_t g = (n + per_block - 1) / per_block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

bool generic_ok(const AirConv2d* p) {
  const bool k33 = p->KH == 3 && p->KW == 3, k11 = p->KH == 1 && p->KW == 1;
  const bool s_ok = (p->sh == 1 && p->sw == 1) || (p->sh == 2 && p->sw == 2);
  return (k33 || k11) && s_ok && p->sh == p->sw && p->Cin % 8 == 0 && p->Cout % BM == 0;  // wgrad tiles are 64 wide
}
bool direct_ok(const AirConv2d* p) {
  return p->Cout <= 16 && p->Cout * p->Cin * p->KH * p->KW <= DIRECT_MAX_W;
}
bool shape_ok(const AirConv2d* p) {
  if (p->B <= 0 || p->Cin <= 0 || p->H <= 0 || p->W <= 0 || p->Cout <= 0) return false;
  if (p->KH <= 0 || p->KW <= 0 || p->sh <= 0 || p->sw <= 0 || p->ph < 0 || p->pw < 0) return false;
  return p->Ho == (p->H + 2 * p->ph - p->KH) / p->sh + 1 &&
         p->Wo == (p->W + 2 * p->pw - p->KW) / p->sw + 1 && p->Ho > 0 && p->Wo > 0;
}

// out[b][c][s] = a[b][c][s] for c < Ca; batch strides differ (channel padding / unpadding)
__global__ __launch_bounds__(256) void copy_rows_kernel(float* __restrict__ out, size_t ob,
                                                        const float* __restrict__ a, size_t ab, size_t n) {
  const size_t b = blockIdx.y;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[b * ob + i] = a[b * ab + i];
}

// 3x3 / stride 1 / pad 1: the Winograd kernels' shape
bool wino_shape(const AirConv2d* p) {
  return p->KH == 3 && p->KW == 3 && p->sh == 1 && p->sw == 1 && p->ph == 1 && p->pw == 1;
}

struct FwdGeom {
  int B, Cin, H, W, Cout, KH, KW, S, ph, pw, Ho, Wo;
  int oh_mul, ow_row, ow_mul, o_off;  // output addressing (see FwdArgs)
  size_t oplane;
  size_t x_bstride, y_bstride;
  const float* bias;
  int out_relu;
  int dil;
  const float* bias_bc;
};

FwdGeom plain_geom(int B, int Cin, int H, int W, int Cout, int KH, int KW, int S, int ph, int pw,
                   int Ho, int Wo) {
  FwdGeom g = {B, Cin, H, W, Cout, KH, KW, S, ph, pw, Ho, Wo, 1, Wo, 1, 0, (size_t)Ho * Wo,
               (size_t)Cin * H * W, (size_t)Cout * Ho * Wo, nullptr, 0, 1, nullptr};
  return g;
}

// channels per K chunk: keep >= 16 k-steps between barriers for 1-4 tap kernels
int pick_ck(int taps, int cin) {
  if (taps == 3) return cin % 16 == 0 ? 16 : 8;  // dilated 1x3 conv1d
  if (taps >= 4) return 8;
  if (taps == 2) return cin % 16 == 0 ? 16 : 8;
  return cin % 32 == 0 ? 32 : (cin % 16 == 0 ? 16 : 8);
}

template <int KH, int KW, int S, int CKT, int DIL = 1>
void launch_fwd(const FwdArgs& a, int mt, int ksplit, hipStream_t st) {
  const int nblk = a.npxg * a.ncot;
#define AIR_LAUNCH(MODE_, MT_)                                                                   \
  hipLaunchKernelGGL((conv_fwd_kernel<KH, KW, S, MODE_, CKT, DIL, MT_>), dim3(nblk, ksplit),     \
                     dim3(NWAVE * 64), 0, st, a)
  if (mt == 2) {
    if (a.scale != nullptr) AIR_LAUNCH(1, 2); else AIR_LAUNCH(0, 2);
  } else {
    if (a.scale != nullptr) AIR_LAUNCH(1, 1); else AIR_LAUNCH(0, 1);
  }
#undef AIR_LAUNCH
}

// 32- or 64-channel workgroup tiles?  Workgroups are equal-sized, so a partially filled last
// round of the 256 CUs is pure loss (layer4: 1152 workgroups on 512 slots = 2.25 rounds, 25 %
// idle).  32-channel tiles need less LDS (3 resident workgroups per CU) and double the count.
int pick_mt(int ntiles, int cout);
int mt_for(int B, int Ho, int Wo, int cout) { return pick_mt(B * Ho * ((Wo + PXT - 1) / PXT), cout); }

int pick_mt(int ntiles, int cout) {
  if (cout <= 32) return 1;
  const int npxg = (ntiles + NWAVE - 1) / NWAVE;
  const double r2 = npxg * ((cout + 63) / 64) / 512.0, r1 = npxg * ((cout + 31) / 32) / 768.0;
  const double eff2 = r2 / ceil(r2), eff1 = r1 / ceil(r1);
  const int force = air_opt(AIR_OPT_CONV_MT);
  if (force == 1 || force == 2) return force;
  return (eff1 > eff2 + 0.08) ? 1 : 2;
}

// How pack() serves the direct kernels' weight slabs.  They depend on the weights only, so - like the Winograd
// transforms - they can be produced off the critical path (air_conv2d_prepack) and handed back as w_packed:
//   PK_RUN      transform `w` into the workspace slab in front of the launch (the default)
//   PK_SIZE     air_conv2d_prepack_bytes: walk the same code path, add up the slab sizes, launch nothing
//   PK_COLLECT  air_conv2d_prepack: transform into consecutive slabs of the caller's buffer, launch no convolution
//   PK_USE      w_packed given: take the next slab of the caller's buffer instead of transforming
// One walk of fwd_generic / dgrad_generic serves all four, so the slabs can never be laid out differently from how
// they are consumed.
enum { PK_RUN = 0, PK_SIZE, PK_COLLECT, PK_USE };
struct PackCtx {
  int mode;
  float* cur;
  size_t used;  // floats
};
thread_local PackCtx g_pk = {PK_RUN, nullptr, 0};
struct PackScope {
  PackCtx saved;
  PackScope(int mode, float* buf) : saved(g_pk) { g_pk = PackCtx{mode, buf, 0}; }
  ~PackScope() { g_pk = saved; }
};
inline bool pk_dry() { return g_pk.mode == PK_SIZE || g_pk.mode == PK_COLLECT; }  // no convolution launches

// y = conv(act(x), packed w): shared by fwd and every dgrad
int reduce_ksplit(const float* partial, float* y, size_t n, int ksplit, hipStream_t st) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_grid(n, ksplit)), dim3(256), 0, st, partial, y, n, ksplit, 1);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

// ksplit > 1: the K loop in `ksplit` slices of whole chunks, each slice's sums in its own copy of y inside `partial`
// (ksplit x B x y_bstride floats), summed in slice order by reduce_partials_kernel - for layers whose pixel x channel
// tiles leave most of the chip's workgroup slots empty (plain convolutions only: no prologue, no epilogue operand)
int run_fwd(const float* x, const float* wp, float* y, const float* scale, const float* shift,
            int relu, const float* residual, const FwdGeom& g, int ck, int mt, double flops,
            hipStream_t st, int ksplit = 1, float* partial = nullptr) {
  if (pk_dry()) return AIR_OK;
  if (ksplit > 1 && !pk_dry() && (partial == nullptr || scale != nullptr || residual != nullptr || g.bias != nullptr ||
                     g.bias_bc != nullptr || g.out_relu))
    return AIR_EINVAL;
  FwdArgs a;
  a.x = x; a.wp = wp; a.y = ksplit > 1 ? partial : y; a.scale = scale; a.shift = shift; a.residual = residual;
  a.B = g.B; a.Cin = g.Cin; a.H = g.H; a.W = g.W; a.Cout = g.Cout; a.Ho = g.Ho; a.Wo = g.Wo;
  a.ph = g.ph; a.pw = g.pw; a.relu = relu;
  a.WT = (g.Wo + PXT - 1) / PXT;
  a.ntiles = g.B * g.Ho * a.WT;
  a.npxg = (a.ntiles + NWAVE - 1) / NWAVE;
  a.ncot = (g.Cout + 32 * mt - 1) / (32 * mt);
  a.oh_mul = g.oh_mul; a.ow_row = g.ow_row; a.ow_mul = g.ow_mul; a.o_off = g.o_off;
  a.oplane = g.oplane;
  a.x_bstride = g.x_bstride; a.y_bstride = g.y_bstride;
  a.bias = g.bias; a.out_relu = g.out_relu; a.bias_bc = g.bias_bc;
  a.last_cbase = (g.Cin % ck != 0) ? g.Cin - ck : (g.Cin / ck - 1) * ck;
  const int nchunk_all = (g.Cin + ck - 1) / ck;
  a.kchunks = (nchunk_all + ksplit - 1) / ksplit;
  a.ksplit_stride = ksplit > 1 ? (size_t)g.B * g.y_bstride : 0;
  if (g.Cin < ck) return AIR_EUNSUPPORTED;
  if (g.Cin % ck != 0 && scale != nullptr) return AIR_EUNSUPPORTED;  // ragged Cin: plain input only
  const int key = g.KH * 1000 + g.KW * 100 + g.S * 10 + (g.dil - 1);
#define AIR_FWD_CASE_D(KH_, KW_, S_, CK_, DIL_, KID_)                         \
  if (key == KH_ * 1000 + KW_ * 100 + S_ * 10 + (DIL_ - 1) && ck == CK_) {      \
    AirProfScope ps(KID_, flops, st);                                           \
    launch_fwd<KH_, KW_, S_, CK_, DIL_>(a, mt, ksplit, st);                     \
    AIR_CHECK_LAUNCH();                                                         \
    return ksplit > 1 ? reduce_ksplit(partial, y, (size_t)g.B * g.y_bstride, ksplit, st) : AIR_OK; \
  }
#define AIR_FWD_CASE(KH_, KW_, S_, CK_, KID_) AIR_FWD_CASE_D(KH_, KW_, S_, CK_, 1, KID_)
  AIR_FWD_CASE(3, 3, 1, 8, AIR_K_CONV_FWD_331)
  AIR_FWD_CASE(3, 3, 2, 8, AIR_K_CONV_FWD_332)
  AIR_FWD_CASE(3, 3, 2, 4, AIR_K_CONV_FWD_332)
  AIR_FWD_CASE(1, 1, 1, 32, AIR_K_CONV_FWD_111)
  AIR_FWD_CASE(1, 1, 1, 16, AIR_K_CONV_FWD_111)
  AIR_FWD_CASE(1, 1, 1, 8, AIR_K_CONV_FWD_111)
  AIR_FWD_CASE(1, 1, 2, 32, AIR_K_CONV_FWD_112)
  AIR_FWD_CASE(1, 1, 2, 16, AIR_K_CONV_FWD_112)
  AIR_FWD_CASE(1, 1, 2, 8, AIR_K_CONV_FWD_112)
  AIR_FWD_CASE(1, 2, 1, 16, AIR_K_CONV_FWD_CLS)
  AIR_FWD_CASE(2, 1, 1, 16, AIR_K_CONV_FWD_CLS)
  AIR_FWD_CASE(2, 2, 1, 8, AIR_K_CONV_FWD_CLS)
  // ECAPA-TDNN conv1d layers seen as H = 1 images (ecapa_tdnn.py:46,111)
  AIR_FWD_CASE(1, 5, 1, 8, AIR_K_CONV_FWD_1D)
  AIR_FWD_CASE_D(1, 3, 1, 16, 1, AIR_K_CONV_FWD_CLS)  // one kernel row of a full-height conv's dgrad (conv5)
  AIR_FWD_CASE_D(1, 3, 1, 16, 2, AIR_K_CONV_FWD_1D)
  AIR_FWD_CASE_D(1, 3, 1, 16, 3, AIR_K_CONV_FWD_1D)
  AIR_FWD_CASE_D(1, 3, 1, 16, 4, AIR_K_CONV_FWD_1D)
#undef AIR_FWD_CASE
#undef AIR_FWD_CASE_D
  return AIR_EUNSUPPORTED;
}

// air_conv2d_prepack_begin .. _flush: slabs are recorded (this thread) and packed by one launch per 32 at the flush
thread_local bool g_pk_defer = false;
thread_local PackJobs g_pk_jobs;
int pack_run_jobs(hipStream_t st) {
  if (g_pk_jobs.n == 0) return AIR_OK;
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(g_pk_jobs.blk0[g_pk_jobs.n]), dim3(256), 0, st, g_pk_jobs);
  g_pk_jobs.n = 0;
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int pack(const float* w, float*& wp, int Cout, int Cin, int taps_full, int transpose, int ck,
         int mt, const TapSel& sel, hipStream_t st) {
  const int M = transpose ? Cin : Cout, Kc = transpose ? Cout : Cin;
  const int bm = 32 * mt;
  const size_t n = (size_t)((M + bm - 1) / bm * bm) * ((Kc + ck - 1) / ck * ck) * sel.n;  // (a multiple of 32)
  if (g_pk.mode == PK_SIZE) { g_pk.used += n; return AIR_OK; }
  if (g_pk.mode == PK_USE) { wp = g_pk.cur + g_pk.used; g_pk.used += n; return AIR_OK; }
  if (g_pk.mode == PK_COLLECT) { wp = g_pk.cur + g_pk.used; g_pk.used += n; }
  if (g_pk_defer && g_pk.mode != PK_RUN) {  // (only slabs that go to a caller's buffer: a workspace slab is consumed at once)
    if (g_pk_jobs.n == PK_JOBS) {
      const int rc = pack_run_jobs(st);
      if (rc != AIR_OK) return rc;
      g_pk_jobs.blk0[0] = 0;
    }
    const int j = g_pk_jobs.n++;
    g_pk_jobs.w[j] = w; g_pk_jobs.wp[j] = wp; g_pk_jobs.Cout[j] = Cout; g_pk_jobs.Cin[j] = Cin;
    g_pk_jobs.taps_full[j] = (unsigned char)taps_full; g_pk_jobs.transpose[j] = (unsigned char)transpose;
    g_pk_jobs.ck[j] = (unsigned char)ck; g_pk_jobs.bm[j] = (unsigned char)bm;
    g_pk_jobs.seln[j] = (unsigned char)sel.n;
    for (int t = 0; t < 9; ++t) g_pk_jobs.selidx[j][t] = (unsigned char)(t < sel.n ? sel.idx[t] : 0);
    g_pk_jobs.blk0[j + 1] = g_pk_jobs.blk0[j] + grid_for(n);
    return AIR_OK;
  }
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(n)), dim3(256), 0, st, w, wp, Cout, Cin,
                     taps_full, transpose, ck, bm, sel);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

// algorithmic FLOPs (2 * MACs) of the convolution, whatever kernel computes it
double conv_flops(const AirConv2d* p) {
  return 2.0 * p->B * p->Cout * p->Ho * p->Wo * (double)p->Cin * p->KH * p->KW;
}

// stride-2 3x3 wgrad holds a 65-column patch per channel: use 32-channel tiles there
int wgrad_ct(const AirConv2d* p) { return (p->sh == 2 && p->Cout % 128 == 0) ? 32 : 64; }

// workgroups of the split-K direct weight gradient: one per CU, two for the 1x1 stride-2 layers (16 MFMAs between
// the barriers of a tile and 50 KB of LDS: a second resident workgroup covers the waits, 240 -> 212 us over the
// ResNet's three shortcuts)
int wgrad_wgs(int KH, int KW, int S) { return air_opt(AIR_OPT_WGRAD_WGS) * ((KH == 1 && KW == 1 && S == 2) ? 2 : 1); }

int wgrad_nsplit(const AirConv2d* p) {
  const int WT = (p->Wo + PXT - 1) / PXT;
  const int ntiles = p->B * p->Ho * WT;
  const int ct = wgrad_ct(p);
  const int ncot = p->Cout / (ct == 32 ? 128 : 64), ncit = (p->Cin + ct - 1) / ct;
  const int total = wgrad_wgs(p->KH, p->KW, p->sh);
  int target = total / (ncot * ncit);  // workgroups in flight over the whole chip
  if (target < 1) target = 1;
  if (target > ntiles) target = ntiles;
  return target;
}

size_t packed_dgrad_elems(const AirConv2d* p) {  // dgrad packs with Cin padded to 64
  return (size_t)((p->Cin + BM - 1) / BM * BM) * ((p->Cout + 31) / 32 * 32) * p->KH * p->KW;
}


// ---- 1x1 stride-1 weight gradient of a NARROW layer (the ResNet's 16 -> 64 shortcut, resnet.py:62): a
// (Cout x Cin) = 64 x 16 output contracted over B H W = 864,000 pixels - 276 MB of operands for 1.8 GFLOP, i.e.
// a streaming job (55 us at 5 TB/s) that the 64-channel-tile kernel above ran at 6 TFLOP/s.  Here the pixels are
// the k index of v_mfma_f32_16x16x4_f32: a lane owns one channel row of a 16-row block and four consecutive
// pixels (one 16-byte load per block), the four values feed four MFMAs (k = the lane's quarter, the same pixel on
// the A and the B side), so a wave turns MB + NB loads into 4 MB NB MFMAs.  Workgroup = (utterance, pixel chunk),
// its four waves split the chunk; their sums meet in LDS in wave order and go out as one partial per workgroup
// (reduce_partials_kernel).  PRO: x' = relu(x scale[ci] + shift[ci]) applied as the values arrive (the BN + ReLU
// in front of the shortcut conv).
struct SkinnyWg {
  const float* x;
  const float* dy;
  float* partial;
  const float* scale;
  const float* shift;
  size_t x_bs, dy_bs;
  int HW, chunk, nchunk;
};

template <int MB, int NB, bool PRO>
__global__ __launch_bounds__(256) void conv_wgrad_1x1_skinny_kernel(const SkinnyWg a) {
  __shared__ float red[4][MB * NB * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.nchunk, ch = blockIdx.x - b * a.nchunk;
  const int r = lane & 15, kq = lane >> 4;
  const int per_wave = a.chunk / 4;  // a multiple of 16
  const int p_lo = ch * a.chunk + wave * per_wave;
  const int p_hi = min(p_lo + per_wave, a.HW);
  const float* __restrict__ xb = a.x + (size_t)b * a.x_bs;
  const float* __restrict__ yb = a.dy + (size_t)b * a.dy_bs;
  float sc[NB], sh[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    sc[nb] = PRO ? a.scale[nb * 16 + r] : 1.0f;
    sh[nb] = PRO ? a.shift[nb * 16 + r] : 0.0f;
  }
  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int p0 = p_lo; p0 < p_hi; p0 += 32) {  // two 16-pixel steps per trip: 2 (MB + NB) loads in flight
    f32x4 av[2][MB], bv[2][NB];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = p0 + 16 * u + 4 * kq;
      const bool ok = p < p_hi;  // HW % 4 == 0: a lane's four pixels are in or out together
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        av[u][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (ok) av[u][mb] = *reinterpret_cast<const f32x4*>(yb + (size_t)(mb * 16 + r) * a.HW + p);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        bv[u][nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (ok) bv[u][nb] = *reinterpret_cast<const f32x4*>(xb + (size_t)(nb * 16 + r) * a.HW + p);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (PRO) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[u][nb][e] = fmaxf(fmaf(bv[u][nb][e], sc[nb], sh[nb]), 0.0f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mb][e], bv[u][nb][e], acc[mb][nb], 0, 0, 0);
    }
  }
  // D[i = 4 (lane / 16) + v][j = lane % 16]
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave][((mb * 16 + 4 * kq + v) * NB + nb) * 16 + r] = acc[mb][nb][v];
  __syncthreads();
  float* __restrict__ out = a.partial + (size_t)blockIdx.x * (MB * NB * 256);
  for (int e = tid; e < MB * NB * 256; e += 256) out[e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

// ---- 3x3 / stride 1 / pad 1 weight gradient of the same narrow layer shape (16 -> 64: the ResNet's layer1.0.conv1,
// resnet.py:58): nine (64 x 16) products over the pixels, tap (dh, dw) contracting dy[h][w] with
// x[h + dh - 1][w + dw - 1].  Same pixels-as-k mapping as conv_wgrad_1x1_skinny_kernel: a lane owns one channel row
// of a 16-row block and four consecutive pixels of ONE image row; per input row (h - 1, h, h + 1) it loads its four
// pixels and the two neighbours and forms the three column shifts in registers (zeros outside the image - applied
// AFTER the optional BN + ReLU prologue), so 4 dy loads + 9 x loads feed 9 x 16 MFMAs: the kernel runs at the
// f32 MFMA rate of the direct algorithm (15.9 GFLOP, ~0.1 ms) where the Winograd weight-gradient kernel needed the
// input zero-padded to 64 channels (a 221 MB copy and 4x the products).  The waves of one workgroup per CU deal the
// (utterance, row, quarter-row) units among themselves; sums meet in LDS three taps at a time; one partial
// [tap][co][ci] per workgroup.
struct Skinny3Wg {
  const float* x;
  const float* dy;
  float* partial;
  const float* scale;
  const float* shift;
  size_t x_bs, dy_bs;
  int B, H, W;
};

template <int MB, bool PRO>
__global__ __launch_bounds__(256) void conv_wgrad_3x3_skinny_kernel(const Skinny3Wg a) {
  __shared__ float red[4][3 * MB * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int W = a.W, H = a.H;
  const float sc = PRO ? a.scale[r] : 1.0f, sh = PRO ? a.shift[r] : 0.0f;
  f32x4 acc[9][MB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[t][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // one 16-pixel step: this lane's four dy pixels of MB channel rows and its 3 x 6 input neighbourhood
  auto load_step = [&](int b, int h, int wb, f32x4 (&av)[MB], float (&xv)[3][6]) __attribute__((always_inline)) {
    const int w0 = wb + 4 * kq;
    const float* __restrict__ yb = a.dy + (size_t)b * a.dy_bs + (size_t)h * W + w0;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float* __restrict__ src = yb + (size_t)(mb * 16 + r) * H * W;
      if (w0 + 3 < W) {
        av[mb] = *reinterpret_cast<const f32x4*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) av[mb][e] = w0 + e < W ? src[e] : 0.0f;
      }
    }
    const float* __restrict__ xr = a.x + (size_t)b * a.x_bs + (size_t)r * H * W + w0;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hh = h + dh - 1;
      const bool rok = hh >= 0 && hh < H;
      const float* __restrict__ src = xr + (size_t)(rok ? hh : 0) * W;
#pragma unroll
      for (int e = 0; e < 6; ++e) xv[dh][e] = 0.0f;
      if (rok) {
        if (w0 + 3 < W) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
          xv[dh][1] = v[0]; xv[dh][2] = v[1]; xv[dh][3] = v[2]; xv[dh][4] = v[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (w0 + e < W) xv[dh][1 + e] = src[e];
        }
        if (w0 > 0 && w0 - 1 < W) xv[dh][0] = src[-1];
        if (w0 + 4 < W) xv[dh][5] = src[4];
      }
    }
  };
  auto mfma_step = [&](int h, int wb, const f32x4 (&av)[MB], float (&xv)[3][6]) __attribute__((always_inline)) {
    if (PRO) {  // BN + ReLU on the values that exist; the zero padding stays zero
      const int w0 = wb + 4 * kq;
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int hh = h + dh - 1;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          const int wc = w0 - 1 + e;
          xv[dh][e] = (hh >= 0 && hh < H && wc >= 0 && wc < W) ? fmaxf(fmaf(xv[dh][e], sc, sh), 0.0f) : 0.0f;
        }
      }
    }
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc[dh * 3 + dw][mb] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb][e], xv[dh][e + dw], acc[dh * 3 + dw][mb], 0, 0, 0);
  };

  // units = (utterance, image row, quarter of the row), dealt round-robin over all waves of the launch; the loads of
  // a step are issued before the MFMAs of the step before it
  const int spr = (W + 15) / 16, spq = (spr + 3) / 4;  // steps per row / per quarter
  const int units = a.B * H * 4;
  const int nwaves = gridDim.x * 4;
  // two steps of loads in flight: slot 0 is multiplied while slots 1 and 2 are on their way
  f32x4 av0[MB], av1[MB], av2[MB];
  float xv0[3][6], xv1[3][6], xv2[3][6];
  int have = 0, h0 = 0, wb0 = 0, h1 = 0, wb1 = 0;
  auto shift = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) { av0[mb] = av1[mb]; av1[mb] = av2[mb]; }
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int e = 0; e < 6; ++e) { xv0[dh][e] = xv1[dh][e]; xv1[dh][e] = xv2[dh][e]; }
    h0 = h1; wb0 = wb1;
  };
  for (int u = blockIdx.x * 4 + wave; u < units; u += nwaves) {
    const int q = u & 3, bh = u >> 2;
    const int b = bh / H, h = bh - b * H;
    const int s_lo = q * spq, s_hi = min(spr, s_lo + spq);
    for (int st = s_lo; st < s_hi; ++st) {
      load_step(b, h, st * 16, av2, xv2);
      if (have == 2) mfma_step(h0, wb0, av0, xv0);
      shift();
      h1 = h; wb1 = st * 16;
      if (have < 2) ++have;
    }
  }
  // drain: slot 1 holds the last step loaded, slot 0 the one before it (when there were two)
  if (have == 2) mfma_step(h0, wb0, av0, xv0);
  if (have >= 1) mfma_step(h1, wb1, av1, xv1);

  float* __restrict__ out = a.partial + (size_t)blockIdx.x * (9 * MB * 256);
#pragma unroll
  for (int tg = 0; tg < 3; ++tg) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int v = 0; v < 4; ++v) red[wave][(t * MB * 16 + mb * 16 + 4 * kq + v) * 16 + r] = acc[tg * 3 + t][mb][v];
    __syncthreads();
    for (int e = tid; e < 3 * MB * 256; e += 256)
      out[tg * 3 * MB * 256 + e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    __syncthreads();
  }
}

struct WgradGeom {
  int B, Cin, H, W, Cout, KH, KW, S, ph, pw, Ho, Wo;
  int dil;
  size_t x_bstride, dy_bstride;
};

int run_wgrad(const WgradGeom& g, const float* x, const float* dy, float* dw, const float* in_scale,
              const float* in_shift, int relu, void* ws, size_t ws_bytes, double flops,
              hipStream_t st) {
  const size_t wsz = (size_t)g.Cout * g.Cin * g.KH * g.KW;
  WgradArgs a;
  a.x = x; a.dy = dy; a.partial = reinterpret_cast<float*>(ws);
  a.scale = in_scale; a.shift = in_shift;
  a.B = g.B; a.Cin = g.Cin; a.H = g.H; a.W = g.W; a.Cout = g.Cout; a.Ho = g.Ho; a.Wo = g.Wo;
  a.ph = g.ph; a.pw = g.pw; a.relu = relu;
  a.x_bstride = g.x_bstride; a.dy_bstride = g.dy_bstride;
  a.WT = (g.Wo + PXT - 1) / PXT;
  a.ntiles = g.B * g.Ho * a.WT;
  const int ct = (g.S == 2 && g.Cout % 128 == 0) ? 32 : 64;
  a.ncot = g.Cout / (ct == 32 ? 128 : 64);
  a.ncit = (g.Cin + ct - 1) / ct;
  const int total = g.dil == 1 ? wgrad_wgs(g.KH, g.KW, g.S) : air_opt(AIR_OPT_WGRAD_WGS);
  int nsplit = total / (a.ncot * a.ncit);  // workgroups in flight over the whole chip
  if (nsplit < 1) nsplit = 1;
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  a.nsplit = nsplit;
  a.tiles_per_split = (a.ntiles + a.nsplit - 1) / a.nsplit;
  if (!ws || ws_bytes < (size_t)a.nsplit * wsz * sizeof(float)) return AIR_EWORKSPACE;
  const int nblk = a.ncot * a.ncit * a.nsplit;
  const int key = g.KH * 1000 + g.KW * 100 + g.S * 10 + (g.dil - 1);
#define AIR_WG_CASE(KH_, KW_, S_, CT_, DIL_, KID_)                                              \
  if (key == KH_ * 1000 + KW_ * 100 + S_ * 10 + (DIL_ - 1) && ct == CT_) {                       \
    AirProfScope ps(KID_, flops, st);                                                            \
    if (a.scale != nullptr)                                                                      \
      hipLaunchKernelGGL((conv_wgrad_kernel<KH_, KW_, S_, CT_, 1, DIL_>), dim3(nblk), dim3(256), \
                         0, st, a);                                                              \
    else                                                                                         \
      hipLaunchKernelGGL((conv_wgrad_kernel<KH_, KW_, S_, CT_, 0, DIL_>), dim3(nblk), dim3(256), \
                         0, st, a);                                                              \
    AIR_CHECK_LAUNCH();                                                                          \
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_grid(wsz, a.nsplit)), dim3(256), 0, st, \
                       reinterpret_cast<const float*>(ws), dw, wsz, a.nsplit, g.KH * g.KW);      \
    AIR_CHECK_LAUNCH();                                                                          \
    return AIR_OK;                                                                               \
  }
  AIR_WG_CASE(3, 3, 1, 64, 1, AIR_K_CONV_WG_331)
  AIR_WG_CASE(3, 3, 2, 32, 1, AIR_K_CONV_WG_332)
  AIR_WG_CASE(1, 1, 1, 64, 1, AIR_K_CONV_WG_111)
  AIR_WG_CASE(1, 1, 2, 32, 1, AIR_K_CONV_WG_112)
  AIR_WG_CASE(1, 5, 1, 64, 1, AIR_K_CONV_WG_1D)
  AIR_WG_CASE(1, 3, 1, 64, 2, AIR_K_CONV_WG_1D)
  AIR_WG_CASE(1, 3, 1, 64, 3, AIR_K_CONV_WG_1D)
  AIR_WG_CASE(1, 3, 1, 64, 4, AIR_K_CONV_WG_1D)
#undef AIR_WG_CASE
  return AIR_EUNSUPPORTED;
}

}  // namespace

extern "C" {

// A 3x3 / stride-1 layer with fewer than 64 input channels (resnet.py:56 in layer1.0: 16 -> 64) wastes three
// quarters of the direct weight-gradient kernel's 64-channel tile (20 TF).  Zero-padding x to 64 channels and
// running the Winograd weight-gradient kernel wastes the same fraction of 2.25x fewer MFMAs at a much higher
// rate: 0.78 -> 0.45 ms at B = 64.
static bool wino_pad_wgrad(const AirConv2d* p) {
  return wino_shape(p) && p->Cin < 64 && p->Cin % 8 == 0 && p->Cout % 64 == 0 &&
         air_wino_wgrad_ok(p->B, 64, p->H, p->W, p->Cout);
}

// narrow 1x1 / stride 1 layers served by conv_wgrad_1x1_skinny_kernel: pixel chunks per utterance (0 = not served)
static int skinny_wgrad_chunks(const AirConv2d* p) {
  if (p->KH != 1 || p->KW != 1 || p->sh != 1 || p->sw != 1 || p->ph != 0 || p->pw != 0) return 0;
  if (p->Cout != 64 || p->Cin != 16 || (p->H * p->W) % 4 != 0 || !air_opt(AIR_OPT_SKINNY_WGRAD)) return 0;
  const int hw = p->H * p->W;
  int n = 512 / p->B;  // ~2 workgroups per CU over the chip
  if (n < 1) n = 1;
  while (n > 1 && hw / n < 256) --n;
  return n;
}
static bool skinny3_wgrad_ok(const AirConv2d* p) {
  return p->KH == 3 && p->KW == 3 && p->sh == 1 && p->sw == 1 && p->ph == 1 && p->pw == 1 && p->Cout == 64 &&
         p->Cin == 16 && p->W >= 4 && air_opt(AIR_OPT_SKINNY_WGRAD);
}
static int skinny3_wgrad_parts(const AirConv2d* p) {
  const int units = p->B * p->H;  // one workgroup (4 waves, a row's quarters each) per row, at most one per CU
  return units < 256 ? units : 256;
}
static int skinny_wgrad_chunk_px(const AirConv2d* p, int nchunk) {
  const int hw = p->H * p->W;
  return ((hw + nchunk - 1) / nchunk + 63) / 64 * 64;
}

// the split-bf16 forward of a 3x3 / stride 2 / pad 1 layer (conv_bf3.hip; option CONV_S2 bit 4)
static bool bf3_fwd_ok(const AirConv2d* p) {
  return p->KH == 3 && p->KW == 3 && p->sh == 2 && p->sw == 2 && p->ph == 1 && p->pw == 1 && generic_ok(p) && !direct_ok(p) &&
         air_bf3_s2_ok(p->B, p->Cin, p->H, p->W, p->Cout);
}

static bool bf3_wgrad_ok(const AirConv2d* p) {
  return p->KH == 3 && p->KW == 3 && p->sh == 2 && p->sw == 2 && p->ph == 1 && p->pw == 1 && generic_ok(p) && !direct_ok(p) &&
         air_bf3_s2w_ok(p->B, p->Cin, p->H, p->W, p->Cout);
}

size_t air_conv2d_ws_bytes(const AirConv2d* p) {
  if (!p || !shape_ok(p)) return 0;
  const size_t wsz = (size_t)p->Cout * p->Cin * p->KH * p->KW;
  if (direct_ok(p)) return (size_t)p->B * p->Ho * wsz * sizeof(float) + 256;
  if (!generic_ok(p)) return 0;
  size_t fwd = wsz;
  {  // K-split forward: the weight slab and up to four partial copies of y behind it (fwd_generic)
    const size_t slab = (size_t)((p->Cout + 63) / 64 * 64) * ((p->Cin + 31) / 32 * 32) * p->KH * p->KW + 64;
    const int ntiles = p->B * p->Ho * ((p->Wo + PXT - 1) / PXT);
    if (((ntiles + NWAVE - 1) / NWAVE) * ((p->Cout + 31) / 32) <= 1536)
      fwd = slab + 4 * (size_t)p->B * p->Cout * p->Ho * p->Wo;
  }
  if (bf3_fwd_ok(p)) {  // bf16 planes: 6 bytes per weight
    const size_t pl = (air_bf3_s2_packed_bytes(p->Cout, p->Cin, true) + 3) / 4;
    if (pl > fwd) fwd = pl;
  }
  size_t dgrad = packed_dgrad_elems(p);
  if (wino_shape(p)) {  // transformed weights are 16/9 the size
    const size_t wf = air_wino_packed_elems(p->Cout, p->Cin), wd = air_wino_packed_elems(p->Cin, p->Cout);
    if (wf > fwd) fwd = wf;
    if (wd > dgrad) dgrad = wd;
    const size_t wf4 = air_wino4_packed_elems(p->Cout, p->Cin), wd4 = air_wino4_packed_elems(p->Cin, p->Cout);
    if (wf4 > fwd) fwd = wf4;  // F(4x4,3x3): 36/9 the size
    if (wd4 > dgrad) dgrad = wd4;
  }
  size_t wgrad = (size_t)wgrad_nsplit(p) * wsz;
  if (wino_shape(p) && air_wino_wgrad_ok(p->B, p->Cin, p->H, p->W, p->Cout)) {
    const size_t ww = (size_t)air_wino_wgrad_nsplit(p->B, p->Cin, p->H, p->W, p->Cout) * wsz;
    if (ww > wgrad) wgrad = ww;
  }
  if (wino_pad_wgrad(p)) {  // zero-padded copy of x (64 channels) + partials + 64-channel dW
    const size_t pad = (size_t)p->B * 64 * p->H * p->W + 64 +
                       (size_t)(air_wino_wgrad_nsplit(p->B, 64, p->H, p->W, p->Cout) + 1) * p->Cout * 64 * 9;
    if (pad > wgrad) wgrad = pad;
  }
  if (const int nch = skinny_wgrad_chunks(p)) {
    const size_t sk = (size_t)p->B * nch * wsz;
    if (sk > wgrad) wgrad = sk;
  }
  if (bf3_wgrad_ok(p)) {
    const size_t sk = (size_t)air_bf3_s2w_nseg(p->B, p->Cin, p->Ho, p->Wo, p->Cout) * wsz;
    if (sk > wgrad) wgrad = sk;
  }
  if (skinny3_wgrad_ok(p)) {
    const size_t sk = (size_t)skinny3_wgrad_parts(p) * wsz;
    if (sk > wgrad) wgrad = sk;
  }
  size_t m = fwd > dgrad ? fwd : dgrad;
  if (wgrad > m) m = wgrad;
  return m * sizeof(float) + 256;
}

// which Winograd kernel takes this 3x3 / stride 1 / pad 1 layer for `pass` (0 forward, 1 dgrad): 4, 2 or 0 (none)
static int wino_kind(const AirConv2d* p, int pass) {
  if (!wino_shape(p) || !generic_ok(p)) return 0;
  const int M = pass ? p->Cin : p->Cout, Kc = pass ? p->Cout : p->Cin;
  if (air_wino4_ok(p->B, Kc, p->H, p->W, M)) return 4;
  if (air_wino_ok(p->B, Kc, p->H, p->W, M)) return 2;
  return 0;
}

// floats of the forward weight slab (channel tiles and K chunks zero-padded), rounded up to 64
static size_t fwd_pack_elems(const AirConv2d* p, int ck, int mt) {
  const int bm = 32 * mt;
  const size_t n = (size_t)((p->Cout + bm - 1) / bm * bm) * ((p->Cin + ck - 1) / ck * ck) * p->KH * p->KW;
  return (n + 63) / 64 * 64;
}
// K slices of the plain forward (1 = none).  A layer with few pixel x channel tiles and a long K loop - conv5 of the
// ResNet (resnet.py:140): 384 workgroups of 64 chunks on 256 CUs, one or two waves per SIMD with every chunk-end
// DMA wait exposed - runs as up to four times as many workgroups of a quarter of the loop, summed in slice order.
static int fwd_ksplit(const AirConv2d* p, int ck, int mt) {
  const int ntiles = p->B * p->Ho * ((p->Wo + PXT - 1) / PXT);
  const int wgs = ((ntiles + NWAVE - 1) / NWAVE) * ((p->Cout + 32 * mt - 1) / (32 * mt));
  const int nchunk = (p->Cin + ck - 1) / ck;
  if (p->Cin % ck != 0) return 1;
  int k = 1536 / (wgs > 0 ? wgs : 1);
  if (k > 4) k = 4;
  while (k > 1 && nchunk / k < 8) --k;
  return k < 1 ? 1 : k;
}

// forward on the direct kernels (whatever pack mode is current: see PackCtx)
static int fwd_generic(const AirConv2d* p, const float* x, const float* w, float* y, const float* in_scale,
                       const float* in_shift, int relu, const float* residual, float* wp, hipStream_t st,
                       size_t ws_floats = 0) {
  const int taps = p->KH * p->KW;
  // stride-2 3x3: the 65-column patches of 8 channels leave room for ONE workgroup per CU (88 KB of LDS);
  // 4-channel chunks fit three (option CONV_S2, bit 1)
  const bool s2ck4 = taps == 9 && p->sh == 2 && p->Cin % 4 == 0 && (air_opt(AIR_OPT_CONV_S2) & 1);
  // 1x1 stride 2 (the shortcuts): 32-channel chunks of 63-column rows are 80 KB - one workgroup per CU again; 16 fit three
  const bool s2ck16 = taps == 1 && p->sh == 2 && p->Cin % 16 == 0 && (air_opt(AIR_OPT_CONV_S2) & 1);
  const int ck = s2ck4 ? 4 : (s2ck16 ? 16 : pick_ck(taps, p->Cin));
  TapSel sel;
  sel.n = taps;
  for (int t = 0; t < taps; ++t) sel.idx[t] = t;
  // (with 4-channel chunks the 64-channel tile measured faster on every ResNet shape, whatever the round count)
  const int mt = s2ck4 && p->Cout > 32 && !air_opt(AIR_OPT_CONV_MT) ? 2 : mt_for(p->B, p->Ho, p->Wo, p->Cout);
  float* const ws0 = wp;  // (pack() points wp at the caller's slab under PK_USE)
  int rc = pack(w, wp, p->Cout, p->Cin, taps, 0, ck, mt, sel, st);
  if (rc != AIR_OK) return rc;
  int ksplit = (in_scale == nullptr && residual == nullptr && !pk_dry()) ? fwd_ksplit(p, ck, mt) : 1;
  // (a caller that sized its workspace by an older rule gets the unsplit loop)
  if (ksplit > 1 && ws_floats < fwd_pack_elems(p, ck, mt) + (size_t)ksplit * p->B * p->Cout * p->Ho * p->Wo) ksplit = 1;
  return run_fwd(x, wp, y, in_scale, in_shift, relu, residual,
                 plain_geom(p->B, p->Cin, p->H, p->W, p->Cout, p->KH, p->KW, p->sh, p->ph, p->pw,
                            p->Ho, p->Wo),
                 ck, mt, conv_flops(p), st, ksplit, ksplit > 1 ? ws0 + fwd_pack_elems(p, ck, mt) : nullptr);
}

// stride-2 3x3 data gradient in one pass (conv_s2_dgrad_kernel); option CONV_S2 bit 2, 0 = the four class launches
constexpr int S2D_CK = 8;
static bool s2d_ok(const AirConv2d* p) {
  return p->KH == 3 && p->KW == 3 && p->sh == 2 && p->sw == 2 && p->ph == 1 && p->pw == 1 && p->Cout % S2D_CK == 0 &&
         (air_opt(AIR_OPT_CONV_S2) & 2) != 0;
}
static int s2d_mt(const AirConv2d* p) { return mt_for(p->B, p->Ho, p->Wo, p->Cin); }

// wp / wpsc: the 3x3 and the 1x1 weights as pack() lays them out for (S2D_CK, mt), roles swapped; dysc / wpsc null =
// no shortcut term
static int run_s2d(const AirConv2d* p, const float* dy, const float* dysc, const float* wp, const float* wpsc, float* dx,
                   const float* accumulate, int mt, hipStream_t st) {
  if (pk_dry()) return AIR_OK;
  S2dArgs a;
  a.dy = dy; a.dysc = dysc; a.wp = wp; a.wpsc = wpsc; a.dx = dx; a.accumulate = accumulate;
  a.B = p->B; a.K = p->Cout; a.M = p->Cin; a.H = p->H; a.W = p->W; a.Ho = p->Ho; a.Wo = p->Wo;
  a.WT = (p->Wo + PXT - 1) / PXT;
  a.ntiles = p->B * p->Ho * a.WT;
  a.npxg = (a.ntiles + NWAVE - 1) / NWAVE;
  a.ncot = (p->Cin + 32 * mt - 1) / (32 * mt);
  a.pair = (p->W % 2 == 0) && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(accumulate)) & 7) == 0;
  const int nblk = a.npxg * a.ncot;
  const double flops = conv_flops(p) * (dysc ? 10.0 / 9.0 : 1.0);
  AirProfScope ps(AIR_K_CONV_S2_DGRAD, flops, st);
#define AIR_S2D(MT_, SC_) \
  hipLaunchKernelGGL((conv_s2_dgrad_kernel<S2D_CK, MT_, SC_>), dim3(nblk), dim3(NWAVE * 64), 0, st, a)
  if (mt == 2) {
    if (dysc) AIR_S2D(2, true); else AIR_S2D(2, false);
  } else {
    if (dysc) AIR_S2D(1, true); else AIR_S2D(1, false);
  }
#undef AIR_S2D
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

// data gradient on the direct kernels (whatever pack mode is current: see PackCtx)
static int dgrad_generic(const AirConv2d* p, const float* dy, const float* w, float* dx, const float* accumulate,
                         float* wp, hipStream_t st) {
  const int taps = p->KH * p->KW;
  if (p->sh == 1 && p->Ho == 1 && p->ph == 0 && p->KH == p->H && p->KH > 1 && p->KW == 3 && p->Cout % 16 == 0) {
    // The kernel spans the whole image height (resnet.py:140 conv5: (num_nodes, 3) taps, no vertical padding,
    // one output row): input row h only ever meets kernel row h.  As ONE KH x 3 convolution over the
    // zero-padded single dy row, (KH-1)/KH of the MFMAs would multiply padding; instead every kernel row is
    // its own dense 1 x 3 convolution of dy that writes row h of dx.
    const size_t plane = (size_t)p->H * p->W;
    const int ck = pick_ck(3, p->Cout);
    const int mt = mt_for(p->B, 1, p->W, p->Cin);
    for (int h = 0; h < p->KH; ++h) {
      TapSel sel;
      sel.n = 3;
      for (int j = 0; j < 3; ++j) sel.idx[j] = h * 3 + (2 - j);  // flipped along w
      int rc = pack(w, wp, p->Cout, p->Cin, taps, 1, ck, mt, sel, st);
      if (rc != AIR_OK) return rc;
      FwdGeom g = plain_geom(p->B, p->Cout, 1, p->Wo, p->Cin, 1, 3, 1, 0, 2 - p->pw, 1, p->W);
      g.oh_mul = 1; g.ow_row = p->W; g.ow_mul = 1; g.o_off = h * p->W; g.oplane = plane;
      g.y_bstride = (size_t)p->Cin * plane;
      rc = run_fwd(dy, wp, dx, nullptr, nullptr, 0, accumulate, g, ck, mt, conv_flops(p) / p->KH, st);
      if (rc != AIR_OK) return rc;
    }
    return AIR_OK;
  }
  if (p->sh == 1) {
    // stride 1: dx = conv(dy, flipped taps) with padding K-1-p
    const int ck = pick_ck(taps, p->Cout);
    TapSel sel;
    sel.n = taps;
    for (int t = 0; t < taps; ++t) sel.idx[t] = taps - 1 - t;
    const int mt = mt_for(p->B, p->H, p->W, p->Cin);
    int rc = pack(w, wp, p->Cout, p->Cin, taps, 1, ck, mt, sel, st);
    if (rc != AIR_OK) return rc;
    return run_fwd(dy, wp, dx, nullptr, nullptr, 0, accumulate,
                   plain_geom(p->B, p->Cout, p->Ho, p->Wo, p->Cin, p->KH, p->KW, 1,
                              p->KH - 1 - p->ph, p->KW - 1 - p->pw, p->H, p->W),
                   ck, mt, conv_flops(p), st);
  }
  // stride 2: input pixel (2i+a, 2j+b) only sees the taps with kh = (a+1+ph') parity etc.
  // Each of the 4 parity classes is a dense stride-1 conv of dy with 1, 2, 2 or 4 taps that
  // writes its own interleaved quarter of dx: no zero-upsampling, no wasted MFMAs.
  if (!((p->KH == 3 && p->ph == 1) || (p->KH == 1 && p->ph == 0)) || p->ph != p->pw)
    return AIR_EUNSUPPORTED;
  const size_t plane = (size_t)p->H * p->W;
  const size_t dx_elems = (size_t)p->B * p->Cin * plane;
  if (p->KH == 1) {
    // 1x1 stride 2: only even/even pixels receive gradient
    if (pk_dry()) {
      // (weights only: nothing touches dx)
    } else if (accumulate == nullptr) {
      if (hipMemsetAsync(dx, 0, dx_elems * sizeof(float), st) != hipSuccess) return AIR_ELAUNCH;
    } else if (accumulate != dx) {
      if (hipMemcpyAsync(dx, accumulate, dx_elems * sizeof(float), hipMemcpyDeviceToDevice, st) !=
          hipSuccess)
        return AIR_ELAUNCH;
    }
    const int ck = pick_ck(1, p->Cout);
    TapSel sel;
    sel.n = 1;
    sel.idx[0] = 0;
    const int mt = mt_for(p->B, p->Ho, p->Wo, p->Cin);
    int rc = pack(w, wp, p->Cout, p->Cin, 1, 1, ck, mt, sel, st);
    if (rc != AIR_OK) return rc;
    FwdGeom g = plain_geom(p->B, p->Cout, p->Ho, p->Wo, p->Cin, 1, 1, 1, 0, 0, p->Ho, p->Wo);
    g.oh_mul = 2; g.ow_row = p->W; g.ow_mul = 2; g.o_off = 0; g.oplane = plane;
    g.y_bstride = (size_t)p->Cin * plane;
    return run_fwd(dy, wp, dx, nullptr, nullptr, 0, dx, g, ck, mt, conv_flops(p), st);
  }
  if (s2d_ok(p)) {
    TapSel sel;
    sel.n = 9;
    for (int t = 0; t < 9; ++t) sel.idx[t] = t;
    const int mt = s2d_mt(p);
    int rc = pack(w, wp, p->Cout, p->Cin, 9, 1, S2D_CK, mt, sel, st);
    if (rc != AIR_OK) return rc;
    return run_s2d(p, dy, nullptr, wp, nullptr, dx, accumulate, mt, st);
  }
  for (int a = 0; a < 2; ++a) {
    for (int b = 0; b < 2; ++b) {
      // class (a, b): rows h = 2i + a.  a == 0: kh = 1 reads dy[i]; a == 1: kh = 2 reads dy[i],
      // kh = 0 reads dy[i+1].  Same along w.
      const int nh = a ? 2 : 1, nw = b ? 2 : 1;
      const int khs[2] = {a ? 2 : 1, 0}, kws[2] = {b ? 2 : 1, 0};
      const int Hc = (p->H - a + 1) / 2, Wc = (p->W - b + 1) / 2;
      if (Hc <= 0 || Wc <= 0) continue;
      TapSel sel;
      sel.n = nh * nw;
      for (int i = 0; i < nh; ++i)
        for (int j = 0; j < nw; ++j) sel.idx[i * nw + j] = khs[i] * 3 + kws[j];
      const int ck = pick_ck(sel.n, p->Cout);
      const int mt = mt_for(p->B, Hc, Wc, p->Cin);
      int rc = pack(w, wp, p->Cout, p->Cin, 9, 1, ck, mt, sel, st);
      if (rc != AIR_OK) return rc;
      FwdGeom g = plain_geom(p->B, p->Cout, p->Ho, p->Wo, p->Cin, nh, nw, 1, 0, 0, Hc, Wc);
      g.oh_mul = 2; g.ow_row = p->W; g.ow_mul = 2; g.o_off = a * p->W + b; g.oplane = plane;
      g.y_bstride = (size_t)p->Cin * plane;
      rc = run_fwd(dy, wp, dx, nullptr, nullptr, 0, accumulate, g, ck, mt,
                   conv_flops(p) * sel.n / 9.0, st);
      if (rc != AIR_OK) return rc;
    }
  }
  return AIR_OK;
}

// (direct_ok: the ResNet's 1 -> 16 first layer reads its weights as they are)
static bool generic_pack_ok(const AirConv2d* p, int pass) {
  return generic_ok(p) && !direct_ok(p) && wino_kind(p, pass) == 0;
}

size_t air_conv2d_prepack_bytes(const AirConv2d* p, int pass) {
  if (!p || !shape_ok(p)) return 0;
  const int kind = wino_kind(p, pass);
  const int M = pass ? p->Cin : p->Cout, Kc = pass ? p->Cout : p->Cin;
  if (kind == 4) return air_wino4_packed_elems(M, Kc) * sizeof(float);
  if (kind == 2) return air_wino_packed_elems(M, Kc) * sizeof(float);
  if (pass == 0 && bf3_fwd_ok(p)) return air_bf3_s2_packed_bytes(p->Cout, p->Cin, false);
  if (!generic_pack_ok(p, pass)) return 0;
  PackScope size(PK_SIZE, nullptr);
  float* none = nullptr;
  const int rc = pass ? dgrad_generic(p, nullptr, nullptr, nullptr, nullptr, none, nullptr)
                      : fwd_generic(p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, none, nullptr);
  return rc == AIR_OK ? g_pk.used * sizeof(float) : 0;
}

// Which layout air_conv2d_prepack(p, ., pass) writes under the CURRENT dispatch options (a buffer packed under
// other options - or the split-bf16 planes handed to a call that takes the f32 slabs - must be refused, not walked).
int air_conv2d_prepack_layout(const AirConv2d* p, int pass) {
  if (!p || !shape_ok(p)) return AIR_PACK_NONE;
  const int kind = wino_kind(p, pass);
  if (kind == 4) return AIR_PACK_WINO4;
  if (kind == 2) return AIR_PACK_WINO2;
  if (pass == 0 && bf3_fwd_ok(p)) return AIR_PACK_BF3;
  return generic_pack_ok(p, pass) ? AIR_PACK_F32_SLABS : AIR_PACK_NONE;
}

int air_conv2d_prepack(const AirConv2d* p, const float* w, int pass, void* out, size_t out_bytes, air_stream_t stream) {
  if (!p || !w || !out || !shape_ok(p)) return AIR_EINVAL;
  const size_t need = air_conv2d_prepack_bytes(p, pass);
  if (need == 0) return AIR_EUNSUPPORTED;
  if (out_bytes < need) return AIR_EWORKSPACE;
  const int M = pass ? p->Cin : p->Cout, Kc = pass ? p->Cout : p->Cin;
  float* up = reinterpret_cast<float*>(out);
  const int kind = wino_kind(p, pass);
  if (kind == 0 && pass == 0 && bf3_fwd_ok(p)) return air_bf3_s2_weights(w, nullptr, out, p->Cout, p->Cin, air_stream(stream));
  if (kind == 0) {  // the direct kernels' slabs, in the order fwd_generic / dgrad_generic consume them
    PackScope collect(PK_COLLECT, up);
    float* none = nullptr;
    return pass ? dgrad_generic(p, nullptr, w, nullptr, nullptr, none, air_stream(stream))
                : fwd_generic(p, nullptr, w, nullptr, nullptr, nullptr, 0, nullptr, none, air_stream(stream));
  }
  return kind == 4 ? air_wino4_weights(w, up, M, Kc, p->H, pass, air_stream(stream), true)
                   : air_wino_weights(w, up, M, Kc, pass, air_stream(stream));
}

size_t air_conv2d_fwd_stats_bytes(const AirConv2d* p) {
  if (!p || !shape_ok(p) || direct_ok(p) || !generic_ok(p)) return 0;
  if (!wino_shape(p) || !air_wino4_ok(p->B, p->Cin, p->H, p->W, p->Cout) || p->Cout % 32 != 0) return 0;
  return air_wino4_stats_bytes(p->B, p->H, p->W, p->Cout);
}

int air_conv2d_fwd_pre(const AirConv2d* p, const float* x, const float* w, const void* w_packed, float* y,
                       const float* in_scale, const float* in_shift, int relu, const float* residual,
                       double* stats, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!p || !x || !w || !y || !shape_ok(p)) return AIR_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return AIR_EINVAL;
  // fused BatchNorm statistics: from the Winograd F(4x4 / 3x4, 3x3) epilogue only (air_conv2d_fwd_stats_bytes says
  // whether this layer takes that path under the current dispatch options)
  if (stats != nullptr && air_conv2d_fwd_stats_bytes(p) == 0) return AIR_EUNSUPPORTED;
  if (stats != nullptr && in_scale != nullptr) return AIR_EUNSUPPORTED;
  hipStream_t st = air_stream(stream);
  if (direct_ok(p)) {
    if (in_scale || relu || residual) return AIR_EUNSUPPORTED;
    DirectArgs a = {x, w, y, nullptr, nullptr, p->B, p->Cin, p->H, p->W, p->Cout, p->KH, p->KW,
                    p->sh, p->sw, p->ph, p->pw, p->Ho, p->Wo};
    hipLaunchKernelGGL(conv_direct_fwd_kernel, dim3(grid_for((size_t)p->B * p->Ho * p->Wo)),
                       dim3(256), 0, st, a);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  if (!generic_ok(p)) return AIR_EUNSUPPORTED;
  // fused prologue = BatchNorm-apply + ReLU together (the pre-activation block), or none
  if ((in_scale != nullptr) != (relu != 0)) return AIR_EUNSUPPORTED;
  if (in_scale != nullptr && p->Cin > MAXC) return AIR_EUNSUPPORTED;
  const size_t wsz = (size_t)p->Cout * p->Cin * p->KH * p->KW;
  if (!ws || ws_bytes < wsz * sizeof(float)) return AIR_EWORKSPACE;
  float* wp = reinterpret_cast<float*>(ws);
  if (in_scale == nullptr && wino_shape(p) && air_wino4_ok(p->B, p->Cin, p->H, p->W, p->Cout)) {
    if (ws_bytes < air_wino4_packed_elems(p->Cout, p->Cin) * sizeof(float)) return AIR_EWORKSPACE;
    if (w_packed)  // transformed earlier (air_conv2d_prepack): no weights kernel in front of the conv
      return air_wino4_conv(x, nullptr, y, residual, p->B, p->Cin, p->H, p->W, p->Cout, 0,
                            const_cast<float*>(reinterpret_cast<const float*>(w_packed)), conv_flops(p), st,
                            reinterpret_cast<float*>(stats));
    return air_wino4_conv(x, w, y, residual, p->B, p->Cin, p->H, p->W, p->Cout, 0, wp,
                          conv_flops(p), st, reinterpret_cast<float*>(stats));
  }
  if (in_scale == nullptr && wino_shape(p) && air_wino_ok(p->B, p->Cin, p->H, p->W, p->Cout)) {
    if (ws_bytes < air_wino_packed_elems(p->Cout, p->Cin) * sizeof(float)) return AIR_EWORKSPACE;
    if (w_packed)
      return air_wino_conv(x, nullptr, y, residual, p->B, p->Cin, p->H, p->W, p->Cout, 0,
                           const_cast<float*>(reinterpret_cast<const float*>(w_packed)), conv_flops(p), st);
    return air_wino_conv(x, w, y, residual, p->B, p->Cin, p->H, p->W, p->Cout, 0, wp,
                         conv_flops(p), st);
  }
  if (in_scale == nullptr && residual == nullptr && bf3_fwd_ok(p)) {
    const size_t need = air_bf3_s2_packed_bytes(p->Cout, p->Cin, false);
    if (w_packed == nullptr) {
      if (ws_bytes < need) return AIR_EWORKSPACE;
      const int rc = air_bf3_s2_weights(w, nullptr, ws, p->Cout, p->Cin, st);
      if (rc != AIR_OK) return rc;
    }
    return air_bf3_s2_fwd(x, w_packed ? w_packed : ws, y, nullptr, p->B, p->Cin, p->H, p->W, p->Cout, p->Ho, p->Wo,
                          conv_flops(p), st);
  }
  // slabs from air_conv2d_prepack (a Winograd-shaped layer's buffer is not ours; neither are the split-bf16 planes that
  // prepack writes for a bf3-eligible layer - a call with a prologue or a residual lands here and packs from w instead)
  if (w_packed != nullptr && wino_kind(p, 0) == 0 && !bf3_fwd_ok(p)) {
    PackScope use(PK_USE, const_cast<float*>(reinterpret_cast<const float*>(w_packed)));
    return fwd_generic(p, x, w, y, in_scale, in_shift, relu, residual, wp, st, ws_bytes / sizeof(float));
  }
  return fwd_generic(p, x, w, y, in_scale, in_shift, relu, residual, wp, st, ws_bytes / sizeof(float));
}

// ---- forward of a stride-2 PreActBlock's two convolutions on the same input (resnet.py:56-66) in one launch
size_t air_conv2d_fwd_s2_pair_prepack_bytes(const AirConv2d* p) {
  if (!p || !shape_ok(p) || !bf3_fwd_ok(p)) return 0;
  return air_bf3_s2_packed_bytes(p->Cout, p->Cin, true);
}

int air_conv2d_fwd_s2_pair_prepack(const AirConv2d* p, const float* w, const float* w_sc, void* out, size_t out_bytes,
                                   air_stream_t stream) {
  if (!p || !w || !w_sc || !out || !shape_ok(p)) return AIR_EINVAL;
  const size_t need = air_conv2d_fwd_s2_pair_prepack_bytes(p);
  if (need == 0) return AIR_EUNSUPPORTED;
  if (out_bytes < need) return AIR_EWORKSPACE;
  return air_bf3_s2_weights(w, w_sc, out, p->Cout, p->Cin, air_stream(stream));
}

int air_conv2d_fwd_s2_pair(const AirConv2d* p, const float* x, const float* w, const float* w_sc, const void* packed,
                           float* y, float* y_sc, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!p || !x || !w || !w_sc || !y || !y_sc || !shape_ok(p)) return AIR_EINVAL;
  const size_t need = air_conv2d_fwd_s2_pair_prepack_bytes(p);
  if (need == 0) return AIR_EUNSUPPORTED;
  hipStream_t st = air_stream(stream);
  if (packed == nullptr) {
    if (g_pk_defer) return AIR_EINVAL;  // (nothing packs in place inside a prepack block)
    if (!ws || ws_bytes < need) return AIR_EWORKSPACE;
    const int rc = air_bf3_s2_weights(w, w_sc, ws, p->Cout, p->Cin, st);
    if (rc != AIR_OK) return rc;
    packed = ws;
  }
  // (3x3 + 1x1: 10 / 9 of the 3x3 layer's multiply-adds)
  return air_bf3_s2_fwd(x, packed, y, y_sc, p->B, p->Cin, p->H, p->W, p->Cout, p->Ho, p->Wo, conv_flops(p) * 10.0 / 9.0, st);
}

int air_conv2d_fwd(const AirConv2d* p, const float* x, const float* w, float* y,
                   const float* in_scale, const float* in_shift, int relu, const float* residual,
                   double* stats, void* ws, size_t ws_bytes, air_stream_t stream) {
  return air_conv2d_fwd_pre(p, x, w, nullptr, y, in_scale, in_shift, relu, residual, stats, ws, ws_bytes, stream);
}

int air_conv2d_dgrad_pre(const AirConv2d* p, const float* dy, const float* w, const void* w_packed, float* dx,
                         const float* accumulate, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!p || !dy || !w || !dx || !shape_ok(p)) return AIR_EINVAL;
  if (!generic_ok(p)) return AIR_EUNSUPPORTED;
  hipStream_t st = air_stream(stream);
  const size_t wsz = packed_dgrad_elems(p);
  if (!ws || ws_bytes < wsz * sizeof(float)) return AIR_EWORKSPACE;
  float* wp = reinterpret_cast<float*>(ws);
  // roles swap: "input" channels = Cout, "output" channels = Cin
  if (wino_shape(p) && air_wino4_ok(p->B, p->Cout, p->H, p->W, p->Cin)) {
    if (ws_bytes < air_wino4_packed_elems(p->Cin, p->Cout) * sizeof(float)) return AIR_EWORKSPACE;
    if (w_packed)
      return air_wino4_conv(dy, nullptr, dx, accumulate, p->B, p->Cout, p->H, p->W, p->Cin, 1,
                            const_cast<float*>(reinterpret_cast<const float*>(w_packed)), conv_flops(p), st);
    return air_wino4_conv(dy, w, dx, accumulate, p->B, p->Cout, p->H, p->W, p->Cin, 1, wp,
                          conv_flops(p), st);
  }
  if (wino_shape(p) && air_wino_ok(p->B, p->Cout, p->H, p->W, p->Cin)) {
    if (ws_bytes < air_wino_packed_elems(p->Cin, p->Cout) * sizeof(float)) return AIR_EWORKSPACE;
    if (w_packed)
      return air_wino_conv(dy, nullptr, dx, accumulate, p->B, p->Cout, p->H, p->W, p->Cin, 1,
                           const_cast<float*>(reinterpret_cast<const float*>(w_packed)), conv_flops(p), st);
    return air_wino_conv(dy, w, dx, accumulate, p->B, p->Cout, p->H, p->W, p->Cin, 1, wp,
                         conv_flops(p), st);
  }
  if (w_packed != nullptr && wino_kind(p, 1) == 0) {
    PackScope use(PK_USE, const_cast<float*>(reinterpret_cast<const float*>(w_packed)));
    return dgrad_generic(p, dy, w, dx, accumulate, wp, st);
  }
  return dgrad_generic(p, dy, w, dx, accumulate, wp, st);
}

int air_conv2d_dgrad(const AirConv2d* p, const float* dy, const float* w, float* dx,
                     const float* accumulate, void* ws, size_t ws_bytes, air_stream_t stream) {
  return air_conv2d_dgrad_pre(p, dy, w, nullptr, dx, accumulate, ws, ws_bytes, stream);
}

size_t air_conv2d_dgrad_bn_sums_bytes(const AirConv2d* p) {
  if (!p || !shape_ok(p) || !generic_ok(p)) return 0;
  // the Winograd F(3x4 / 4x4, 3x3) data-gradient kernel, whole 32-channel slabs of dx
  if (!wino_shape(p) || !air_wino4_ok(p->B, p->Cout, p->H, p->W, p->Cin) || p->Cin % 32 != 0) return 0;
  return air_wino4_stats_bytes(p->B, p->H, p->W, p->Cin);
}

int air_conv2d_dgrad_bn(const AirConv2d* p, const float* dy, const float* w, const void* w_packed, float* dx,
                        const float* accumulate, const float* bn_x, const float* bn_mean, const float* bn_invstd,
                        const float* bn_gamma, const float* bn_beta, void* sums, void* ws, size_t ws_bytes,
                        air_stream_t stream) {
  if (!p || !dy || !w || !dx || !shape_ok(p) || !bn_x || !bn_mean || !bn_invstd || !bn_gamma || !bn_beta || !sums)
    return AIR_EINVAL;
  if (air_conv2d_dgrad_bn_sums_bytes(p) == 0) return AIR_EUNSUPPORTED;
  if (ws_bytes < air_wino4_packed_elems(p->Cin, p->Cout) * sizeof(float) || !ws) return AIR_EWORKSPACE;
  const float* bn[5] = {bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta};
  hipStream_t st = air_stream(stream);
  if (w_packed)
    return air_wino4_conv(dy, nullptr, dx, accumulate, p->B, p->Cout, p->H, p->W, p->Cin, 1,
                          const_cast<float*>(reinterpret_cast<const float*>(w_packed)), conv_flops(p), st,
                          reinterpret_cast<float*>(sums), bn);
  return air_wino4_conv(dy, w, dx, accumulate, p->B, p->Cout, p->H, p->W, p->Cin, 1, reinterpret_cast<float*>(ws),
                        conv_flops(p), st, reinterpret_cast<float*>(sums), bn);
}

static size_t s2d_pack_elems(const AirConv2d* p, int taps) {
  const int bm = 32 * s2d_mt(p);
  return (size_t)((p->Cin + bm - 1) / bm * bm) * p->Cout * taps;
}

// the split-bf16 form of the paired data gradient (conv_bf3.hip; option CONV_S2 bit 8)
static bool bf3_s2d_ok(const AirConv2d* p) {
  return s2d_ok(p) && generic_ok(p) && air_bf3_s2d_ok(p->B, p->Cin, p->H, p->W, p->Cout);
}

size_t air_conv2d_dgrad_s2_pair_prepack_bytes(const AirConv2d* p) {
  if (!p || !shape_ok(p) || !generic_ok(p) || !s2d_ok(p)) return 0;
  if (bf3_s2d_ok(p)) return air_bf3_s2d_packed_bytes(p->Cout, p->Cin);
  return (s2d_pack_elems(p, 9) + s2d_pack_elems(p, 1)) * sizeof(float);
}

int air_conv2d_dgrad_s2_pair_prepack(const AirConv2d* p, const float* w, const float* w_sc, void* out, size_t out_bytes,
                                     air_stream_t stream) {
  if (!p || !w || !w_sc || !out || !shape_ok(p)) return AIR_EINVAL;
  const size_t need = air_conv2d_dgrad_s2_pair_prepack_bytes(p);
  if (need == 0) return AIR_EUNSUPPORTED;
  if (out_bytes < need) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  if (bf3_s2d_ok(p)) return air_bf3_s2d_weights(w, w_sc, out, p->Cout, p->Cin, st);
  const int mt = s2d_mt(p);
  float* w9 = reinterpret_cast<float*>(out);
  float* w1 = w9 + s2d_pack_elems(p, 9);
  TapSel s9, s1;
  s9.n = 9;
  for (int t = 0; t < 9; ++t) s9.idx[t] = t;
  s1.n = 1;
  s1.idx[0] = 0;
  PackScope collect(PK_COLLECT, w9);  // (the two slabs back to back in `out`; deferred like any prepack between begin / flush)
  int rc = pack(w, w9, p->Cout, p->Cin, 9, 1, S2D_CK, mt, s9, st);
  if (rc != AIR_OK) return rc;
  return pack(w_sc, w1, p->Cout, p->Cin, 1, 1, S2D_CK, mt, s1, st);
}

int air_conv2d_prepack_begin(void) {
  g_pk_defer = true;
  g_pk_jobs.n = 0;
  g_pk_jobs.blk0[0] = 0;
  air_wino4_weights_defer(true);
  return AIR_OK;
}

int air_conv2d_prepack_flush(air_stream_t stream) {
  hipStream_t st = air_stream(stream);
  g_pk_defer = false;
  const int rc = pack_run_jobs(st);
  const int rc4 = air_wino4_weights_flush(st);
  air_wino4_weights_defer(false);
  return rc != AIR_OK ? rc : rc4;
}

int air_conv2d_dgrad_s2_pair(const AirConv2d* p, const float* dy, const float* w, const float* dy_sc, const float* w_sc,
                             const void* packed, float* dx, const float* accumulate, void* ws, size_t ws_bytes,
                             air_stream_t stream) {
  if (!p || !dy || !w || !dy_sc || !w_sc || !dx || !shape_ok(p)) return AIR_EINVAL;
  const size_t need = air_conv2d_dgrad_s2_pair_prepack_bytes(p);
  if (need == 0) return AIR_EUNSUPPORTED;
  const float* w9 = reinterpret_cast<const float*>(packed);
  if (bf3_s2d_ok(p)) {
    if (packed == nullptr) {
      if (g_pk_defer) return AIR_EINVAL;  // (one rule for every pair launch: nothing packs in place inside a prepack block)
      if (!ws || ws_bytes < need) return AIR_EWORKSPACE;
      const int rc = air_bf3_s2d_weights(w, w_sc, ws, p->Cout, p->Cin, air_stream(stream));
      if (rc != AIR_OK) return rc;
      packed = ws;
    }
    return air_bf3_s2d_dgrad(dy, dy_sc, packed, dx, accumulate, p->B, p->Cin, p->H, p->W, p->Cout, p->Ho, p->Wo,
                             conv_flops(p) * 10.0 / 9.0, air_stream(stream));
  }
  if (w9 == nullptr) {
    if (g_pk_defer) return AIR_EINVAL;  // between air_conv2d_prepack_begin and _flush nothing is packed yet
    if (!ws || ws_bytes < need) return AIR_EWORKSPACE;
    const int rc = air_conv2d_dgrad_s2_pair_prepack(p, w, w_sc, ws, ws_bytes, stream);
    if (rc != AIR_OK) return rc;
    w9 = reinterpret_cast<const float*>(ws);
  }
  return run_s2d(p, dy, dy_sc, w9, w9 + s2d_pack_elems(p, 9), dx, accumulate, s2d_mt(p), air_stream(stream));
}

int air_conv2d_wgrad(const AirConv2d* p, const float* x, const float* dy, float* dw,
                     const float* in_scale, const float* in_shift, int relu, void* ws,
                     size_t ws_bytes, air_stream_t stream) {
  if (!p || !x || !dy || !dw || !shape_ok(p)) return AIR_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return AIR_EINVAL;
  hipStream_t st = air_stream(stream);
  const size_t wsz = (size_t)p->Cout * p->Cin * p->KH * p->KW;
  if (direct_ok(p)) {
    if (in_scale || relu) return AIR_EUNSUPPORTED;
    const int rows = p->B * p->Ho;
    if (!ws || ws_bytes < (size_t)rows * wsz * sizeof(float)) return AIR_EWORKSPACE;
    DirectArgs a = {x, nullptr, nullptr, dy, reinterpret_cast<float*>(ws), p->B, p->Cin, p->H,
                    p->W, p->Cout, p->KH, p->KW, p->sh, p->sw, p->ph, p->pw, p->Ho, p->Wo};
    const int npair = p->Cout * p->Cin * p->KH;
    int nseg = npair > 0 && npair <= 512 ? 512 / npair : 1;
    if (nseg > 4) nseg = 4;
    // the staged rows and, once they are dead, the per-segment sums red[nseg][npair][KW] share the allocation:
    // short rows (W <= 48 for the ResNet's conv1) make the sums the larger of the two
    const size_t stage_f = (size_t)p->Cout * ((p->Wo + 3) & ~3) +
                           (size_t)p->Cin * p->KH * (((p->W + 2 * p->pw + 3) & ~3) + 4);
    const size_t red_f = (size_t)nseg * npair * p->KW;
    const size_t lds = (stage_f > red_f ? stage_f : red_f) * sizeof(float);
    const int use_rows = air_opt(AIR_OPT_DIRECT_WGRAD_ROWS);
    if (use_rows && p->KW <= DIRECT_KW_MAX && npair <= 512 && lds <= 150 * 1024) {
      const int nthr = (npair * nseg + 63) / 64 * 64;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_direct_wgrad_rows_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return AIR_ELAUNCH;
      hipLaunchKernelGGL(conv_direct_wgrad_rows_kernel, dim3(rows), dim3(nthr), lds, st, a, nseg);
    } else {
      hipLaunchKernelGGL(conv_direct_wgrad_kernel, dim3(rows), dim3(256), 0, st, a);
    }
    AIR_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_grid(wsz, rows)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), dw, wsz, rows, 1);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  if (!generic_ok(p)) return AIR_EUNSUPPORTED;
  if ((in_scale != nullptr) != (relu != 0)) return AIR_EUNSUPPORTED;  // BN-apply + ReLU together, or none
  if (in_scale == nullptr && wino_shape(p) && air_wino_wgrad_ok(p->B, p->Cin, p->H, p->W, p->Cout)) {
    const int nsplit = air_wino_wgrad_nsplit(p->B, p->Cin, p->H, p->W, p->Cout);
    if (!ws || ws_bytes < (size_t)nsplit * wsz * sizeof(float)) return AIR_EWORKSPACE;
    int rc = air_wino_wgrad_partials(x, dy, reinterpret_cast<float*>(ws), p->B, p->Cin, p->H, p->W,
                                     p->Cout, conv_flops(p), st);
    if (rc != AIR_OK) return rc;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_grid(wsz, nsplit)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), dw, wsz, nsplit, 9);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  if (in_scale == nullptr && bf3_wgrad_ok(p)) {  // split-bf16 stride-2 weight gradient (conv_bf3.hip)
    const int nseg = air_bf3_s2w_nseg(p->B, p->Cin, p->Ho, p->Wo, p->Cout);
    if (!ws || ws_bytes < (size_t)nseg * wsz * sizeof(float)) return AIR_EWORKSPACE;
    int rc = air_bf3_s2w_partials(x, dy, reinterpret_cast<float*>(ws), p->B, p->Cin, p->H, p->W, p->Cout, p->Ho, p->Wo,
                                  conv_flops(p), st);
    if (rc != AIR_OK) return rc;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_grid(wsz, nseg)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), dw, wsz, nseg, 9);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  if (skinny3_wgrad_ok(p)) {
    const int nparts = skinny3_wgrad_parts(p);
    if (!ws || ws_bytes < (size_t)nparts * wsz * sizeof(float)) return AIR_EWORKSPACE;
    Skinny3Wg a = {x, dy, reinterpret_cast<float*>(ws), in_scale, in_shift, (size_t)p->Cin * p->H * p->W,
                   (size_t)p->Cout * p->H * p->W, p->B, p->H, p->W};
    {
      AirProfScope ps(AIR_K_CONV_WG_331, conv_flops(p), st);
      if (in_scale)
        hipLaunchKernelGGL((conv_wgrad_3x3_skinny_kernel<4, true>), dim3(nparts), dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((conv_wgrad_3x3_skinny_kernel<4, false>), dim3(nparts), dim3(256), 0, st, a);
      AIR_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_grid(wsz, nparts)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), dw, wsz, nparts, 9);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  if (in_scale == nullptr && wino_pad_wgrad(p)) {
    const size_t hw = (size_t)p->H * p->W;
    const size_t xpad_n = ((size_t)p->B * 64 * hw + 63) / 64 * 64;
    const int nsplit = air_wino_wgrad_nsplit(p->B, 64, p->H, p->W, p->Cout);
    const size_t w64 = (size_t)p->Cout * 64 * 9;
    if (!ws || ws_bytes < (xpad_n + (size_t)(nsplit + 1) * w64) * sizeof(float)) return AIR_EWORKSPACE;
    float* xpad = reinterpret_cast<float*>(ws);
    float* partial = xpad + xpad_n;
    float* dw64 = partial + (size_t)nsplit * w64;
    if (hipMemsetAsync(xpad, 0, xpad_n * sizeof(float), st) != hipSuccess) return AIR_ELAUNCH;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for((size_t)p->Cin * hw), p->B), dim3(256), 0, st, xpad,
                       (size_t)64 * hw, x, (size_t)p->Cin * hw, (size_t)p->Cin * hw);
    AIR_CHECK_LAUNCH();
    int rc = air_wino_wgrad_partials(xpad, dy, partial, p->B, 64, p->H, p->W, p->Cout, conv_flops(p), st);
    if (rc != AIR_OK) return rc;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_grid(w64, nsplit)), dim3(256), 0, st, partial, dw64, w64, nsplit, 9);
    AIR_CHECK_LAUNCH();
    // dw[co][ci < Cin][tap] = dw64[co][ci][tap]: the first Cin * 9 floats of every 64 * 9 row
    hipLaunchKernelGGL(copy_rows_kernel, dim3(1, p->Cout), dim3(256), 0, st, dw, (size_t)p->Cin * 9, dw64,
                       (size_t)64 * 9, (size_t)p->Cin * 9);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  if (const int nch = skinny_wgrad_chunks(p)) {
    const int nparts = p->B * nch;
    if (!ws || ws_bytes < (size_t)nparts * wsz * sizeof(float)) return AIR_EWORKSPACE;
    SkinnyWg a = {x, dy, reinterpret_cast<float*>(ws), in_scale, in_shift, (size_t)p->Cin * p->H * p->W,
                  (size_t)p->Cout * p->H * p->W, p->H * p->W, skinny_wgrad_chunk_px(p, nch), nch};
    {
      AirProfScope ps(AIR_K_CONV_WG_111, conv_flops(p), st);
      if (in_scale)
        hipLaunchKernelGGL((conv_wgrad_1x1_skinny_kernel<4, 1, true>), dim3(nparts), dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((conv_wgrad_1x1_skinny_kernel<4, 1, false>), dim3(nparts), dim3(256), 0, st, a);
      AIR_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(reduce_grid(wsz, nparts)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), dw, wsz, nparts, 1);
    AIR_CHECK_LAUNCH();
    return AIR_OK;
  }
  WgradGeom g = {p->B, p->Cin, p->H, p->W, p->Cout, p->KH, p->KW, p->sh, p->ph, p->pw, p->Ho, p->Wo,
                 1, (size_t)p->Cin * p->H * p->W, (size_t)p->Cout * p->Ho * p->Wo};
  return run_wgrad(g, x, dy, dw, in_scale, in_shift, relu, ws, ws_bytes, conv_flops(p), st);
}

// ------------------------------------------------------------------ conv1d
// nn.Conv1d (stride 1) of ecapa_tdnn.py (:39,:46,:55,:111,:118,:140,:143) seen as an H = 1
// image: the same implicit-GEMM kernels, with bias / ReLU / per-utterance bias epilogues
// (conv -> ReLU -> BN ordering, ecapa_tdnn.py:67-69) and batch strides so the Res2 channel
// groups and the (x1,x2,x3) concat are views, never copies.
static bool c1d_ok(const AirConv1d* p) {
  if (!p || p->B <= 0 || p->Cin <= 0 || p->T <= 0 || p->Cout <= 0) return false;
  if (p->K == 1) return p->pad == 0;
  if (p->K == 3) return p->dil >= 2 && p->dil <= 4 && p->pad == p->dil;
  if (p->K == 5) return p->dil == 1 && p->pad == 2;
  return false;
}
static size_t c1d_xb(const AirConv1d* p) { return p->x_bstride ? p->x_bstride : (size_t)p->Cin * p->T; }
static size_t c1d_yb(const AirConv1d* p) { return p->y_bstride ? p->y_bstride : (size_t)p->Cout * p->T; }

size_t air_conv1d_ws_bytes(const AirConv1d* p) {
  if (!c1d_ok(p)) return 0;
  const size_t cin_p = (size_t)(p->Cin + 63) / 64 * 64, cout_p = (size_t)(p->Cout + 63) / 64 * 64;
  const size_t packed = cin_p * cout_p * p->K;
  const size_t wgrad = (size_t)256 * p->Cout * p->Cin * p->K;
  return (packed > wgrad ? packed : wgrad) * sizeof(float) + 256;
}

int air_conv1d_fwd(const AirConv1d* p, const float* x, const float* w, const float* bias,
                   const float* bias_bc, int relu, float* y, void* ws, size_t ws_bytes,
                   air_stream_t stream) {
  if (!c1d_ok(p) || !x || !w || !y) return AIR_EINVAL;
  if (p->Cout % BM != 0) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_ws_bytes(p)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  float* wp = reinterpret_cast<float*>(ws);
  const int ck = pick_ck(p->K, p->Cin);
  TapSel sel;
  sel.n = p->K;
  for (int t = 0; t < p->K; ++t) sel.idx[t] = t;
  const int mt = mt_for(p->B, 1, p->T, p->Cout);
  int rc = pack(w, wp, p->Cout, p->Cin, p->K, 0, ck, mt, sel, st);
  if (rc != AIR_OK) return rc;
  FwdGeom g = plain_geom(p->B, p->Cin, 1, p->T, p->Cout, 1, p->K, 1, 0, p->pad, 1, p->T);
  g.x_bstride = c1d_xb(p); g.y_bstride = c1d_yb(p);
  g.bias = bias; g.bias_bc = bias_bc; g.out_relu = relu; g.dil = p->K == 3 ? p->dil : 1;
  return run_fwd(x, wp, y, nullptr, nullptr, 0, nullptr, g, ck, mt,
                 2.0 * p->B * p->T * (double)p->Cout * p->Cin * p->K, st);
}

int air_conv1d_dgrad(const AirConv1d* p, const float* dy, const float* w, float* dx,
                     const float* accumulate, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!c1d_ok(p) || !dy || !w || !dx) return AIR_EINVAL;
  if (p->Cout % 8 != 0) return AIR_EUNSUPPORTED;
  if (!ws || ws_bytes < air_conv1d_ws_bytes(p)) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  float* wp = reinterpret_cast<float*>(ws);
  const int ck = pick_ck(p->K, p->Cout);
  TapSel sel;
  sel.n = p->K;
  for (int t = 0; t < p->K; ++t) sel.idx[t] = p->K - 1 - t;
  const int mt = mt_for(p->B, 1, p->T, p->Cin);
  int rc = pack(w, wp, p->Cout, p->Cin, p->K, 1, ck, mt, sel, st);
  if (rc != AIR_OK) return rc;
  const int dil = p->K == 3 ? p->dil : 1;
  FwdGeom g = plain_geom(p->B, p->Cout, 1, p->T, p->Cin, 1, p->K, 1, 0, dil * (p->K - 1) - p->pad,
                         1, p->T);
  g.x_bstride = c1d_yb(p); g.y_bstride = c1d_xb(p); g.dil = dil;
  return run_fwd(dy, wp, dx, nullptr, nullptr, 0, accumulate, g, ck, mt,
                 2.0 * p->B * p->T * (double)p->Cout * p->Cin * p->K, st);
}

int air_conv1d_wgrad(const AirConv1d* p, const float* x, const float* dy, float* dw, void* ws,
                     size_t ws_bytes, air_stream_t stream) {
  if (!c1d_ok(p) || !x || !dy || !dw) return AIR_EINVAL;
  if (p->Cout % BM != 0) return AIR_EUNSUPPORTED;
  WgradGeom g = {p->B, p->Cin, 1, p->T, p->Cout, 1, p->K, 1, 0, p->pad, 1, p->T,
                 p->K == 3 ? p->dil : 1, c1d_xb(p), c1d_yb(p)};
  return run_wgrad(g, x, dy, dw, nullptr, nullptr, 0, ws, ws_bytes,
                   2.0 * p->B * p->T * (double)p->Cout * p->Cin * p->K, air_stream(stream));
}

}  // extern "C"
