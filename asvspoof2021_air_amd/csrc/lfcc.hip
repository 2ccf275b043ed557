// Fused LFCC front-end for gfx950.
//
// Replaces LFCC.forward (feature_extraction.py:93-138): pre-emphasis (:105-106)
// -> framing + periodic Hamming window + 512-point rFFT (torch.stft, :109-111)
// -> power (:113) -> triangular linear filterbank + log10 (:116-117)
// -> DCT-II (:120) -> delta / delta-delta (:41-58, :130-133), in ONE launch:
// PCM is read once from HBM (coalesced), everything in between lives in LDS /
// registers, and the (B,T,60) rows are written once, fully coalesced.
//
// Mapping (wave64): a workgroup of 8 waves owns 60 consecutive output frames
// of one utterance (+2 halo frames each side for the delta-deltas) and stages
// their 10,400 pre-emphasised samples in LDS.  Each wave processes 4 frames at
// a time, 16 lanes per frame.  The 512-point real FFT is a 256-point complex
// FFT (even/odd packing) done as 16 x 16: every lane runs a 16-point FFT in
// registers on a stride-16 decimation, the 16x16 transpose goes through a
// padded (17-column) wave-private LDS tile (conflict-free both ways), a second
// 16-point FFT finishes it, and the real-FFT untangle pairs bin k with 256-k
// held by lane (16-q) of the same 16-lane group.  torch.stft centres the
// 320-tap window inside the 512 frame (offset 96); placing it at offset 0
// only changes the phase, which |X|^2 discards.
#include <math.h>
#include <string.h>

#include "air_common.h"
#include "air_prof.h"
#include "air_fft16.h"

namespace {

constexpr int FN = 512, FL = 320, FS = 160;
constexpr int NBIN = FN / 2 + 1;  // 257
constexpr int MAXF = 32;          // filters supported (2 per lane of a 16-lane group)
constexpr int MAXW = 32;          // bins per filter supported
// (A/B knobs, round 6: -DLFCC_NW / -DLFCC_GPW / -DLFCC_MINB - waves per workgroup, 4-frame groups per wave, resident
// workgroups per CU the compiler plans for; profiles/r06_lfcc.md)
#ifndef LFCC_NW
#define LFCC_NW 8
#endif
#ifndef LFCC_GPW
#define LFCC_GPW 1
#endif
#ifndef LFCC_MINB
#define LFCC_MINB 2
#endif
constexpr int NW = LFCC_NW;       // waves per workgroup
constexpr int NTHREADS = NW * 64;
constexpr int GPW = LFCC_GPW;           // 4-frame groups per wave
constexpr int FCOMP = NW * GPW * 4;     // 32 frames computed per workgroup
constexpr int HALO = 2;                 // delta-delta reaches 2 frames each side
constexpr int FOUT = FCOMP - 2 * HALO;  // 28 frames written per workgroup
constexpr int NSAMP = FS * (FCOMP + 1); // 5,280 staged samples
constexpr int XROW = 17;                // padded transpose row (complex elements)
// wave-private exchange tile: the 4 x 16 x 17 transposition (real and imaginary parts in two passes), then the 4
// power rows + 4 x 32 filterbank outputs; 69.6 KB for the workgroup - two workgroups per CU (round 1: one)
constexpr int XCH_FLOATS = 4 * 264 + 4 * 32;
constexpr int PROW = 264;               // power-spectrum row stride (floats)
constexpr int FBUF_OFF = 4 * PROW;      // filterbank outputs live after the 4 power rows

struct LfccPlan {
  int nfilt, maxw, fl, fs, fn, nbin, pad0, pad1;
  float window[FL];
  float tw256[512];           // e^{-2 pi i n/256}: (re, im), n = 0..255
  float tw512[32];            // e^{-2 pi i q/512}: (re, im), q = 0..15
  int lo[MAXF];               // first bin of filter j
  int cnt[MAXF];              // bins spanned by filter j
  float fbwT[MAXW * MAXF];    // [i][j] = fb[lo[j]+i][j]
  float dctT[MAXF * MAXF];    // [j][i] = dct[i][j]
};

__device__ __constant__ const float W32C[16] = {
    1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f,
    0.382683432f, 0.195090322f, 0.000000000f, -0.195090322f, -0.382683432f, -0.555570233f,
    -0.707106781f, -0.831469612f, -0.923879533f, -0.980785280f};
__device__ __constant__ const float W32S[16] = {
    0.000000000f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f,
    0.923879533f, 0.980785280f, 1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f,
    0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f};

struct LfccArgs {
  const float* pcm;
  const short* pcm16;  // non-null: 16-bit PCM as stored in the corpus' wav/flac files; x = s / 32768 (exact)
  float* out;
  const LfccPlan* plan;
  const int* start;  // padded mode: per-utterance crop start (may be null)
  const float* silence;  // padded mode, AIR_PAD_SILENCE: the D-float frame to prepend (dataset.py:13-16)
  int L, T, tiles, flags, feat_len, pad_mode;
};

constexpr int FLAG_EMPH = AIR_LFCC_EMPHASIS;
constexpr int FLAG_DELTA = AIR_LFCC_DELTA;
constexpr int FLAG_PADDED = 1 << 8;  // internal: (B, D, feat_len) output
#ifndef LFCC_EXACT_NORM
#define LFCC_EXACT_NORM 0
#endif

// NF_C / MW_C > 0: the filter count and the widest filter are compile-time (the reference's LFCC(320, 160, 512, 16000,
// 20): 20 filters, 26 bins) - the filterbank and DCT loops unroll onto immediate LDS offsets with no index arithmetic
// (round 5: the kernel is VALU-issue bound - SQ_ACTIVE_INST_VALU 59 % of its cycles, 1314 VALU instructions per wave,
// profiles/r05_lfcc.md - and ~40 % of them were address arithmetic, clamps and the correctly rounded sqrt).  0: any plan.
template <int NF_C, int MW_C>
__device__ __forceinline__ void lfcc_body(const LfccArgs& a, float* __restrict__ s_pcm, float* __restrict__ s_xch,
                                          float* __restrict__ s_c, float* __restrict__ s_fbwT,
                                          float* __restrict__ s_dctT) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int b = blockIdx.x / a.tiles;
  const int tile = blockIdx.x - b * a.tiles;
  const int t0 = tile * FOUT;
  const LfccPlan* __restrict__ plan = a.plan;
  const int nfilt = NF_C ? NF_C : plan->nfilt;
  const int maxw = MW_C ? MW_C : plan->maxw;
  const int L = a.L, T = a.T;
  const float* __restrict__ row = a.pcm + (size_t)b * L;
  const short* __restrict__ row16 = a.pcm16 ? a.pcm16 + (size_t)b * L : nullptr;
  // sample m of this utterance as the float the reference's loader hands to LFCC.forward
  auto sample = [&](long m) -> float { return row16 ? (float)row16[m] * (1.0f / 32768.0f) : row[m]; };

  // ---- stage tables and pre-emphasised PCM ---------------------------------
  for (int e = tid; e < MAXW * MAXF; e += NTHREADS) s_fbwT[e] = plan->fbwT[e];
  for (int e = tid; e < MAXF * MAXF; e += NTHREADS) s_dctT[e] = plan->dctT[e];
  const long s0 = (long)FS * (t0 - HALO) - FL / 2;  // first staged sample (may be < 0)
  const bool emph = (a.flags & FLAG_EMPH) != 0;
  const bool vec_ok = row16 == nullptr && ((((size_t)row) & 15) == 0);
  for (int e4 = tid; e4 < NSAMP / 4; e4 += NTHREADS) {
    const long n = s0 + 4 * (long)e4;
    float v[4];
    if (vec_ok && n >= 0 && n + 3 < L) {
      const float4 q = *reinterpret_cast<const float4*>(row + n);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      if (emph) {
        const float prev = n > 0 ? row[n - 1] : 0.0f;
        // non-recursive FIR on the ORIGINAL samples, two roundings like the reference
        v[3] = __fsub_rn(v[3], __fmul_rn(0.97f, v[2]));
        v[2] = __fsub_rn(v[2], __fmul_rn(0.97f, v[1]));
        v[1] = __fsub_rn(v[1], __fmul_rn(0.97f, v[0]));
        if (n > 0) v[0] = __fsub_rn(v[0], __fmul_rn(0.97f, prev));
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long m = n + k;
        float x = 0.0f;
        if (m >= 0 && m < L) {
          x = sample(m);
          if (emph && m > 0) x = __fsub_rn(x, __fmul_rn(0.97f, sample(m - 1)));
        }
        v[k] = x;
      }
    }
    *reinterpret_cast<float4*>(&s_pcm[4 * e4]) = make_float4(v[0], v[1], v[2], v[3]);
  }

  // ---- per-lane constants --------------------------------------------------
  const int f = lane >> 4;  // frame within the 4-frame group
  const int g = lane & 15;  // lane within the frame
  float wre[10], wim[10];
#pragma unroll
  for (int m = 0; m < 10; ++m) {
    wre[m] = plan->window[2 * (g + 16 * m)];
    wim[m] = plan->window[2 * (g + 16 * m) + 1];
  }
  cf tw[16];  // w256^(g*q)
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int n = (g * q) & 255;
    tw[q] = cf{plan->tw256[2 * n], plan->tw256[2 * n + 1]};
  }
  const float cq = plan->tw512[2 * g];        //  cos(2 pi g/512)
  const float sq = -plan->tw512[2 * g + 1];   //  sin(2 pi g/512)
  const int j0 = g, j1 = g + 16;
  const int lo0 = plan->lo[j0];
  const int lo1 = plan->lo[j1 < MAXF ? j1 : 0];
  const int partner = (lane & 48) | ((16 - g) & 15);
  float* xch = s_xch + wave * XCH_FLOATS;

  __syncthreads();

  // ---- frames -> cepstra ---------------------------------------------------
  for (int gg = 0; gg < GPW; ++gg) {
    const int gi = wave * GPW + gg;       // group index in the workgroup
    const int i0 = gi * 4;                // first local frame of the group
    const int tc_first = t0 - HALO + i0;  // its global frame index
    if (tc_first + 3 < 0 || tc_first >= T) continue;  // wave-uniform: nothing to compute
    const int i = i0 + f;

    // windowed even/odd packing: z[n] = w[2n] s[2n] + i w[2n+1] s[2n+1], n = g + 16 m
    cf x[16];
#pragma unroll
    for (int m = 0; m < 10; ++m) {
      const float2 s = *reinterpret_cast<const float2*>(&s_pcm[FS * i + 2 * (g + 16 * m)]);
      x[m] = cf{s.x * wre[m], s.y * wim[m]};
    }
#pragma unroll
    for (int m = 10; m < 16; ++m) x[m] = cf{0.0f, 0.0f};
    fft16<true>(x);  // over m: Y[g][q]
#pragma unroll
    for (int q = 1; q < 16; ++q) x[q] = cmul(x[q], tw[q]);
    // transpose through LDS: row q, column g - real parts, then imaginary parts through the same 4.3 KB
    float xre[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) xch[(f * 16 + q) * XROW + g] = x[q].x;
    air_wave_lds_fence();
#pragma unroll
    for (int l = 0; l < 16; ++l) xre[l] = xch[(f * 16 + g) * XROW + l];
    air_wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 16; ++q) xch[(f * 16 + q) * XROW + g] = x[q].y;
    air_wave_lds_fence();
#pragma unroll
    for (int l = 0; l < 16; ++l) x[l] = cf{xre[l], xch[(f * 16 + g) * XROW + l]};
    fft16<false>(x);  // over l: Z[g + 16 p] = x[p]
    air_wave_lds_fence();  // exchange tile is reused for the power rows below

    // real-FFT untangle + power.  Partner lane holds Z[(16-g)%16 + 16 p'].
    float pnyq = 0.0f;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      // bin k = g + 16 p pairs with 256 - k: lane (16-g)&15, index 15-p  (g != 0)
      //                                      own lane, index (16-p)&15   (g == 0)
      const int pp = 15 - p;
      float bre = __shfl(x[pp].x, partner, 64);
      float bim = __shfl(x[pp].y, partner, 64);
      const int p0 = (16 - p) & 15;
      if (g == 0) {
        bre = x[p0].x;
        bim = x[p0].y;
      }
      // E2 = A + conj(B), O2 = A - conj(B);  2X = E2 - i w512^k O2 - on register pairs (same roundings as the scalar
      // form: xr = (ere + c oim) - s ore, xi = (eim - c ore) - s oim)
      const cf A = x[p], Bv = cf{bre, bim};
      const cf E = __builtin_elementwise_fma(Bv, cf{1.0f, -1.0f}, A);  // (A.re + bre, A.im - bim)
      const cf O = __builtin_elementwise_fma(Bv, cf{-1.0f, 1.0f}, A);  // (A.re - bre, A.im + bim)
      const cf cs = cmul(cf{cq, sq}, cf{W32C[p], W32S[p]});  // (cos, sin)(2 pi k / 512)
      const cf P1 = LFCC_LO(cs) * LFCC_SWAP(O);              // (c oim, c ore)
      const cf X1 = __builtin_elementwise_fma(P1, cf{1.0f, -1.0f}, E);
      const cf X = X1 - LFCC_HI(cs) * O;                     // - (s ore, s oim)
      const cf X2 = X * X;
      const float xr2 = X2.x, xi2 = X2.y;
      // reference: norm(.,2,-1).pow(2) (feature_extraction.py:113), i.e. fl(fl(sqrt(s))^2) - within 1.5 ulp of s itself.
      // Round 5: the power is taken directly (X holds 2 x the bin: the factor 0.25 is exact); LFCC_EXACT_NORM = 1
      // keeps the square root and the square (+ ~160 VALU instructions per wave for the correctly rounded sqrtf).
#if LFCC_EXACT_NORM
      const float mag = 0.5f * sqrtf(xr2 + xi2);
      xch[f * PROW + g + 16 * p] = mag * mag;
#else
      xch[f * PROW + g + 16 * p] = 0.25f * (xr2 + xi2);
#endif
      if (p == 0) {
        const float ny = x[0].x - x[0].y;  // X[256] = Re Z0 - Im Z0 (real)
        pnyq = ny * ny;
      }
    }
    if (g == 0) xch[f * PROW + 256] = pnyq;
    if (MW_C && g >= 1 && g < PROW - 256) xch[f * PROW + 256 + g] = 0.0f;  // zeros behind the Nyquist bin (below)
    air_wave_lds_fence();

    // sparse triangular filterbank + log10 (feature_extraction.py:116-117)
    float acc0 = 0.0f, acc1 = 0.0f;
    if (MW_C) {
      // filter j reads bins lo[j] .. lo[j] + MW_C - 1 <= PROW - 1 (checked where the plan is built): past its own
      // support the weights are zero, past the Nyquist bin the row is zero - no clamp, immediate offsets
      const float* __restrict__ pr0 = xch + f * PROW + lo0;
      const float* __restrict__ pr1 = xch + f * PROW + lo1;
      const float* __restrict__ wr = s_fbwT + j0;
#pragma unroll
      for (int w = 0; w < MW_C; ++w) {
        acc0 = fmaf(pr0[w], wr[w * MAXF], acc0);
        acc1 = fmaf(pr1[w], wr[w * MAXF + 16], acc1);
      }
    } else {
      for (int w = 0; w < maxw; ++w) {
        const int k0 = min(lo0 + w, NBIN - 1), k1 = min(lo1 + w, NBIN - 1);
        acc0 = fmaf(xch[f * PROW + k0], s_fbwT[w * MAXF + j0], acc0);
        acc1 = fmaf(xch[f * PROW + k1], s_fbwT[w * MAXF + j1], acc1);
      }
    }
    xch[FBUF_OFF + f * MAXF + j0] = log10f(acc0 + 1.1920928955078125e-07f);
    xch[FBUF_OFF + f * MAXF + j1] = log10f(acc1 + 1.1920928955078125e-07f);
    air_wave_lds_fence();

    // DCT-II as a 20x20 product (feature_extraction.py:120)
    float c0 = 0.0f, c1 = 0.0f;
    if (NF_C) {
      const float* __restrict__ fr = xch + FBUF_OFF + f * MAXF;
      const float* __restrict__ dr = s_dctT + j0;
#pragma unroll
      for (int j = 0; j < NF_C; ++j) {
        const float fj = fr[j];
        c0 = fmaf(fj, dr[j * MAXF], c0);
        c1 = fmaf(fj, dr[j * MAXF + 16], c1);
      }
    } else {
      for (int j = 0; j < nfilt; ++j) {
        const float fj = xch[FBUF_OFF + f * MAXF + j];
        c0 = fmaf(fj, s_dctT[j * MAXF + j0], c0);
        c1 = fmaf(fj, s_dctT[j * MAXF + j1], c1);
      }
    }
    s_c[i * MAXF + j0] = c0;
    s_c[i * MAXF + j1] = c1;
    air_wave_lds_fence();  // next group overwrites the exchange tile
  }
  __syncthreads();

  // ---- deltas into an LDS tile, then coalesced stores -----------------------
  // Round 5: a thread takes one coefficient of TWO neighbouring frames - six clamped cepstra c(t-2) .. c(t+3) give the
  // static value, the delta and the delta-delta of both (same subtractions in the same order as value() of rounds 1-4) -
  // and writes them into the exchange tile (dead by now); the copy out is then index arithmetic only.
  const bool with_delta = (a.flags & FLAG_DELTA) != 0;
  const int D = with_delta ? 3 * nfilt : nfilt;
  const int base = t0 - HALO;  // global frame of local row 0
  const bool padded = (a.flags & FLAG_PADDED) != 0;
  constexpr int OSTR = FOUT + 1;  // padded layout: s_out[c][fo], odd stride (conflict-free along fo)
  float* __restrict__ s_out = s_xch;
  {
    const int npair = FOUT / 2;
    for (int e = tid; e < nfilt * npair; e += NTHREADS) {
      const int c = e / npair, pr = e - c * npair;
      const int fo = 2 * pr, t = t0 + fo;
      float v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = s_c[(min(max(t - 2 + k, 0), T - 1) - base) * MAXF + c];
      // d[k] = delta at frame t - 1 + k:  c(cl(u + 1)) - c(cl(u - 1))
      const float d0 = v[2] - v[0], d1 = v[3] - v[1], d2 = v[4] - v[2], d3 = v[5] - v[3];
      // delta-delta(u) = delta(min(u + 1, T - 1)) - delta(max(u - 1, 0))   (feature_extraction.py:41-58 applied twice)
      const float dd0 = (t + 1 <= T - 1 ? d2 : d1) - (t - 1 >= 0 ? d0 : d1);
      const float dd1 = (t + 2 <= T - 1 ? d3 : d2) - d1;  // (t >= 0)
      if (!padded) {
        s_out[fo * D + c] = v[2];
        s_out[(fo + 1) * D + c] = v[3];
        if (with_delta) {
          s_out[fo * D + nfilt + c] = d1;
          s_out[(fo + 1) * D + nfilt + c] = d2;
          s_out[fo * D + 2 * nfilt + c] = dd0;
          s_out[(fo + 1) * D + 2 * nfilt + c] = dd1;
        }
      } else {
        s_out[c * OSTR + fo] = v[2];
        s_out[c * OSTR + fo + 1] = v[3];
        if (with_delta) {
          s_out[(nfilt + c) * OSTR + fo] = d1;
          s_out[(nfilt + c) * OSTR + fo + 1] = d2;
          s_out[(2 * nfilt + c) * OSTR + fo] = dd0;
          s_out[(2 * nfilt + c) * OSTR + fo + 1] = dd1;
        }
      }
    }
  }
  __syncthreads();
  if (!padded) {
    float* __restrict__ orow = a.out + ((size_t)b * T + t0) * D;
    const int nvalid = min(FOUT, T - t0);
    const int n = nvalid * D;
    if ((D & 3) == 0 && ((reinterpret_cast<size_t>(orow) & 15) == 0)) {
      for (int e = tid; e < n / 4; e += NTHREADS)
        reinterpret_cast<float4*>(orow)[e] = reinterpret_cast<const float4*>(s_out)[e];
    } else {
      for (int e = tid; e < n; e += NTHREADS) orow[e] = s_out[e];
    }
  } else {
    // (B, D, feat_len): frame t lands on t' = t - start (+ k T when repeating)
    const int flen = a.feat_len;
    // crop start clamped to the valid range: a bad offset must not leave columns unwritten
    const int start = (a.start != nullptr && T > flen) ? min(max(a.start[b], 0), T - flen) : 0;
    float* __restrict__ obase = a.out + (size_t)b * D * flen;
    const int npad = flen - T;  // > 0: pad (dataset.py:72-79)
    const int shift = (npad > 0 && a.pad_mode == AIR_PAD_SILENCE) ? npad : 0;  // silence is PREPENDED (:528)
    for (int e = tid; e < FOUT * D; e += NTHREADS) {
      const int c = e / FOUT, fo = e - c * FOUT;
      const int t = t0 + fo;
      if (t >= T) continue;
      const float v = s_out[c * OSTR + fo];
      if (T >= flen) {
        const int tp = t - start;
        if (tp >= 0 && tp < flen) obase[(size_t)c * flen + tp] = v;
      } else if (a.pad_mode == AIR_PAD_REPEAT) {
        for (int tp = t; tp < flen; tp += T) obase[(size_t)c * flen + tp] = v;
      } else {
        obase[(size_t)c * flen + t + shift] = v;
      }
    }
    if (npad > 0 && a.pad_mode != AIR_PAD_REPEAT) {
      // the npad constant frames (zeros appended, :513-517, or the silence frame prepended, :524-528),
      // shared out over the utterance's tiles
      const int lo = a.pad_mode == AIR_PAD_SILENCE ? 0 : T;
      for (int e = tile * NTHREADS + tid; e < npad * D; e += a.tiles * NTHREADS) {
        const int c = e / npad, k = e - c * npad;
        obase[(size_t)c * flen + lo + k] = a.pad_mode == AIR_PAD_SILENCE ? a.silence[c] : 0.0f;
      }
    }
  }
}

__global__ __launch_bounds__(NTHREADS, LFCC_MINB) void lfcc_kernel(LfccArgs a) {
  __shared__ __attribute__((aligned(16))) float s_pcm[NSAMP];
  __shared__ __attribute__((aligned(16))) float s_xch[NW * XCH_FLOATS];
  __shared__ float s_c[FCOMP * MAXF];
  __shared__ float s_fbwT[MAXW * MAXF];
  __shared__ float s_dctT[MAXF * MAXF];
  // (pad0 = 1: the plan is the reference's geometry - 20 filters of at most 26 bins that end inside the padded power row)
  if (a.plan->pad0 == 1)
    lfcc_body<20, 26>(a, s_pcm, s_xch, s_c, s_fbwT, s_dctT);
  else
    lfcc_body<0, 0>(a, s_pcm, s_xch, s_c, s_fbwT, s_dctT);
}

// ---- in-place pre-emphasis (the reference mutates its input, :106) ----------
constexpr int PE_CHUNK = 4096;
__global__ void preemph_save_kernel(const float* x, int L, int nchunk, int total, float* ws) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over B*nchunk
  if (idx >= total) return;
  const int b = idx / nchunk, c = idx - b * nchunk;
  const long n = (long)c * PE_CHUNK - 1;
  ws[idx] = n >= 0 ? x[(size_t)b * L + n] : 0.0f;
}
__global__ __launch_bounds__(256) void preemph_apply_kernel(float* x, int L, int nchunk, float coef,
                                                            const float* ws) {
  __shared__ float s[PE_CHUNK + 1];
  const int b = blockIdx.x / nchunk, c = blockIdx.x - b * nchunk;
  float* row = x + (size_t)b * L;
  const long n0 = (long)c * PE_CHUNK;
  const int cnt = (int)min((long)PE_CHUNK, L - n0);
  if (threadIdx.x == 0) s[0] = ws[blockIdx.x];
  for (int e = threadIdx.x; e < cnt; e += 256) s[e + 1] = row[n0 + e];
  __syncthreads();
  for (int e = threadIdx.x; e < cnt; e += 256) {
    if (n0 + e == 0) continue;  // y[0] = x[0]
    row[n0 + e] = __fsub_rn(s[e + 1], __fmul_rn(coef, s[e]));
  }
}

// ---- (B,T,D) -> (B,D,feat_len): repeat-pad / chop + transpose ----------------
__global__ __launch_bounds__(256) void pad_transpose_kernel(const float* feat, int T, int D,
                                                            float* out, int flen,
                                                            const int* start, int pad_mode,
                                                            const float* silence) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int tp0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int st = (start != nullptr && T > flen) ? min(max(start[b], 0), T - flen) : 0;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {  // r: frame within tile, tx: coefficient
    const int tp = tp0 + r, c = c0 + tx;
    float v = 0.0f;
    if (tp < flen && c < D) {
      int t;  // source frame, or -1 = zero, -2 = the silence frame
      if (T >= flen) t = tp + st;
      else if (pad_mode == AIR_PAD_REPEAT) t = tp % T;
      else if (pad_mode == AIR_PAD_ZERO) t = tp < T ? tp : -1;
      else t = tp >= flen - T ? tp - (flen - T) : -2;
      v = t >= 0 ? feat[((size_t)b * T + t) * D + c] : (t == -2 ? silence[c] : 0.0f);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {  // r: coefficient, tx: frame
    const int c = c0 + r, tp = tp0 + tx;
    if (c < D && tp < flen) out[((size_t)b * D + c) * flen + tp] = tile[tx][r];
  }
}

int lfcc_launch(const float* pcm, const short* pcm16, int B, int L, float* out, int feat_len, const int* start,
                const void* plan_dev, int flags, hipStream_t stream, int pad_mode = AIR_PAD_REPEAT,
                const float* silence = nullptr) {
  if ((!pcm && !pcm16) || !out || !plan_dev || B <= 0 || L <= 0) return AIR_EINVAL;
  if (pad_mode < AIR_PAD_REPEAT || pad_mode > AIR_PAD_SILENCE) return AIR_EINVAL;  // dataset.py:79 raises ValueError
  if (pad_mode == AIR_PAD_SILENCE && !silence) return AIR_EINVAL;
  const int T = 1 + L / FS;
  LfccArgs a;
  a.pcm = pcm;
  a.pcm16 = pcm16;
  a.out = out;
  a.plan = reinterpret_cast<const LfccPlan*>(plan_dev);
  a.start = start;
  a.silence = silence;
  a.pad_mode = pad_mode;
  a.L = L;
  a.T = T;
  a.tiles = (T + FOUT - 1) / FOUT;
  a.flags = flags;
  a.feat_len = feat_len;
  {
    // algorithmic bytes: fp32 PCM in + fp32 features out
    const int nf = (flags & FLAG_DELTA) ? 3 : 1;
    const double out_frames = (flags & FLAG_PADDED) ? (double)feat_len : (double)T;
    AirProfScope ps(AIR_K_LFCC, B * ((pcm16 ? 2.0 : 4.0) * L + 4.0 * out_frames * nf * 20.0), stream);
    hipLaunchKernelGGL(lfcc_kernel, dim3((unsigned)(B * a.tiles)), dim3(NTHREADS), 0, stream, a);
  }
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

}  // namespace

extern "C" {

size_t air_lfcc_plan_bytes(void) { return sizeof(LfccPlan); }

int air_lfcc_plan_build(const float* fb_host, int nbin, int nfilt, const float* dct_host,
                        const float* window_host, int fl, int fs, int fn, void* plan_host_out) {
  if (!fb_host || !dct_host || !plan_host_out) return AIR_EINVAL;
  if (fl != FL || fs != FS || fn != FN || nbin != NBIN) return AIR_EUNSUPPORTED;
  if (nfilt < 1 || nfilt > MAXF) return AIR_EUNSUPPORTED;
  LfccPlan* p = reinterpret_cast<LfccPlan*>(plan_host_out);
  memset(p, 0, sizeof(LfccPlan));
  p->nfilt = nfilt;
  p->fl = fl;
  p->fs = fs;
  p->fn = fn;
  p->nbin = nbin;
  const double pi = 3.14159265358979323846;
  for (int n = 0; n < FL; ++n)
    p->window[n] = window_host ? window_host[n] : (float)(0.54 - 0.46 * cos(2.0 * pi * n / FL));
  for (int n = 0; n < 256; ++n) {
    p->tw256[2 * n] = (float)cos(2.0 * pi * n / 256.0);
    p->tw256[2 * n + 1] = (float)(-sin(2.0 * pi * n / 256.0));
  }
  for (int q = 0; q < 16; ++q) {
    p->tw512[2 * q] = (float)cos(2.0 * pi * q / 512.0);
    p->tw512[2 * q + 1] = (float)(-sin(2.0 * pi * q / 512.0));
  }
  int maxw = 1;
  for (int j = 0; j < nfilt; ++j) {
    int first = -1, last = -1;
    for (int k = 0; k < nbin; ++k) {
      if (fb_host[(size_t)k * nfilt + j] != 0.0f) {
        if (first < 0) first = k;
        last = k;
      }
    }
    if (first < 0) {
      p->lo[j] = 0;
      p->cnt[j] = 0;
      continue;
    }
    const int cnt = last - first + 1;
    if (cnt > MAXW) return AIR_EUNSUPPORTED;  // filter wider than the sparse table
    p->lo[j] = first;
    p->cnt[j] = cnt;
    if (cnt > maxw) maxw = cnt;
    for (int i = 0; i < cnt; ++i) p->fbwT[i * MAXF + j] = fb_host[(size_t)(first + i) * nfilt + j];
  }
  p->maxw = maxw;
  // the specialised instance of lfcc_kernel (20 filters, <= 26 bins each, every filter's 26-bin window inside the padded
  // power row - PROW floats, zeros behind the Nyquist bin): the reference's LFCC(320, 160, 512, 16000, 20)
  bool fast = nfilt == 20 && maxw <= 26;
  for (int j = 0; j < MAXF; ++j) fast = fast && p->lo[j] + 26 <= PROW;
  p->pad0 = fast ? 1 : 0;
  for (int i = 0; i < nfilt; ++i)
    for (int j = 0; j < nfilt; ++j) p->dctT[j * MAXF + i] = dct_host[(size_t)i * nfilt + j];
  return AIR_OK;
}

int air_lfcc_fwd(const float* pcm, int B, int L, float* out, const void* plan_dev, int flags,
                 air_stream_t stream) {
  return lfcc_launch(pcm, nullptr, B, L, out, 0, nullptr, plan_dev,
                     flags & (AIR_LFCC_EMPHASIS | AIR_LFCC_DELTA), air_stream(stream));
}

int air_lfcc_fwd_padded(const float* pcm, int B, int L, float* out, int feat_len,
                        const int* start_dev, const void* plan_dev, int flags,
                        air_stream_t stream) {
  if (feat_len <= 0) return AIR_EINVAL;
  return lfcc_launch(pcm, nullptr, B, L, out, feat_len, start_dev, plan_dev,
                     (flags & (AIR_LFCC_EMPHASIS | AIR_LFCC_DELTA)) | FLAG_PADDED,
                     air_stream(stream));
}

int air_lfcc_fwd_padded_ex(const float* pcm, const int16_t* pcm16, int B, int L, float* out, int feat_len,
                           const int* start_dev, const void* plan_dev, int flags, int pad_mode,
                           const float* silence_dev, air_stream_t stream) {
  if (feat_len <= 0 || (pcm != nullptr) == (pcm16 != nullptr)) return AIR_EINVAL;
  return lfcc_launch(pcm, reinterpret_cast<const short*>(pcm16), B, L, out, feat_len, start_dev, plan_dev,
                     (flags & (AIR_LFCC_EMPHASIS | AIR_LFCC_DELTA)) | FLAG_PADDED, air_stream(stream), pad_mode,
                     silence_dev);
}

int air_lfcc_fwd_padded_i16(const int16_t* pcm16, int B, int L, float* out, int feat_len,
                            const int* start_dev, const void* plan_dev, int flags, air_stream_t stream) {
  // feat_len <= 0: the (B, T, D) layout of air_lfcc_fwd
  return lfcc_launch(nullptr, reinterpret_cast<const short*>(pcm16), B, L, out, feat_len > 0 ? feat_len : 0,
                     start_dev, plan_dev,
                     (flags & (AIR_LFCC_EMPHASIS | AIR_LFCC_DELTA)) | (feat_len > 0 ? FLAG_PADDED : 0),
                     air_stream(stream));
}

size_t air_preemph_ws_bytes(int B, int L) {
  if (B <= 0 || L <= 0) return 0;
  const size_t nchunk = ((size_t)L + PE_CHUNK - 1) / PE_CHUNK;
  return (size_t)B * nchunk * sizeof(float);
}

int air_preemph_inplace(float* pcm, int B, int L, float coef, void* ws, size_t ws_bytes,
                        air_stream_t stream) {
  if (!pcm || !ws || B <= 0 || L <= 0) return AIR_EINVAL;
  if (ws_bytes < air_preemph_ws_bytes(B, L)) return AIR_EWORKSPACE;
  const int nchunk = (L + PE_CHUNK - 1) / PE_CHUNK;
  const int n = B * nchunk;
  hipLaunchKernelGGL(preemph_save_kernel, dim3((n + 255) / 256), dim3(256), 0, air_stream(stream),
                     pcm, L, nchunk, n, reinterpret_cast<float*>(ws));
  AIR_CHECK_LAUNCH();
  hipLaunchKernelGGL(preemph_apply_kernel, dim3(n), dim3(256), 0, air_stream(stream), pcm, L,
                     nchunk, coef, reinterpret_cast<const float*>(ws));
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_pad_transpose_ex(const float* feat, int B, int T, int D, float* out, int feat_len,
                         const int* start_dev, int pad_mode, const float* silence_dev, air_stream_t stream) {
  if (!feat || !out || B <= 0 || T <= 0 || D <= 0 || feat_len <= 0) return AIR_EINVAL;
  if (pad_mode < AIR_PAD_REPEAT || pad_mode > AIR_PAD_SILENCE) return AIR_EINVAL;
  if (pad_mode == AIR_PAD_SILENCE && !silence_dev) return AIR_EINVAL;
  dim3 grid((feat_len + 63) / 64, (D + 63) / 64, B);
  hipLaunchKernelGGL(pad_transpose_kernel, grid, dim3(256), 0, air_stream(stream), feat, T, D,
                     out, feat_len, start_dev, pad_mode, silence_dev);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_pad_transpose(const float* feat, int B, int T, int D, float* out, int feat_len,
                      const int* start_dev, air_stream_t stream) {
  return air_pad_transpose_ex(feat, B, T, D, out, feat_len, start_dev, AIR_PAD_REPEAT, nullptr, stream);
}

}  // extern "C"
