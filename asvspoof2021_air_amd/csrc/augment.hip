// On-the-fly channel augmentation in the front-end (BASELINE.json configs[4]):
// y_b = (x_b * h_{idx[b]})[:L], optionally rescaled so max|y_b| = max|x_b| ("safe": no clipping, no
// level change).  The reference does this OFFLINE by shelling out to idiap/acoustic-simulator's
// degrade-audio-safe-random.py (channel_simulation/simulated_device.py:33-35,46-50,57-61), a tool
// that is not vendored: PARITY UNPINNED - the arithmetic here follows this repo's own spec
// (oracle/channel.py, checked against scipy.signal.fftconvolve).
//
// Direct time-domain FIR on the fp32 VALU (2*L*H FLOP per utterance: 131 MFLOP at H = 1024 taps;
// HBM traffic is only 2 x 256 KB per utterance, so this is compute-bound, not a byte mover):
//   * a workgroup owns 2048 consecutive outputs of one utterance, a thread 8 consecutive ones;
//   * taps are consumed in chunks of 1024: the x segment a chunk needs (3071 samples) and the
//     tap chunk are staged in LDS once, zero-filled outside [0, L);
//   * per 8 taps a thread holds a 16-sample register window (two aligned groups of 8), does 64
//     FMAs, and slides by ONE new group: 2 ds_read_b128 of x + 2 broadcast ds_read_b128 of taps
//     per 64 FMAs;
//   * LDS layout: every group of 8 samples sits at a 48-byte stride, so the 16 lanes a
//     ds_read_b128 cycle serves (thread t reads group t - jb + const) cover all 64 banks.
//
// Round 5: impulse responses of 128 .. 1025 taps (the bank's 1024) go through OVERLAP-SAVE FFT convolution instead -
// 26 x fewer floating-point operations than the direct form at H = 1024 (the direct kernel runs at 71 % of the fp32
// VALU rate: packed fp32 issues at half rate on this part, there was nothing left in it).  One workgroup transforms TWO
// consecutive 3072-sample output blocks of one utterance at once, as the real and imaginary parts of one 4096-point
// complex FFT (h is real, so Re / Im of the product's inverse are the two blocks' convolutions): 4096 = 16 x 16 x 16,
// three radix-16 passes in registers (air_fft16.h, the LFCC kernel's butterfly) with the data crossing LDS between
// them; the forward transform leaves the spectrum digit-reversed, the IR spectra are stored in the same order and the
// inverse passes run the other way round, so nothing is ever re-ordered; the first pass reads PCM straight from
// global memory and the last one writes y straight to it.  Same result as the direct form to fp32 FFT rounding
// (~3e-7 of the output scale; tests/test_augment.py holds both to the same bound).
#include "air_common.h"
#include "air_fft16.h"
#include "air_options.h"

namespace {

constexpr int FIR_NT = 256, FIR_R = 8, FIR_BLK = FIR_NT * FIR_R, FIR_KC = 1024;
constexpr int FIR_GROUPS = (FIR_BLK + FIR_KC) / 8;  // 384 groups of 8 samples per staged segment
constexpr int FIR_GS = 12;                           // floats per group slot (8 data + 4 pad)

__device__ __forceinline__ void atomic_max_pos(unsigned* p, float v) {
  atomicMax(p, __float_as_uint(v));  // v >= 0: unsigned order == float order
}

__global__ __launch_bounds__(FIR_NT) void fir_kernel(const float* __restrict__ x, int L,
                                                     const float* __restrict__ irs, int H,
                                                     const int* __restrict__ idx, float* __restrict__ y,
                                                     unsigned* __restrict__ peaks, int n_ir) {
  __shared__ __attribute__((aligned(16))) float xs[FIR_GROUPS * FIR_GS];
  __shared__ __attribute__((aligned(16))) float hs[FIR_KC];
  const int b = blockIdx.y, n0 = blockIdx.x * FIR_BLK, t = threadIdx.x;
  const float* __restrict__ xb = x + (size_t)b * L;
  float* __restrict__ yb = y + (size_t)b * L;
  const int ir = idx ? min(idx[b], n_ir - 1) : 0;  // indices are caller data: never read past the IR table
  if (ir < 0) {  // pass-through utterance
    for (int i = t; i < FIR_BLK; i += FIR_NT)
      if (n0 + i < L) yb[n0 + i] = xb[n0 + i];
    return;
  }
  const float* __restrict__ hb = irs + (size_t)ir * H;
  float out[FIR_R];
#pragma unroll
  for (int r = 0; r < FIR_R; ++r) out[r] = 0.0f;
  float xpeak = 0.0f;

  for (int kc = 0; kc < H; kc += FIR_KC) {
    __syncthreads();
    // staged position p <-> sample m = n0 - kc - (FIR_KC - 1) + p
    const int m0 = n0 - kc - (FIR_KC - 1);
    for (int p = t; p < FIR_GROUPS * 8; p += FIR_NT) {
      const int m = m0 + p;
      const float v = (m >= 0 && m < L) ? xb[m] : 0.0f;
      xs[(p >> 3) * FIR_GS + (p & 7)] = v;
      if (kc == 0 && p >= FIR_KC - 1) xpeak = fmaxf(xpeak, fabsf(v));  // this block's own samples
    }
    for (int j = t; j < FIR_KC; j += FIR_NT) hs[j] = kc + j < H ? hb[kc + j] : 0.0f;
    __syncthreads();
    const int nj = (min(FIR_KC, H - kc) + 7) >> 3;
    // window: whi = group (t + 127 - jb) + 1, wlo = group (t + 127 - jb); see header
    float wlo[8], whi[8];
    {
      const float4* g = reinterpret_cast<const float4*>(xs + (t + 128) * FIR_GS);
      const float4 a = g[0], c = g[1];
      whi[0] = a.x; whi[1] = a.y; whi[2] = a.z; whi[3] = a.w;
      whi[4] = c.x; whi[5] = c.y; whi[6] = c.z; whi[7] = c.w;
    }
    for (int jb = 0; jb < nj; ++jb) {
      const float4* g = reinterpret_cast<const float4*>(xs + (t + 127 - jb) * FIR_GS);
      const float4 a = g[0], c = g[1];
      wlo[0] = a.x; wlo[1] = a.y; wlo[2] = a.z; wlo[3] = a.w;
      wlo[4] = c.x; wlo[5] = c.y; wlo[6] = c.z; wlo[7] = c.w;
      const float4 h0 = reinterpret_cast<const float4*>(hs + jb * 8)[0];
      const float4 h1 = reinterpret_cast<const float4*>(hs + jb * 8)[1];
      const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < FIR_R; ++r) {
          const int w = r - i + 7;  // 0..14 within [wlo | whi]
          out[r] = fmaf(h[i], w < 8 ? wlo[w] : whi[w - 8], out[r]);
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) whi[i] = wlo[i];
    }
  }
  float ypeak = 0.0f;
#pragma unroll
  for (int r = 0; r < FIR_R; ++r) {
    const int n = n0 + t * FIR_R + r;
    if (n < L) {
      yb[n] = out[r];
      ypeak = fmaxf(ypeak, fabsf(out[r]));
    }
  }
  if (peaks) {
    xpeak = air_wave_max(xpeak);
    ypeak = air_wave_max(ypeak);
    if ((t & 63) == 0) {
      atomic_max_pos(peaks + 2 * b, xpeak);
      atomic_max_pos(peaks + 2 * b + 1, ypeak);
    }
  }
}

// y_b *= max|x_b| / max|y_b|  (augmented utterances only)
__global__ __launch_bounds__(256) void fir_rescale_kernel(float* __restrict__ y, int L, const int* __restrict__ idx,
                                                          const unsigned* __restrict__ peaks) {
  const int b = blockIdx.y;
  if (idx && idx[b] < 0) return;
  const float px = __uint_as_float(peaks[2 * b]), py = __uint_as_float(peaks[2 * b + 1]);
  if (!(py > 0.0f)) return;
  const float g = px / py;
  float* __restrict__ yb = y + (size_t)b * L;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) yb[i] *= g;
}

// ---- overlap-save FFT form -------------------------------------------------------------------------------------------
constexpr int FC_N = 4096;            // complex transform length = real block length (two real blocks per transform)
constexpr int FC_V = 3072;            // outputs kept per real block
constexpr int FC_OV = FC_N - FC_V;    // 1024 samples of history in front of a block: taps - 1 <= 1024
constexpr int FC_NT = 256;
constexpr int FC_MINH = 128, FC_MAXH = FC_OV + 1;
constexpr int FC_LDS = FC_N + FC_N / 16;  // one pad element per 16: every pass reads and writes conflict-free

__device__ __forceinline__ int fc_pad(int i) { return i + (i >> 4); }
__device__ __forceinline__ cf fc_conj(cf a) { return a * cf{1.0f, -1.0f}; }

template <bool INV>
__device__ __forceinline__ void fc_fft16(cf (&v)[16]) {  // INV: sum_k v[k] w16^(-q k) = conj(FFT(conj v))
  if (INV) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = fc_conj(v[j]);
  }
  fft16<false>(v);
  if (INV) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = fc_conj(v[j]);
  }
}

// tw[j] = e^(-2 pi i j / 4096)
__global__ __launch_bounds__(256) void fc_twiddle_kernel(cf* __restrict__ tw) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  double s, c;
  sincospi((double)j / 2048.0, &s, &c);
  tw[j] = cf{(float)c, (float)-s};
}

// Forward transform.  In: v[j] = z[256 j + t].  Out: v[k3] = Z[k1 + 16 k2 + 256 k3] with k1 = t >> 4, k2 = t & 15 - the
// spectrum element that lives at position 256 k1 + 16 k2 + k3 ("digit-reversed").
__device__ __forceinline__ void fc_forward(cf (&v)[16], cf* __restrict__ buf, const cf* __restrict__ tw, int t) {
  const int hi = t >> 4, lo = t & 15;
  fft16<false>(v);  // over a (n = 256 a + b, b = t) -> k1
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw[t * k]);  // w4096^(b k1)
#pragma unroll
  for (int k = 0; k < 16; ++k) buf[fc_pad(k * 256 + t)] = v[k];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = buf[fc_pad(hi * 256 + 16 * c + lo)];  // k1 = hi, b = 16 c + d, d = lo
  fft16<false>(v);  // over c -> k2
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw[16 * lo * k]);  // w256^(d k2)
#pragma unroll
  for (int k = 0; k < 16; ++k) buf[fc_pad(hi * 256 + 16 * k + lo)] = v[k];  // (the positions this thread has just read)
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 16; ++d) v[d] = buf[fc_pad(hi * 256 + 16 * lo + d)];  // k1 = hi, k2 = lo
  fft16<false>(v);  // over d -> k3
}

// Inverse of fc_forward (unscaled).  In: v[k3] as fc_forward leaves it.  Out: v[a] = sum over the spectrum for n = 256 a + t.
__device__ __forceinline__ void fc_inverse(cf (&v)[16], cf* __restrict__ buf, const cf* __restrict__ tw, int t) {
  const int hi = t >> 4, lo = t & 15;
  fc_fft16<true>(v);  // over k3 -> d
#pragma unroll
  for (int d = 1; d < 16; ++d) v[d] = cmul(v[d], fc_conj(tw[16 * d * lo]));  // w256^(-d k2), k2 = lo
#pragma unroll
  for (int d = 0; d < 16; ++d) buf[fc_pad(hi * 256 + 16 * lo + d)] = v[d];  // (read by this thread only, in fc_forward)
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = buf[fc_pad(hi * 256 + 16 * k + lo)];  // k1 = hi, d = lo, over k2
  fc_fft16<true>(v);  // -> c
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (16 * c + lo > 0) v[c] = cmul(v[c], fc_conj(tw[(16 * c + lo) * hi]));  // w4096^(-b k1)
#pragma unroll
  for (int c = 0; c < 16; ++c) buf[fc_pad(hi * 256 + 16 * c + lo)] = v[c];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = buf[fc_pad(k * 256 + t)];  // b = t, over k1
  fc_fft16<true>(v);  // -> a
}

// spectra of the impulse responses, in fc_forward's order, scaled by 1 / 4096 (the inverse is unscaled)
__global__ __launch_bounds__(FC_NT) void fc_spectrum_kernel(const float* __restrict__ irs, int H, const cf* __restrict__ tw,
                                                            cf* __restrict__ spec) {
  __shared__ cf buf[FC_LDS];
  const int t = threadIdx.x;
  const float* __restrict__ h = irs + (size_t)blockIdx.x * H;
  cf v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int n = 256 * j + t;
    v[j] = cf{n < H ? h[n] : 0.0f, 0.0f};
  }
  fc_forward(v, buf, tw, t);
  cf* __restrict__ o = spec + (size_t)blockIdx.x * FC_N + 16 * t;
#pragma unroll
  for (int k = 0; k < 16; ++k) o[k] = v[k] * (1.0f / (float)FC_N);
}

// blockIdx.x = pair of output blocks (2 p, 2 p + 1) of utterance blockIdx.y
__global__ __launch_bounds__(FC_NT) void fc_convolve_kernel(const float* __restrict__ x, int L, const cf* __restrict__ spec,
                                                            const cf* __restrict__ tw, const int* __restrict__ idx,
                                                            float* __restrict__ y, unsigned* __restrict__ peaks, int n_ir) {
  __shared__ cf buf[FC_LDS];
  const int b = blockIdx.y, t = threadIdx.x;
  const float* __restrict__ xb = x + (size_t)b * L;
  float* __restrict__ yb = y + (size_t)b * L;
  const int baseA = 2 * blockIdx.x * FC_V, baseB = baseA + FC_V;  // first output sample of either block
  const int ir = idx ? min(idx[b], n_ir - 1) : 0;
  if (ir < 0) {  // pass-through utterance
    for (int i = baseA + t; i < min(baseA + 2 * FC_V, L); i += FC_NT) yb[i] = xb[i];
    return;
  }
  cf v[16];
  float xpeak = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int mA = baseA - FC_OV + 256 * j + t, mB = mA + FC_V;
    const float a = (mA >= 0 && mA < L) ? xb[mA] : 0.0f;
    const float c = mB < L ? xb[mB] : 0.0f;
    v[j] = cf{a, c};
    if (j >= FC_OV / 256) xpeak = fmaxf(xpeak, fmaxf(fabsf(a), fabsf(c)));  // the blocks' own samples
  }
  fc_forward(v, buf, tw, t);
  const cf* __restrict__ hs = spec + (size_t)ir * FC_N + 16 * t;
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = cmul(v[k], hs[k]);
  fc_inverse(v, buf, tw, t);
  float ypeak = 0.0f;
#pragma unroll
  for (int a = FC_OV / 256; a < 16; ++a) {  // the samples behind the history: block A in the real, block B in the imaginary part
    const int o = 256 * a + t - FC_OV;
    if (baseA + o < L) {
      yb[baseA + o] = v[a][0];
      ypeak = fmaxf(ypeak, fabsf(v[a][0]));
    }
    if (baseB + o < L) {
      yb[baseB + o] = v[a][1];
      ypeak = fmaxf(ypeak, fabsf(v[a][1]));
    }
  }
  if (peaks) {
    xpeak = air_wave_max(xpeak);
    ypeak = air_wave_max(ypeak);
    if ((t & 63) == 0) {
      atomic_max_pos(peaks + 2 * b, xpeak);
      atomic_max_pos(peaks + 2 * b + 1, ypeak);
    }
  }
}

inline size_t fc_align(size_t v) { return (v + 255) & ~(size_t)255; }
inline bool fc_wanted(int H) { return air_opt(AIR_OPT_IR_FFT) != 0 && H >= FC_MINH && H <= FC_MAXH; }

}  // namespace

extern "C" {

size_t air_ir_convolve_ws_bytes(int B) { return B > 0 ? (size_t)B * 2 * sizeof(unsigned) + 256 : 0; }

size_t air_ir_convolve_ws_bytes_ex(int B, int n_ir, int H) {
  if (B <= 0 || n_ir <= 0 || H <= 0) return 0;
  size_t n = fc_align(air_ir_convolve_ws_bytes(B));
  if (fc_wanted(H)) n += fc_align((size_t)FC_N * sizeof(cf)) + (size_t)n_ir * FC_N * sizeof(cf);
  return n;
}

int air_ir_convolve(const float* x, int B, int L, const float* irs, int n_ir, int H, const int* ir_idx,
                    int normalize, float* y, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!x || !y || !irs || B <= 0 || L <= 0 || n_ir <= 0 || H <= 0 || x == y) return AIR_EINVAL;
  if (normalize && (!ws || ws_bytes < air_ir_convolve_ws_bytes(B))) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  unsigned* peaks = normalize ? reinterpret_cast<unsigned*>(ws) : nullptr;
  if (peaks && hipMemsetAsync(peaks, 0, (size_t)B * 2 * sizeof(unsigned), st) != hipSuccess) return AIR_ELAUNCH;
  if (fc_wanted(H) && ws && ws_bytes >= air_ir_convolve_ws_bytes_ex(B, n_ir, H)) {
    // (the tables are rebuilt per call - 16 + n_ir small workgroups - rather than cached against a bank the caller may
    // rewrite in place)
    char* base = reinterpret_cast<char*>(ws) + fc_align(air_ir_convolve_ws_bytes(B));
    cf* tw = reinterpret_cast<cf*>(base);
    cf* spec = reinterpret_cast<cf*>(base + fc_align((size_t)FC_N * sizeof(cf)));
    hipLaunchKernelGGL(fc_twiddle_kernel, dim3(FC_N / 256), dim3(256), 0, st, tw);
    AIR_CHECK_LAUNCH();
    hipLaunchKernelGGL(fc_spectrum_kernel, dim3(n_ir), dim3(FC_NT), 0, st, irs, H, tw, spec);
    AIR_CHECK_LAUNCH();
    const int nblk = (L + FC_V - 1) / FC_V;
    hipLaunchKernelGGL(fc_convolve_kernel, dim3((nblk + 1) / 2, B), dim3(FC_NT), 0, st, x, L, spec, tw, ir_idx, y, peaks, n_ir);
    AIR_CHECK_LAUNCH();
  } else {
    const dim3 grid((L + FIR_BLK - 1) / FIR_BLK, B);
    hipLaunchKernelGGL(fir_kernel, grid, dim3(FIR_NT), 0, st, x, L, irs, H, ir_idx, y, peaks, n_ir);
    AIR_CHECK_LAUNCH();
  }
  if (normalize) {
    hipLaunchKernelGGL(fir_rescale_kernel, dim3(16, B), dim3(256), 0, st, y, L, ir_idx, peaks);
    AIR_CHECK_LAUNCH();
  }
  return AIR_OK;
}

}  // extern "C"
