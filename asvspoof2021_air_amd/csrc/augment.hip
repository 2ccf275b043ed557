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
#include "air_common.h"

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

}  // namespace

extern "C" {

size_t air_ir_convolve_ws_bytes(int B) { return B > 0 ? (size_t)B * 2 * sizeof(unsigned) + 256 : 0; }

int air_ir_convolve(const float* x, int B, int L, const float* irs, int n_ir, int H, const int* ir_idx,
                    int normalize, float* y, void* ws, size_t ws_bytes, air_stream_t stream) {
  if (!x || !y || !irs || B <= 0 || L <= 0 || n_ir <= 0 || H <= 0 || x == y) return AIR_EINVAL;
  if (normalize && (!ws || ws_bytes < air_ir_convolve_ws_bytes(B))) return AIR_EWORKSPACE;
  hipStream_t st = air_stream(stream);
  unsigned* peaks = normalize ? reinterpret_cast<unsigned*>(ws) : nullptr;
  if (peaks && hipMemsetAsync(peaks, 0, (size_t)B * 2 * sizeof(unsigned), st) != hipSuccess) return AIR_ELAUNCH;
  const dim3 grid((L + FIR_BLK - 1) / FIR_BLK, B);
  hipLaunchKernelGGL(fir_kernel, grid, dim3(FIR_NT), 0, st, x, L, irs, H, ir_idx, y, peaks, n_ir);
  AIR_CHECK_LAUNCH();
  if (normalize) {
    hipLaunchKernelGGL(fir_rescale_kernel, dim3(16, B), dim3(256), 0, st, y, L, ir_idx, peaks);
    AIR_CHECK_LAUNCH();
  }
  return AIR_OK;
}

}  // extern "C"
