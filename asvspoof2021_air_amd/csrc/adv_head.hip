// Adversarial channel-classifier head: the small kernels around the two
// nn.Linear layers of model.ChannelClassifier (model.py:1007-1023) and nn.CrossEntropyLoss
// (main_train.py:251, :386, :396-397, :428, :446-450).  All latency-class: B <= a few hundred
// rows of <= 128 floats.
#include <cstdint>

#include "air_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u;
  k[1] += 0xBB67AE85u;
}

// nn.Dropout(p) keep-mask, already scaled: keep = (u >= p) / (1 - p), u ~ U[0,1) from Philox4x32-10
__global__ __launch_bounds__(NT) void dropout_mask_kernel(float* __restrict__ keep, size_t n, float p,
                                                          uint64_t seed, uint64_t offset) {
  const size_t quad = (size_t)blockIdx.x * NT + threadIdx.x;
  if (quad * 4 >= n) return;
  const uint64_t ctr = offset + quad;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
  const float scale = 1.0f / (1.0f - p);
  for (int j = 0; j < 4; ++j)
    if (quad * 4 + j < n) keep[quad * 4 + j] = ((float)c[j] * 2.3283064365386963e-10f >= p) ? scale : 0.0f;
}

// y = relu(x * keep)   (Dropout -> ReLU, model.py:1013-1014; keep NULL in eval mode)
__global__ __launch_bounds__(NT) void mask_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ keep,
                                                           size_t n, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i < n) y[i] = fmaxf(keep ? x[i] * keep[i] : x[i], 0.0f);
}

// dx = alpha * dy * keep * (y > 0)
__global__ __launch_bounds__(NT) void mask_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ keep, size_t n, float alpha,
                                                           float* __restrict__ dx) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.0f ? alpha * dy[i] * (keep ? keep[i] : 1.0f) : 0.0f;
}

__global__ __launch_bounds__(NT) void scale_kernel(float* __restrict__ x, size_t n, float alpha) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i < n) x[i] *= alpha;
}

// One workgroup: probs = softmax(logits) per row, loss = mean_b -log probs[b][label[b]], plus the
// count of rows whose argmax equals the label (the accuracy counters of main_train.py:383-385).
__global__ __launch_bounds__(NT) void softmax_ce_fwd_kernel(const float* __restrict__ logits,
                                                            const long long* __restrict__ labels, int B, int C,
                                                            float* __restrict__ probs, float* __restrict__ loss,
                                                            int* __restrict__ correct) {
  __shared__ double sh[NT / 64];
  __shared__ int shc[NT / 64];
  double acc = 0.0;
  int hit = 0;
  for (int b = threadIdx.x; b < B; b += NT) {
    const float* __restrict__ row = logits + (size_t)b * C;
    float m = row[0];
    int am = 0;
    for (int c = 1; c < C; ++c)
      if (row[c] > m) { m = row[c]; am = c; }  // first maximum, like torch.max
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(row[c] - m);
    const float inv = 1.0f / s;
    for (int c = 0; c < C; ++c) probs[(size_t)b * C + c] = expf(row[c] - m) * inv;
    const int lab = (int)labels[b];
    acc += (double)(logf(s) - (row[lab] - m));
    hit += am == lab;
  }
  acc = air_wave_sum_d(acc);
  for (int o = 32; o > 0; o >>= 1) hit += __shfl_xor(hit, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[wave] = acc; shc[wave] = hit; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    int h = 0;
    for (int w = 0; w < NT / 64; ++w) { t += sh[w]; h += shc[w]; }
    *loss = (float)(t / B);
    if (correct) *correct = h;
  }
}

// dlogits[b][c] = g * (probs[b][c] - [c == label[b]]) / B
__global__ __launch_bounds__(NT) void softmax_ce_bwd_kernel(const float* __restrict__ probs,
                                                            const long long* __restrict__ labels, int B, int C,
                                                            const float* __restrict__ gscale, float* __restrict__ dlogits) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  const float g = (gscale ? *gscale : 1.0f) / (float)B;
  dlogits[i] = g * (probs[i] - (c == (int)labels[b] ? 1.0f : 0.0f));
}

inline unsigned nblk(size_t n) { return (unsigned)((n + NT - 1) / NT); }

}  // namespace

extern "C" {

int air_dropout_mask(float* keep, size_t n, float p, uint64_t seed, uint64_t offset, air_stream_t stream) {
  if (!keep || n == 0 || !(p >= 0.0f) || !(p < 1.0f)) return AIR_EINVAL;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(nblk((n + 3) / 4)), dim3(NT), 0, air_stream(stream), keep, n, p, seed,
                     offset);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_mask_relu_fwd(const float* x, const float* keep, size_t n, float* y, air_stream_t stream) {
  if (!x || !y || n == 0) return AIR_EINVAL;
  hipLaunchKernelGGL(mask_relu_fwd_kernel, dim3(nblk(n)), dim3(NT), 0, air_stream(stream), x, keep, n, y);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_mask_relu_bwd(const float* dy, const float* y, const float* keep, size_t n, float alpha, float* dx,
                      air_stream_t stream) {
  if (!dy || !y || !dx || n == 0) return AIR_EINVAL;
  hipLaunchKernelGGL(mask_relu_bwd_kernel, dim3(nblk(n)), dim3(NT), 0, air_stream(stream), dy, y, keep, n, alpha, dx);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_scale(float* x, size_t n, float alpha, air_stream_t stream) {
  if (!x || n == 0) return AIR_EINVAL;
  hipLaunchKernelGGL(scale_kernel, dim3(nblk(n)), dim3(NT), 0, air_stream(stream), x, n, alpha);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_softmax_ce_fwd(const float* logits, const long long* labels, int B, int C, float* probs, float* loss,
                       int* correct_or_null, air_stream_t stream) {
  if (!logits || !labels || !probs || !loss || B <= 0 || C <= 0) return AIR_EINVAL;
  hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3(1), dim3(NT), 0, air_stream(stream), logits, labels, B, C, probs, loss,
                     correct_or_null);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_softmax_ce_bwd(const float* probs, const long long* labels, int B, int C, const float* gscale_or_null,
                       float* dlogits, air_stream_t stream) {
  if (!probs || !labels || !dlogits || B <= 0 || C <= 0) return AIR_EINVAL;
  hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3(nblk((size_t)B * C)), dim3(NT), 0, air_stream(stream), probs, labels,
                     B, C, gscale_or_null, dlogits);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

}  // extern "C"
