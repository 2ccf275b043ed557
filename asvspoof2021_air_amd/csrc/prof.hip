// Per-kernel HIP-event timing (see air_prof.h).  Test/bench instrumentation only.
#include "air_prof.h"

#include <mutex>
#include <vector>

#include "air_common.h"
#include "air_hip.h"

namespace {
struct Rec {
  hipEvent_t a, b;
  int kid;
  double work, issued, bytes;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
bool g_on = false;
Rec g_cur;
const char* const kNames[AIR_K_COUNT] = {
    "conv_fwd_kernel<3,3,1>", "conv_fwd_kernel<3,3,2>", "conv_fwd_kernel<1,1,1>",
    "conv_fwd_kernel<1,1,2>", "conv_wgrad_kernel<3,3,1,64>", "conv_wgrad_kernel<3,3,2,32>",
    "conv_wgrad_kernel<1,1,1,64>", "conv_wgrad_kernel<1,1,2,32>",
    "conv_fwd_kernel<parity class>", "conv_fwd_kernel<conv1d>",
    "conv_wgrad_kernel<conv1d>", "lfcc_kernel", "wino_conv_kernel", "wino_wgrad_kernel",
    "c1b_fwd_kernel", "c1b_gemm_kernel", "c1b_tap_kernel", "wino4_conv_kernel",
    "c1b_tapw_kernel", "wino4_conv_kernel+bn", "conv_s2_dgrad_kernel", "conv_s2_bf3_kernel", "conv_s2d_bf3_kernel",
    "conv_s2w_bf3_kernel"};

// Test instrumentation (tests/test_cu_mask_gpu.py): a kernel that holds `nblocks` compute units for a while - each
// workgroup takes `lds_bytes` of LDS (so that a 144 KB persistent Winograd workgroup cannot share its CU) and spins on
// the 100 MHz wall clock - standing in for the RCCL kernels that are resident while the backward pass runs.
__global__ void cu_hog_kernel(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned hog_lds[];
  hog_lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  unsigned acc = hog_lds[(threadIdx.x * 7) & 63];
  while (wall_clock64() - t0 < ticks) acc = acc * 1664525u + 1013904223u;
  if (acc == 0x12345u && sink) *sink = acc;  // (keeps the loop)
}

// Measurement instrumentation (bench.py "core_clock"): ONE wave that samples {100 MHz wall clock, s_memtime core clock
// counter} every `interval` wall ticks while other kernels run beside it on another stream - the clock the compute
// units really run at under a given kernel mix (amd-smi answers with a slow average; the MFMA-dense kernels are
// power-capped well below the 2.4 GHz the peaks are quoted at).  out[2 i] = wall ticks, out[2 i + 1] = core clocks.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int n, unsigned long long interval) {
  if (threadIdx.x != 0) return;
  unsigned long long next = wall_clock64();
  for (int i = 0; i < n; ++i) {
    unsigned long long w;
    while ((w = wall_clock64()) < next) __builtin_amdgcn_s_sleep(32);
    out[2 * i] = w;
    out[2 * i + 1] = __builtin_readcyclecounter();
    next += interval;
  }
}
}  // namespace

bool air_prof_on() { return g_on; }

void air_prof_begin(int kid, double work, hipStream_t st, double issued, double bytes) {
  g_cur.bytes = bytes;
  g_cur.kid = kid;
  g_cur.work = work;
  g_cur.issued = issued < 0.0 ? work : issued;
  if (hipEventCreate(&g_cur.a) != hipSuccess || hipEventCreate(&g_cur.b) != hipSuccess) return;
  (void)hipEventRecord(g_cur.a, st);
}

void air_prof_end(hipStream_t st) {
  (void)hipEventRecord(g_cur.b, st);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(g_cur);
}

extern "C" {

int air_debug_cu_hog(int nblocks, int lds_bytes, double milliseconds, air_stream_t stream) {
  if (nblocks <= 0 || lds_bytes < 256 || lds_bytes > 160 * 1024 || milliseconds <= 0.0 || milliseconds > 200.0)
    return AIR_EINVAL;
  static bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(cu_hog_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  if (!attr_ok) return AIR_ELAUNCH;
  hipLaunchKernelGGL(cu_hog_kernel, dim3(nblocks), dim3(64), (size_t)lds_bytes, air_stream(stream),
                     (unsigned long long)(milliseconds * 1e5), (unsigned*)nullptr);
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_debug_clock_probe(unsigned long long* out, int n_samples, double interval_us, air_stream_t stream) {
  if (!out || n_samples < 2 || n_samples > (1 << 20) || interval_us < 1.0 || interval_us * n_samples > 5e6) return AIR_EINVAL;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, air_stream(stream), out, n_samples,
                     (unsigned long long)(interval_us * 100.0));
  AIR_CHECK_LAUNCH();
  return AIR_OK;
}

int air_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  g_on = on != 0;
  return AIR_OK;
}

int air_prof_kernel_count(void) { return AIR_K_COUNT; }

const char* air_prof_kernel_name(int kid) {
  return (kid >= 0 && kid < AIR_K_COUNT) ? kNames[kid] : "";
}

/* Totals for kernel id `kid` since air_prof_enable(1): launches, summed event
 * time (ms) and summed algorithmic work (FLOPs or bytes).  Synchronises. */
int air_prof_collect2(int kid, int* launches, double* total_ms, double* total_work, double* total_issued,
                      double* total_bytes);
int air_prof_collect(int kid, int* launches, double* total_ms, double* total_work) {
  double issued = 0.0, bytes = 0.0;
  return air_prof_collect2(kid, launches, total_ms, total_work, &issued, &bytes);
}

/* The same plus the FLOPs the launches issued to the matrix pipe (== total_work for direct kernels). */
int air_prof_collect2(int kid, int* launches, double* total_ms, double* total_work, double* total_issued,
                      double* total_bytes) {
  if (!launches || !total_ms || !total_work || !total_issued || !total_bytes) return AIR_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  double ms = 0.0, work = 0.0, issued = 0.0, bytes = 0.0;
  for (auto& r : g_recs) {
    if (r.kid != kid) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) return AIR_ELAUNCH;
    float t = 0.0f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return AIR_ELAUNCH;
    ++n;
    ms += t;
    work += r.work;
    issued += r.issued;
    bytes += r.bytes;
  }
  *launches = n;
  *total_ms = ms;
  *total_work = work;
  *total_issued = issued;
  *total_bytes = bytes;
  return AIR_OK;
}

}  // extern "C"
