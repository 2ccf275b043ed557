// Cost of a grid-wide barrier among co-resident workgroups on MI355X ("measure, don't price" - the
// persistent Res2-chain kernel was priced from the guide's 768-workgroup row).  One workgroup per CU (or 2 / 3 / 4), every
// workgroup: N rounds of { optional ~1 us of memory work; barrier }.  Barrier = one device-scope atomic add on a counter
// + spin on a generation word (sense reversal by the last arriver), thread 0 only, __syncthreads on both sides.
// Variants: (a) one counter for the whole grid; (b) hierarchical - a counter per XCD (blockIdx % 8), the last arriver
// of each XCD adds to the global one.   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Bar {
  unsigned count;      // arrivals of the current round
  unsigned pad0[31];
  unsigned gen;        // completed rounds
  unsigned pad1[31];
  unsigned xcount[8 * 32];  // per-XCD arrival counters (one per 128 B)
};

__device__ __forceinline__ void grid_barrier_flat(Bar* b, unsigned nblocks, unsigned round) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (round + 1) * nblocks - 1) {
      __hip_atomic_fetch_max(&b->gen, round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // (max: a give-up mark stays)
    } else {
      unsigned spins = 0;  // (bounded: a grid that is not co-resident must not hang the box)
      while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= round && ++spins < 2000000u) __builtin_amdgcn_s_sleep(1);
      if (spins >= 2000000u) __hip_atomic_store(&b->gen, 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // give up for good
    }
    __threadfence();
  }
  __syncthreads();
}

__device__ __forceinline__ void grid_barrier_xcd(Bar* b, unsigned nblocks, unsigned round) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned x = blockIdx.x & 7u;
    const unsigned per = (nblocks - x + 7u) / 8u;  // workgroups of this XCD (round-robin dispatch)
    const unsigned prev = __hip_atomic_fetch_add(&b->xcount[x * 32], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    bool released = false;
    if (prev == (round + 1) * per - 1) {
      const unsigned p2 = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (p2 == (round + 1) * 8u - 1) {
        __hip_atomic_fetch_max(&b->gen, round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // (max: a give-up mark stays)
        released = true;
      }
    }
    if (!released) {
      unsigned spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= round && ++spins < 2000000u) __builtin_amdgcn_s_sleep(1);
      if (spins >= 2000000u) __hip_atomic_store(&b->gen, 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
  }
  __syncthreads();
}

// variant (c): the arrival is a RELAXED add behind the fence, waiters poll every ~SLEEP x 64 clocks (fewer uncached loads
// competing with the arrivals for the same memory channel)
template <int SLEEP>
__device__ __forceinline__ void grid_barrier_sparse(Bar* b, unsigned nblocks, unsigned round) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (round + 1) * nblocks - 1) {
      __hip_atomic_fetch_max(&b->gen, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= round && ++spins < 400000u) __builtin_amdgcn_s_sleep(SLEEP);
      if (spins >= 400000u) __hip_atomic_store(&b->gen, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
  }
  __syncthreads();
}

template <int KIND>
__global__ __launch_bounds__(256) void k(Bar* b, float* buf, size_t n_per_block, int rounds, int work,
                                         unsigned long long* out) {
  const unsigned long long w0 = wall_clock64();
  float s = 0.0f;
  float* mine = buf + (size_t)blockIdx.x * n_per_block;
  for (int r = 0; r < rounds; ++r) {
    if (work) {  // read + write this workgroup's slice (n_per_block floats): the kind of pass a fused chain stage makes
      for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) {
        const float v = mine[i];
        mine[i] = v * 1.0001f + 1.0f;
        s += v;
      }
    }
    if (KIND == 0) grid_barrier_flat(b, gridDim.x, (unsigned)r);
    else if (KIND == 1) grid_barrier_xcd(b, gridDim.x, (unsigned)r);
    else if (KIND == 3) grid_barrier_sparse<8>(b, gridDim.x, (unsigned)r);
    else if (KIND == 4) grid_barrier_sparse<32>(b, gridDim.x, (unsigned)r);
  }
  const unsigned long long w1 = wall_clock64();
  if (s == 12345.678f) out[0] = 1;
  if (threadIdx.x == 0) out[blockIdx.x + 1] = w1 - w0;
}

int main() {
  Bar* b;
  float* buf;
  unsigned long long* out;
  const int maxb = 1024;
  const size_t per = 16384;  // 64 KB per workgroup when work is on
  hipMalloc(&b, sizeof(Bar));
  hipMalloc(&buf, maxb * per * sizeof(float));
  hipMalloc(&out, (maxb + 1) * 8);
  hipMemset(buf, 0, maxb * per * sizeof(float));
  std::vector<unsigned long long> h(maxb + 1);
  const int rounds = 2000;
  for (int kind : {0, 1, 3, 4})
    for (int work = 0; work < 2; ++work)
      for (int nb : {64, 256, 768}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(b, 0, sizeof(Bar));
          hipEvent_t e0, e1;
          hipEventCreate(&e0);
          hipEventCreate(&e1);
          hipEventRecord(e0, 0);
          if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, b, buf, per, rounds, work, out);
          else if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, b, buf, per, rounds, work, out);
          else if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(256), 0, 0, b, buf, per, rounds, work, out);
          else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, b, buf, per, rounds, work, out);
          hipEventRecord(e1, 0);
          if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
          unsigned g = 0;
          hipMemcpy(&g, &b->gen, 4, hipMemcpyDeviceToHost);
          if (g == 0xffffffffu) { printf("(%d workgroups: barrier timed out - not co-resident)\n", nb); best = -1.0f; break; }
        }
        printf("%s barrier, %4d workgroups x 256 threads, %s: %.2f us per round\n", kind == 0 ? "flat" : kind == 1 ? "per-XCD + global" : kind == 3 ? "flat, relaxed, poll/512clk" : "flat, relaxed, poll/2048clk",
               nb, work ? "64 KB read+write per workgroup and round" : "no work", best * 1e3f / rounds);
        fflush(stdout);
      }
  // the same work as separate launches (what the chain pays today): nb workgroups, one round per launch
  for (int work = 0; work < 2; ++work)
    for (int nb : {256, 768}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipMemset(b, 0, sizeof(Bar));
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, b, buf, per, 1, work, out);
      hipEventRecord(e0, 0);
      const int n = 2000;
      for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, b, buf, per, 1, work, out);  // k<2>: one round, no barrier
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("separate launches, %4d workgroups, %s: %.2f us per launch\n", nb, work ? "64 KB read+write per workgroup" : "no work", ms * 1e3f / n);
    }
  return 0;
}
