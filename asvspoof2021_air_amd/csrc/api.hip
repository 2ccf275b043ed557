// Library identification and the dispatch-option table of the C-ABI (include/air_hip.h).
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "air_common.h"
#include "air_options.h"

namespace {
struct OptDef {
  const char* name;  // option name = environment variable without the AIR_ prefix
  int def;
};
const OptDef kOpts[AIR_OPT_COUNT] = {
    {"NO_WINO4", 0},   {"NO_WINOGRAD", 0},       {"WINO4_SPLIT", 1},       {"WINO4_TH3", 2},
    {"WINO4_XCD", 2},  {"CONV_MT", 0},           {"WGRAD_WGS", 256},       {"WINO_WGRAD_WGS", 256},
    {"DIRECT_WGRAD_ROWS", 1}, {"C1B_PS", 7},     {"C1B_GEMM_PS", 1}, {"SKINNY_WGRAD", 1},
    {"CONV_S2", 31}, {"WINO4_DEPHASE", 0}, {"IR_FFT", 1}, {"TAP_ROWS", 0},
};
std::atomic<int> g_val[AIR_OPT_COUNT];
std::once_flag g_once;
void init_opts() {
  for (int i = 0; i < AIR_OPT_COUNT; ++i) {
    char env[64] = "AIR_";
    strncat(env, kOpts[i].name, sizeof(env) - 5);
    const char* v = getenv(env);
    g_val[i].store(v ? atoi(v) : kOpts[i].def, std::memory_order_relaxed);
  }
}
int find_opt(const char* name) {
  if (!name) return -1;
  if (strncmp(name, "AIR_", 4) == 0) name += 4;
  for (int i = 0; i < AIR_OPT_COUNT; ++i)
    if (strcmp(name, kOpts[i].name) == 0) return i;
  return -1;
}
}  // namespace

static int query_stream_cus(int dev, hipStream_t st) {
  int total = 0;
  if (hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || total <= 0) total = 256;
  int cus = total;
  uint32_t mask[32] = {};
  if (hipExtStreamGetCUMask(st, 32, mask) == hipSuccess) {
    int bits = 0;
    for (int i = 0; i < 32; ++i) bits += __builtin_popcount(mask[i]);
    if (bits > 0 && bits < cus) cus = bits;
  } else {
    (void)hipGetLastError();  // a stream type without the query: the device's count stands
  }
  return cus;
}

// CUs a launch on `st` can occupy.  Cached per (device, stream handle) - the query costs microseconds of host time per
// launch - but a handle can be destroyed and handed out again with another CU mask: every entry is re-queried
// after 64 hits (never during a stream capture, where the query is not allowed), and when the table is full the oldest
// entry is replaced instead of every later stream paying the query per launch.  A stale count can only be wrong for
// 64 launches; the persistent kernels additionally never cut items unless the occupancy query says a workgroup fits.
int air_stream_cus(hipStream_t st) {
  struct Entry { int dev; hipStream_t st; int cus; int hits; };
  static std::mutex mu;
  static Entry cache[64];
  static int n = 0, next = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  {
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < n; ++i)
      if (cache[i].dev == dev && cache[i].st == st) {
        if (++cache[i].hits < 64) return cache[i].cus;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
          (void)hipGetLastError();
          return cache[i].cus;
        }
        cache[i].hits = 0;
        cache[i].cus = query_stream_cus(dev, st);
        return cache[i].cus;
      }
  }
  const int cus = query_stream_cus(dev, st);
  std::lock_guard<std::mutex> lk(mu);
  const int slot = n < 64 ? n++ : (next++ & 63);
  cache[slot] = Entry{dev, st, cus, 0};
  return cus;
}

int air_opt(AirOption o) {
  std::call_once(g_once, init_opts);
  return g_val[o].load(std::memory_order_relaxed);
}

extern "C" {
const char* air_version(void) { return "air_hip gfx950 1"; }
int air_abi_version(void) { return 1; }

int air_set_option(const char* name, int value) {
  const int i = find_opt(name);
  if (i < 0) return AIR_EINVAL;
  std::call_once(g_once, init_opts);
  g_val[i].store(value, std::memory_order_relaxed);
  return AIR_OK;
}
int air_get_option(const char* name, int* value) {
  const int i = find_opt(name);
  if (i < 0 || !value) return AIR_EINVAL;
  *value = air_opt((AirOption)i);
  return AIR_OK;
}
int air_stream_compute_units(air_stream_t stream) { return air_stream_cus(air_stream(stream)); }
int air_option_count(void) { return AIR_OPT_COUNT; }
const char* air_option_name(int index) { return index >= 0 && index < AIR_OPT_COUNT ? kOpts[index].name : nullptr; }
}
