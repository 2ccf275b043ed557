// Opt-in per-kernel timing with HIP events on the launch stream (bench.py's
// roofline leg).  Off by default: zero cost on the product path.
#pragma once
#include <hip/hip_runtime.h>

enum AirKernelId {
  AIR_K_CONV_FWD_331 = 0,  // conv_fwd_kernel<3,3,1>  (forward 3x3 s1 AND every dgrad)
  AIR_K_CONV_FWD_332,      // conv_fwd_kernel<3,3,2>
  AIR_K_CONV_FWD_111,      // conv_fwd_kernel<1,1,1>
  AIR_K_CONV_FWD_112,      // conv_fwd_kernel<1,1,2>
  AIR_K_CONV_WG_331,       // conv_wgrad_kernel<3,3,1,64>
  AIR_K_CONV_WG_332,       // conv_wgrad_kernel<3,3,2,32>
  AIR_K_CONV_WG_111,       // conv_wgrad_kernel<1,1,1,64>
  AIR_K_CONV_WG_112,       // conv_wgrad_kernel<1,1,2,32>
  AIR_K_CONV_FWD_CLS,      // conv_fwd_kernel<1|2, 1|2, 1>: parity classes of stride-2 dgrad
  AIR_K_CONV_FWD_1D,       // conv_fwd_kernel<1, 3|5, 1>: ECAPA conv1d layers
  AIR_K_CONV_WG_1D,        // conv_wgrad_kernel<1, 3|5, 1>: ECAPA conv1d layers
  AIR_K_LFCC,              // lfcc_kernel
  AIR_K_CONV_WINO,         // wino_conv_kernel: 3x3 s1 forward and dgrad, Winograd F(2x2,3x3)
  AIR_K_CONV_WINO_WG,      // wino_wgrad_kernel: 3x3 s1 weight gradient, Winograd F(3x3,2x2)
  AIR_K_C1B_FWD,           // c1b_fwd_kernel / c1b_fwd_ps_kernel<WM> / c1b_fwd_ps2_kernel: bf16 pointwise conv1d forward / dgrad on fp32 tensors
  AIR_K_C1B_GEMM,          // c1b_gemm_kernel / c1b_gemm_ps_kernel: LDS-DMA bf16 GEMM (ECAPA layer4 forward / dgrad, every pointwise wgrad)
  AIR_K_C1B_TAP,           // c1b_tap_kernel: bf16 dilated K=3 Res2 convs, forward / dgrad (ECAPA)
  AIR_K_CONV_WINO4,        // wino4_conv_kernel: 3x3 s1 forward and dgrad, Winograd F(4x4,3x3)
  AIR_K_C1B_TAPW,          // c1b_tapw_kernel: bf16 weight gradient of the dilated K=3 Res2 convs, all branches of a block (ECAPA)
  AIR_K_CONV_WINO4_BN,     // wino4_conv_kernel launches whose epilogue also takes BatchNorm statistics / backward sums (round 4)
  AIR_K_CONV_S2_DGRAD,     // conv_s2_dgrad_kernel: 3x3 stride-2 data gradient (+ the 1x1 shortcut's), all parity classes in one pass
  AIR_K_CONV_S2_BF3,       // conv_s2_bf3_kernel: 3x3 stride-2 forward (+ 1x1 shortcut) as six bf16 products per fp32 product (round 5)
  AIR_K_CONV_S2D_BF3,      // conv_s2d_bf3_kernel: the paired stride-2 data gradient likewise
  AIR_K_CONV_S2W_BF3,      // conv_s2w_bf3_kernel: the stride-2 weight gradient likewise
  AIR_K_COUNT
};

bool air_prof_on();
// work = algorithmic FLOPs (or bytes); issued = FLOPs the kernel really sends to the matrix pipe (Winograd kernels:
// fewer than the algorithmic count, padded tiles included); < 0 = same as work.  bytes = algorithmic HBM bytes of
// the launch (operands read once + results written once), 0 = not stated
void air_prof_begin(int kid, double work, hipStream_t st, double issued = -1.0, double bytes = 0.0);
void air_prof_end(hipStream_t st);

struct AirProfScope {
  hipStream_t st;
  bool on;
  AirProfScope(int kid, double work, hipStream_t s, double issued = -1.0, double bytes = 0.0) : st(s), on(air_prof_on()) {
    if (on) air_prof_begin(kid, work, st, issued, bytes);
  }
  ~AirProfScope() {
    if (on) air_prof_end(st);
  }
};
