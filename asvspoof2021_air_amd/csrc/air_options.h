// Dispatch options of the library: which of several equivalent kernels serves a call.  One table, one setter
// (air_set_option in include/air_hip.h); each option is seeded once from the environment variable AIR_<NAME>
// (the A/B switches of tools/ and the profiling scripts) and can be changed at run time, e.g. by the parity
// tests that run the same assertion on the direct and on the Winograd kernels.  Options never change results
// beyond the documented rounding of the kernel they select.
#pragma once

enum AirOption {
  AIR_OPT_NO_WINO4 = 0,        // 1: 3x3/s1 forward+dgrad skip Winograd F(4x4,3x3) (-> F(2x2,3x3) or direct)
  AIR_OPT_NO_WINOGRAD,         // bit 1: forward/dgrad on the direct f32 MFMA kernels; bit 2: weight gradients too
                               // (3 = the strict-parity configuration: every convolution an fmaf chain)
  AIR_OPT_WINO4_SPLIT,         // 0: never cut the k-step stream; 1: when the last round is > 13 % empty; 2: always
  AIR_OPT_WINO4_TH3,           // 1: F(3x4,3x3) tiles where they issue fewer positions than F(4x4,3x3); 2: on ties too
  AIR_OPT_WINO4_XCD,           // 0: items dealt over the whole chip; 1 / 2: inside an XCD (contiguous / strided quads)
  AIR_OPT_CONV_MT,             // force the direct kernels' pixel-tile count (0 = heuristic)
  AIR_OPT_WGRAD_WGS,           // workgroups of the split-K direct weight gradient
  AIR_OPT_WINO_WGRAD_WGS,      // workgroups of the Winograd weight gradient
  AIR_OPT_DIRECT_WGRAD_ROWS,   // 1: row-staged conv1 weight gradient
  AIR_OPT_C1B_PS,              // bit mask: persistent bf16 pointwise kernels (1 = ps<4>, 2 = ps2, 4 = dgrad too)
  AIR_OPT_C1B_GEMM_PS,         // 1: 256x256 persistent bf16 GEMM
  AIR_OPT_SKINNY_WGRAD,        // 1: streaming 16x16x4-MFMA weight gradient for the 16 -> 64 1x1 layer (0: the 64-channel-tile kernel)
  AIR_OPT_CONV_S2,             // bit 1: stride-2 forward in 4-channel (3x3) / 16-channel (1x1) K chunks (3 resident workgroups per CU, not 1);
                               // bit 2: stride-2 3x3 data gradient as one pass over dy (conv_s2_dgrad_kernel), not four class launches
                               // bit 4: stride-2 3x3 forward as six bf16 products per fp32 product (conv_bf3.hip)
                               // bit 8: the paired stride-2 data gradient likewise; bit 16: the stride-2 3x3 weight gradient
  AIR_OPT_WINO4_DEPHASE,       // every second persistent Winograd workgroup starts N x 4096 cycles late (0 = in phase)
  AIR_OPT_IR_FFT,              // 1: impulse responses of 128 .. 1025 taps are convolved by overlap-save FFT (augment.hip); 0: direct FIR
  AIR_OPT_TAP_ROWS,            // 1: the 64 -> 64 Res2 convs on bf16-resident rows stage 16-byte row pieces (round 6) also without a prologue (measured slower: default 0)
  AIR_OPT_COUNT
};

int air_opt(AirOption o);  // current value (thread-safe relaxed read)
