// Internal interface of the Winograd F(2x2,3x3) convolution kernels (conv_wino.hip), used by
// the C-ABI conv entry points in conv2d.hip for 3x3 / stride 1 / pad 1 layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// true when the Winograd path handles this 3x3 s1 p1 problem (M = output-role channels,
// Kc = input-role channels of the launch: swapped for a dgrad)
bool air_wino_ok(int B, int Kc, int H, int W, int M);

// floats of transformed, packed weights for (M, Kc)
size_t air_wino_packed_elems(int M, int Kc);

// y (B, M, H, W) = conv3x3_s1_p1(x (B, Kc, H, W), w) [+ residual].
// dgrad == 0: w is (M, Kc, 3, 3).  dgrad == 1: w is (Kc, M, 3, 3) and taps are flipped, i.e. the
// data gradient of the forward conv whose weight is w.  up: workspace of air_wino_packed_elems.
// w == nullptr: `up` already holds the transformed weights (air_wino_weights ran earlier, e.g. on another stream).
int air_wino_conv(const float* x, const float* w, float* y, const float* residual, int B, int Kc,
                  int H, int W, int M, int dgrad, float* up, double flops, hipStream_t st);
int air_wino_weights(const float* w, float* up, int M, int Kc, int dgrad, hipStream_t st);

// Weight gradient of the same convolution (Winograd F(3x3,2x2)).  Writes K-split partial sums
// partial[nsplit][9][Cout][Cin] (nsplit = air_wino_wgrad_nsplit) for reduce_partials_kernel.
bool air_wino_wgrad_ok(int B, int Cin, int H, int W, int Cout);
int air_wino_wgrad_nsplit(int B, int Cin, int H, int W, int Cout);
int air_wino_wgrad_partials(const float* x, const float* dy, float* partial, int B, int Cin, int H,
                            int W, int Cout, double flops, hipStream_t st);

// Winograd F(4x4,3x3) forward / data-gradient kernel (conv_wino4.hip): same contract as air_wino_conv with
// its own packing (36 transformed weights per (co, ci)); preferred over F(2x2,3x3) where it applies.
bool air_wino4_ok(int B, int Kc, int H, int W, int M);
size_t air_wino4_packed_elems(int M, int Kc);
// stats (forward only, 16-byte aligned, air_wino4_stats_bytes(B, H, W, M) bytes; nullptr = none): per-(channel,
// tile group) BatchNorm statistics records of y written by the epilogue, for bn_stats_from_records (norm_act.hip)
// bn (data-gradient launches with stats): {bnx, mean, invstd, gamma, beta} - the records become the two sums of the
// BatchNorm backward of relu(batchnorm(bnx)) whose output gradient this launch writes (W4Args::stats_mode 2)
int air_wino4_conv(const float* x, const float* w, float* y, const float* residual, int B, int Kc, int H,
                   int W, int M, int dgrad, float* up, double flops, hipStream_t st, float* stats = nullptr,
                   const float* const* bn = nullptr);
size_t air_wino4_stats_bytes(int B, int H, int W, int M);
// H: image height of the launch the weights are for (it picks the 4- or 3-row tile layout)
int air_wino4_weights(const float* w, float* up, int M, int Kc, int H, int dgrad, hipStream_t st, bool may_defer);
// defer(true): air_wino4_weights(may_defer = true) calls of this thread are recorded instead of launched; flush runs what was recorded as
// one launch (32 layers per launch); defer(false) drops the mode.  See air_conv2d_prepack_begin (conv2d.hip).
void air_wino4_weights_defer(bool on);
int air_wino4_weights_flush(hipStream_t st);
