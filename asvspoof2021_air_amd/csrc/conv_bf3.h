// Split-bf16 ("bf16 x 3", six products on v_mfma_f32_32x32x16_bf16) stride-2 3x3 forward convolution (conv_bf3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

bool air_bf3_s2_ok(int B, int Cin, int H, int W, int Cout);   // option CONV_S2 bit 4 and the shape constraints
// weight planes in fragment order; with_shortcut / w_sc / y_sc: the block's 1x1 / stride 2 shortcut rides along as a tenth
// tap and a second output of the same launch
size_t air_bf3_s2_packed_bytes(int Cout, int Cin, bool with_shortcut);
int air_bf3_s2_weights(const float* w, const float* w_sc, void* packed, int Cout, int Cin, hipStream_t st);
int air_bf3_s2_fwd(const float* x, const void* packed, float* y, float* y_sc, int B, int Cin, int H, int W, int Cout, int Ho,
                   int Wo, double flops, hipStream_t st);

// the data gradient of the same layer (+ the block's 1x1 / stride 2 shortcut's: dy_sc / w_sc, may be null) in one pass;
// option CONV_S2 bit 8
bool air_bf3_s2d_ok(int B, int Cin, int H, int W, int Cout);
size_t air_bf3_s2d_packed_bytes(int Cout, int Cin);
int air_bf3_s2d_weights(const float* w, const float* w_sc, void* packed, int Cout, int Cin, hipStream_t st);
int air_bf3_s2d_dgrad(const float* dy, const float* dy_sc, const void* packed, float* dx, const float* accumulate, int B,
                      int Cin, int H, int W, int Cout, int Ho, int Wo, double flops, hipStream_t st);

// the weight gradient of the same layer: partial sums per segment of output rows, [nseg][9][Cout][Cin] floats, to be added
// in order (reduce_partials_kernel); option CONV_S2 bit 16
bool air_bf3_s2w_ok(int B, int Cin, int H, int W, int Cout);
int air_bf3_s2w_nseg(int B, int Cin, int Ho, int Wo, int Cout);
int air_bf3_s2w_partials(const float* x, const float* dy, float* partial, int B, int Cin, int H, int W, int Cout, int Ho,
                         int Wo, double flops, hipStream_t st);
