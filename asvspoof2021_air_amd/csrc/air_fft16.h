// 16-point complex FFT in registers on packed fp32 pairs (shared by the LFCC front-end, csrc/lfcc.hip, and the FFT form of the
// impulse-response convolution, csrc/augment.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// complex value as a register pair: sums, differences and the partial products of a complex multiply are
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on (re, im) - the same roundings as the scalar forms (the LFCC parity
// tests are unchanged), 512 floating-point VALU instructions per wave for 863 (round 4).  Measured: the launch stays
// at 31 us (B = 64): the kernel is NOT bound by its arithmetic - 2172 instructions per wave are ~4 us of issue at 4
// waves per SIMD - but by the latency chain of a workgroup (tables and PCM staged from L2 / HBM, two barriers, the
// strided feature-row stores) at two 70 KB workgroups per CU over 1.9 rounds.
typedef float cf __attribute__((ext_vector_type(2)));
// Swapped / negated halves are written as a shuffle and a product with (+-1, +-1) - exact, so every result keeps the
// scalar form's roundings - which hipcc folds into the op_sel / neg modifiers of ONE v_pk_fma_f32 / v_pk_mul_f32
// (left a bare negation of one half it builds the operand with v_mov + v_xor instead).
#define LFCC_SWAP(v) __builtin_shufflevector(v, v, 1, 0)
#define LFCC_LO(v) __builtin_shufflevector(v, v, 0, 0)
#define LFCC_HI(v) __builtin_shufflevector(v, v, 1, 1)
__device__ __forceinline__ cf add_mi(cf a, cf b) {  // a + (-i) b = (a.x + b.y, a.y - b.x)
  return __builtin_elementwise_fma(LFCC_SWAP(b), cf{1.0f, -1.0f}, a);
}
__device__ __forceinline__ cf add_pi(cf a, cf b) {  // a + (+i) b = (a.x - b.y, a.y + b.x)
  return __builtin_elementwise_fma(LFCC_SWAP(b), cf{-1.0f, 1.0f}, a);
}
__device__ __forceinline__ cf cmul(cf a, cf b) {  // (ar br - ai bi, ar bi + ai br): the scalar form's six roundings
  const cf t1 = LFCC_LO(a) * b;             // (ar br, ar bi)
  const cf t2 = LFCC_HI(a) * LFCC_SWAP(b);  // (ai bi, ai br)
  return __builtin_elementwise_fma(t2, cf{-1.0f, 1.0f}, t1);
}
__device__ __forceinline__ cf mul_mi(cf a) { return LFCC_SWAP(a) * cf{1.0f, -1.0f}; }  // * (-i) = (a.y, -a.x)
// a w16^2 = R2 (a.x + a.y, a.y - a.x);  a w16^6 = (R2 (a.y - a.x), -R2 (a.x + a.y))
__device__ __forceinline__ cf mul_w2(cf a, float r2) { return add_mi(a, a) * r2; }
__device__ __forceinline__ cf mul_w6(cf a, float r2) {
  const cf t = add_mi(a, a);
  return LFCC_SWAP(t) * cf{r2, -r2};
}

// forward 4-point DFT (w4 = -i), natural order in and out
__device__ __forceinline__ void fft4(cf& a0, cf& a1, cf& a2, cf& a3) {
  cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_mi(t1, t3);
  a3 = add_pi(t1, t3);
}

constexpr float C1 = 0.92387953251128674f;  // cos(pi/8)
constexpr float S1 = 0.38268343236508977f;  // sin(pi/8)
constexpr float R2 = 0.70710678118654752f;  // sqrt(1/2)

// y[m1*4+q2] *= w16^(m1*q2), w16 = e^{-2 pi i/16}
__device__ __forceinline__ void twiddle16(cf (&y)[16]) {
  y[5] = cmul(y[5], cf{C1, -S1});   // w^1
  y[6] = mul_w2(y[6], R2);          // w^2: R2 (re + im, im - re)
  y[7] = cmul(y[7], cf{S1, -C1});   // w^3
  y[9] = mul_w2(y[9], R2);          // w^2
  y[10] = mul_mi(y[10]);            // w^4
  y[11] = mul_w6(y[11], R2);        // w^6: R2 (im - re, -(re + im))
  y[13] = cmul(y[13], cf{S1, -C1}); // w^3
  y[14] = mul_w6(y[14], R2);        // w^6
  y[15] = cmul(y[15], cf{-C1, S1}); // w^9
}

// 16-point forward DFT, natural order in x[m] and out X[q].
// m = m1 + 4 m2, q = 4 q1 + q2.  PRUNED: x[10..15] are known zero (a frame has
// 160 complex samples = 10 per lane) so the first stage skips them.
template <bool PRUNED>
__device__ __forceinline__ void fft16(cf (&x)[16]) {
  cf y[16];
  if (PRUNED) {
#pragma unroll
    for (int m1 = 0; m1 < 2; ++m1) {  // inputs m1, m1+4, m1+8 (m1+12 == 0)
      cf a0 = x[m1], a1 = x[m1 + 4], a2 = x[m1 + 8];
      cf t0 = a0 + a2, t1 = a0 - a2;
      y[m1 * 4 + 0] = t0 + a1;
      y[m1 * 4 + 2] = t0 - a1;
      y[m1 * 4 + 1] = add_mi(t1, a1);
      y[m1 * 4 + 3] = add_pi(t1, a1);
    }
#pragma unroll
    for (int m1 = 2; m1 < 4; ++m1) {  // inputs m1, m1+4 only
      cf a0 = x[m1], a1 = x[m1 + 4];
      y[m1 * 4 + 0] = a0 + a1;
      y[m1 * 4 + 2] = a0 - a1;
      y[m1 * 4 + 1] = add_mi(a0, a1);
      y[m1 * 4 + 3] = add_pi(a0, a1);
    }
  } else {
#pragma unroll
    for (int m1 = 0; m1 < 4; ++m1) {
      cf a0 = x[m1], a1 = x[m1 + 4], a2 = x[m1 + 8], a3 = x[m1 + 12];
      fft4(a0, a1, a2, a3);
      y[m1 * 4 + 0] = a0;
      y[m1 * 4 + 1] = a1;
      y[m1 * 4 + 2] = a2;
      y[m1 * 4 + 3] = a3;
    }
  }
  twiddle16(y);
#pragma unroll
  for (int q2 = 0; q2 < 4; ++q2) {
    cf a0 = y[q2], a1 = y[4 + q2], a2 = y[8 + q2], a3 = y[12 + q2];
    fft4(a0, a1, a2, a3);
    x[q2] = a0;
    x[4 + q2] = a1;
    x[8 + q2] = a2;
    x[12 + q2] = a3;
  }
}

}  // namespace
