/* air_hip.h -- C-ABI of the MI355X (gfx950) hot path of ASVspoof2021_AIR.
 *
 * The reference (yzyouzhang/ASVspoof2021_AIR) is pure Python/PyTorch and has no
 * FFI; the boundary it exposes for this path is the Python module surface
 * (SURVEY.md §8b).  This header is the C-ABI a binding for that surface calls:
 * each entry point names the reference function (file:line) whose device work
 * it replaces.  Conventions:
 *   - plain pointers and sizes only; device pointers unless suffixed _host
 *   - no allocation inside: callers pass workspaces (see *_ws_bytes queries)
 *   - every launch goes to the caller's stream (air_stream_t == hipStream_t)
 *   - return 0 (AIR_OK) or a negative AIR_E* code; never throws
 *   - no per-call global state; re-entrant across streams.  The only process-wide
 *     state is the dispatch-option table below (air_set_option): which of several
 *     equivalent kernels serves a call
 *   - tensors are contiguous fp32, NCHW / (B,C,T) / (B,T,D) as stated per call
 */
#ifndef AIR_HIP_H
#define AIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* air_stream_t; /* hipStream_t */

#define AIR_OK 0
#define AIR_EINVAL (-1)       /* bad argument (null pointer, non-positive size) */
#define AIR_EUNSUPPORTED (-2) /* valid request this build has no kernel for */
#define AIR_ELAUNCH (-3)      /* HIP reported a launch error */
#define AIR_EWORKSPACE (-4)   /* workspace too small */

/* Library identification: returns "air_hip gfx950 <abi-version>". */
const char* air_version(void);
int air_abi_version(void);

/* ------------------------------------------------------- dispatch options --
 * Process-wide switches between equivalent kernels (A/B measurement, and the
 * strict-parity configuration of the tests).  Each option NAME is seeded once
 * from the environment variable AIR_<NAME>; air_set_option changes it at run
 * time (takes effect at the next call; not meant to be flipped while another
 * thread is launching).  Unknown names return AIR_EINVAL.
 *   NO_WINO4 (0)          1: 3x3/stride-1 forward+dgrad skip Winograd F(4x4,3x3)
 *   NO_WINOGRAD (0)       bit 1: 3x3/stride-1 forward+dgrad on the direct f32-MFMA
 *                         kernels; bit 2: weight gradients too.  3 = every
 *                         convolution is an fmaf chain (round-1 tolerances hold)
 *   WINO4_SPLIT (1)       0 never / 1 when the last round is > 13 % empty / 2 always:
 *                         cut the Winograd k-step stream evenly over the workgroups
 *   WINO4_TH3 (2)         0: F(4x4,3x3) tiles always; 1: F(3x4,3x3) where it issues fewer
 *                         positions for the image height (H = 9, 5, 3); 2: also on ties (H = 18)
 *   WINO4_XCD (2)         0: work items dealt round-robin over the whole chip; 1: an XCD owns a
 *                         contiguous range of the item list; 2: an XCD owns the tile quads
 *                         congruent to its index (same L2 sharing as 1, measured faster)
 *   CONV_MT (0)           force the direct kernels' pixel-tile count (1 | 2)
 *   WGRAD_WGS (256), WINO_WGRAD_WGS (256)   workgroups of the split-K weight gradients
 *   DIRECT_WGRAD_ROWS (1) row-staged conv1 weight gradient
 *   C1B_PS (7)            bit mask of the persistent bf16 pointwise kernels
 *   C1B_GEMM_PS (1)       256x256 persistent bf16 GEMM
 *   SKINNY_WGRAD (1)      streaming weight gradient of the 16 -> 64 1x1 layer (0: generic 64-channel tiles)
 *   WINO4_DEPHASE (0)     every second persistent Winograd workgroup starts N x 4096 cycles late, so that the store
 *                         sections of neighbouring workgroups do not coincide (measured: profiles/r05_wino4_dephase.md)
 *   CONV_S2 (31)          bit 1: stride-2 forward in 4-channel (3x3) / 16-channel (1x1) K chunks (three resident
 *                         workgroups per CU instead of one; same arithmetic, chunk boundaries only); bit 2: stride-2
 *                         3x3 data gradient in one pass; bit 4: stride-2 3x3 forward on the bf16 matrix cores as six
 *                         bf16 products per fp32 product (operands split exactly into three bf16 planes: fp32-
 *                         equivalent arithmetic, csrc/conv_bf3.hip); bit 8: the same for the paired stride-2 data
 *                         gradient (air_conv2d_dgrad_s2_pair); bit 16: the stride-2 3x3 weight gradient likewise
 *   TAP_ROWS (0)          the 64 -> 64 Res2 convs on bf16-resident rows (air_h_conv1d_tap*) stage their operand as 16-byte
 *                         row pieces (8 frames of a channel, transposed into LDS) instead of 2-byte loads down the
 *                         channels; same sums in the same order (air_h_conv1d_tap_pro always does; measured -0.7 % of
 *                         the ECAPA step without a prologue: off)
 */
int air_set_option(const char* name, int value);
int air_get_option(const char* name, int* value_out);
int air_option_count(void);
const char* air_option_name(int index); /* NULL when out of range */

/* Compute units a launch on `stream` can occupy: the stream's CU mask (hipExtStreamCreateWithCUMask, or the
 * process-wide ROC_GLOBAL_CU_MASK) capped by the device's CU count.  The persistent kernels (one workgroup per
 * CU: the Winograd convolutions) size their grids by it, and wino4_conv_kernel cuts tail items between two
 * workgroups only when every workgroup of the launch is resident at once.  No reference counterpart
 * (main_train.py:101 pins one whole GPU through CUDA_VISIBLE_DEVICES). */
int air_stream_compute_units(air_stream_t stream);

/* ------------------------------------------------------------------ LFCC --
 * Replaces LFCC.forward (feature_extraction.py:93-138) incl. delta (:41-58),
 * with the filterbank of LFCC.__init__ (:77-86) and the LinearDCT weight
 * (utils_dsp.py:233-244) folded into a small "plan" blob built on the host.
 */
size_t air_lfcc_plan_bytes(void);
/* fb_host: [nbin][nfilt] row-major (module.lfcc_fb); dct_host: [nfilt][nfilt]
 * (module.l_dct.weight, y = x @ W^T); window_host: [fl] analysis window
 * (torch.hamming_window(fl), :110) or NULL for the periodic Hamming closed form.
 * fl/fs/fn = window / hop / FFT length (this build: 320/160/512, nfilt <= 32).
 * Writes air_lfcc_plan_bytes() bytes to plan_host_out; the caller uploads it. */
int air_lfcc_plan_build(const float* fb_host, int nbin, int nfilt, const float* dct_host,
                        const float* window_host, int fl, int fs, int fn, void* plan_host_out);
#define AIR_LFCC_EMPHASIS 1 /* pre-emphasis 0.97 (feature_extraction.py:105-106) */
#define AIR_LFCC_DELTA 2    /* append delta and delta-delta (:130-133) */
/* pcm: (B, L) fp32, NOT modified.  out: (B, 1 + L/fs, nfilt * (3 if DELTA else 1)). */
int air_lfcc_fwd(const float* pcm, int B, int L, float* out, const void* plan_dev, int flags,
                 air_stream_t stream);
/* Fused variant writing the model-input layout the trainer builds on the host
 * (dataset.py:66-79 pad/chop + main_train.py:338 transpose): out is
 * (B, D, feat_len) with frame t' taken from frame (start[b] + t') mod T when
 * T < feat_len ("repeat" padding, dataset.py:519-522) or start[b] + t' when
 * T >= feat_len (chop; start_dev may be NULL = 0).  */
int air_lfcc_fwd_padded(const float* pcm, int B, int L, float* out, int feat_len,
                        const int* start_dev, const void* plan_dev, int flags, air_stream_t stream);
/* All three pad modes of the reference's --padding flag (main_train.py:45, dataset.py:72-79) for inputs
 * shorter than feat_len; longer inputs are chopped at start[b] (clamped to [0, T - feat_len]) whatever the mode:
 *   AIR_PAD_REPEAT  tile the frames (dataset.py:519-522)
 *   AIR_PAD_ZERO    append zero frames (dataset.py:513-517)
 *   AIR_PAD_SILENCE PREPEND silence_dev (D floats: the LFCC frame of digital silence, dataset.py:13-16, :524-528)
 * Exactly one of pcm (fp32) / pcm16 (16-bit PCM) is non-NULL.  Any other pad_mode: AIR_EINVAL (the reference
 * raises ValueError, dataset.py:79). */
#define AIR_PAD_REPEAT 0
#define AIR_PAD_ZERO 1
#define AIR_PAD_SILENCE 2
int air_lfcc_fwd_padded_ex(const float* pcm, const int16_t* pcm16, int B, int L, float* out, int feat_len,
                           const int* start_dev, const void* plan_dev, int flags, int pad_mode,
                           const float* silence_dev, air_stream_t stream);
/* Same from 16-bit PCM as stored in the corpus' wav/flac files: the kernel converts s -> s / 32768 (exact, what
 * soundfile/librosa hand the reference, preprocess.py:239-241), so the features are bit-identical to the fp32
 * entry points on the converted samples while the kernel reads half the bytes (128 KB instead of 256 KB per
 * 4 s utterance) and the caller uploads half as much.  feat_len <= 0 selects the (B, T, D) layout. */
int air_lfcc_fwd_padded_i16(const int16_t* pcm16, int B, int L, float* out, int feat_len,
                            const int* start_dev, const void* plan_dev, int flags, air_stream_t stream);
/* The reference mutates its input (feature_extraction.py:106).  This applies
 * the same in-place x[n] -= coef*x[n-1] (n>=1).  ws: >= air_preemph_ws_bytes. */
size_t air_preemph_ws_bytes(int B, int L);
int air_preemph_inplace(float* pcm, int B, int L, float coef, void* ws, size_t ws_bytes,
                        air_stream_t stream);
/* (B,T,D) -> (B,D,feat_len) repeat-pad / chop + transpose (dataset.py:66-79, main_train.py:338). */
int air_pad_transpose(const float* feat, int B, int T, int D, float* out, int feat_len,
                      const int* start_dev, air_stream_t stream);
/* ... with the pad mode of air_lfcc_fwd_padded_ex (dataset.py:513-528). */
int air_pad_transpose_ex(const float* feat, int B, int T, int D, float* out, int feat_len,
                         const int* start_dev, int pad_mode, const float* silence_dev, air_stream_t stream);

/* --------------------------------------------------------------- conv2d --
 * Replaces nn.Conv2d forward/backward as used by resnet.py:131,56-61,140
 * (bias=False everywhere).  NCHW fp32.  Implicit GEMM on f32 MFMA.
 * Optional fused prologue: per-input-channel y = max(0, x*scale[c]+shift[c])
 * (BatchNorm apply + ReLU of the pre-activation block, resnet.py:64,67)
 * applied while staging the input; optional epilogue residual add
 * (resnet.py:68).  3x3 / stride 1 / pad 1 layers without the fused prologue run
 * as Winograd F(4x4,3x3) (forward, dgrad: csrc/conv_wino4.hip) and F(3x3,2x2)
 * (wgrad: csrc/conv_wino.hip); results within 2e-5 of the output scale of the
 * fp64 convolution (direct kernels: 3e-6).
 */
typedef struct AirConv2d {
  int B, Cin, H, W;       /* input */
  int Cout, KH, KW;       /* filter */
  int sh, sw, ph, pw;     /* stride, zero padding */
  int Ho, Wo;             /* output (must equal the conv arithmetic) */
} AirConv2d;

size_t air_conv2d_ws_bytes(const AirConv2d* p);
/* y = conv(act(x), w) [+ residual].  in_scale/in_shift NULL = identity prologue.
 * relu: apply max(0,.) after the affine prologue.
 * stats: NULL, or a 16-byte aligned buffer of air_conv2d_fwd_stats_bytes(p) bytes (0 = this layer has no fused
 * statistics under the current dispatch options: pass NULL) that receives BatchNorm statistics of y (of y + residual)
 * from the convolution's epilogue: one record {n, K, sum(y - K), sum((y - K)^2)} per (channel, tile group) of the
 * Winograd F(3x4 / 4x4, 3x3) kernel - the (count, mean, M2) of the group's outputs in shifted form.  Hand the buffer
 * to air_bn_stats as stats_in and the BatchNorm that follows the convolution (resnet.py:63-69: conv1 -> bn2, block
 * output -> the next block's bn1) merges the records in fp64 instead of reading y again: 12 of the 18 statistics
 * passes of a ResNet-18 step.  Round 4, measured (rounds 2 and 3 had priced it and not built it): see DESIGN.md. */
size_t air_conv2d_fwd_stats_bytes(const AirConv2d* p);
int air_conv2d_fwd(const AirConv2d* p, const float* x, const float* w, float* y,
                   const float* in_scale, const float* in_shift, int relu,
                   const float* residual, double* stats, void* ws, size_t ws_bytes,
                   air_stream_t stream);
/* dx = conv_transpose(dy, w) [+ accumulate]: gradient w.r.t. the (activated) conv
 * input.  accumulate: NULL or a tensor shaped like dx that is added (may alias dx;
 * used to join the 1x1-shortcut and 3x3 branches of a PreActBlock). */
int air_conv2d_dgrad(const AirConv2d* p, const float* dy, const float* w, float* dx,
                     const float* accumulate, void* ws, size_t ws_bytes, air_stream_t stream);
/* Weight transforms off the critical path.  The 3x3 / stride 1 / pad 1 layers run as Winograd kernels whose
 * transformed weights (G g G^T, packed) are otherwise produced by a small kernel in front of EVERY forward and dgrad
 * launch (25 + 25 launches of ~9 us per ResNet-18 step, serialised with the convs).  They depend on the weights
 * only: air_conv2d_prepack writes them for `pass` (0 forward, 1 dgrad) into a caller-owned buffer of
 * air_conv2d_prepack_bytes(p, pass) bytes (0 = this layer / pass has no such form) - on any stream, e.g. a side
 * stream at the start of the step - and the _pre entry points take that buffer as w_packed (NULL = transform here).
 * Round 3: the direct kernels' weight slabs (stride-2 3x3, 1x1, conv5, the parity classes of a stride-2 data
 * gradient: 28 more ~5 us launches per step) travel the same way: air_conv2d_prepack writes every slab the pass
 * consumes, in consumption order (one code path walks them for sizing, packing and use).  A buffer must be consumed
 * under the dispatch options it was produced under. */
size_t air_conv2d_prepack_bytes(const AirConv2d* p, int pass);
/* The layout air_conv2d_prepack writes for (p, pass) under the current dispatch options: a binding records it beside the
 * buffer and compares at use (options can change in between; the layouts differ in element type, not only in size). */
enum { AIR_PACK_NONE = 0, AIR_PACK_F32_SLABS = 1, AIR_PACK_WINO2 = 2, AIR_PACK_BF3 = 3, AIR_PACK_WINO4 = 4 };
int air_conv2d_prepack_layout(const AirConv2d* p, int pass);
int air_conv2d_prepack(const AirConv2d* p, const float* w, int pass, void* out, size_t out_bytes,
                       air_stream_t stream);
/* Between _begin and _flush the air_conv2d_prepack / air_conv2d_dgrad_s2_pair_prepack calls of the calling thread only
 * RECORD their transforms; _flush runs them on `stream` as one launch per 32 Winograd transforms and one per 32 direct
 * slabs (a ResNet-18 step: 3 launches for 47).  The output buffers are valid behind _flush in stream order. */
int air_conv2d_prepack_begin(void);
int air_conv2d_prepack_flush(air_stream_t stream);
int air_conv2d_fwd_pre(const AirConv2d* p, const float* x, const float* w, const void* w_packed, float* y,
                       const float* in_scale, const float* in_shift, int relu, const float* residual,
                       double* stats, void* ws, size_t ws_bytes, air_stream_t stream);
int air_conv2d_dgrad_pre(const AirConv2d* p, const float* dy, const float* w, const void* w_packed, float* dx,
                         const float* accumulate, void* ws, size_t ws_bytes, air_stream_t stream);
/* The same data gradient when dx is the gradient with respect to the OUTPUT of relu(batchnorm(bn_x)) - the activated
 * tensor a 3x3 / stride 1 convolution of a PreActBlock reads (resnet.py:63-69: bn2 -> relu -> conv2, and bn1 -> relu
 * -> conv1 in the blocks without a 1x1 shortcut).  The epilogue that writes dx also takes the two per-channel sums the
 * BatchNorm backward needs - sum g and sum g * xhat with g = dx where bn_x * scale + shift > 0, xhat = (bn_x - mean) *
 * invstd, the arithmetic of air_bn_bwd's own first pass - into `sums` (air_conv2d_dgrad_bn_sums_bytes(p) bytes,
 * 16-byte aligned; 0 = this layer has no such form under the current dispatch options: call air_conv2d_dgrad_pre).
 * Hand `sums` to air_bn_bwd_ex3 and the BatchNorm backward skips the pass that re-reads dx and bn_x (12 of the 18
 * BatchNorm backward passes of a ResNet-18 step).  bn_x has dx's shape; the four vectors have Cin entries. */
size_t air_conv2d_dgrad_bn_sums_bytes(const AirConv2d* p);
int air_conv2d_dgrad_bn(const AirConv2d* p, const float* dy, const float* w, const void* w_packed, float* dx,
                        const float* accumulate, const float* bn_x, const float* bn_mean, const float* bn_invstd,
                        const float* bn_gamma, const float* bn_beta, void* sums, void* ws, size_t ws_bytes,
                        air_stream_t stream);
/* Forward of the two convolutions a stride-2 PreActBlock applies to the SAME activated input (resnet.py:56 conv1 3x3 /
 * stride 2 / pad 1 and :61-66 the 1x1 / stride 2 shortcut, both reading `out` of :64) in one launch (round 5, split-bf16
 * kernel of csrc/conv_bf3.hip: the shortcut's operand is the centre tap's fragment, already loaded and split): y =
 * conv1(x), y_sc = shortcut(x).  p describes the 3x3 layer.  _prepack_bytes 0 / AIR_EUNSUPPORTED: not this shape, or
 * option CONV_S2 bit 4 off - a binding then makes the two air_conv2d_fwd calls.  packed: the buffer _prepack wrote under
 * the same options, or NULL (then ws, >= _prepack_bytes, receives the planes in front of the launch). */
size_t air_conv2d_fwd_s2_pair_prepack_bytes(const AirConv2d* p);
int air_conv2d_fwd_s2_pair_prepack(const AirConv2d* p, const float* w, const float* w_sc, void* out, size_t out_bytes,
                                   air_stream_t stream);
int air_conv2d_fwd_s2_pair(const AirConv2d* p, const float* x, const float* w, const float* w_sc, const void* packed,
                           float* y, float* y_sc, void* ws, size_t ws_bytes, air_stream_t stream);

/* The data gradient of a PreActBlock's stride-2 pair in one pass (resnet.py:56-66: conv1 3x3 / stride 2 / pad 1 and
 * the 1x1 / stride 2 shortcut read the same activated tensor): dx = conv_transpose(dy, w) + conv_transpose(dy_sc, w_sc)
 * [+ accumulate].  p describes the 3x3 convolution; the shortcut has its Cin, Cout, input and output shape.  A wave
 * keeps all four parity classes of its dx tile in registers, so dy is read once and dx leaves as whole rows (rounds 1-3:
 * four class launches + a read-modify-write launch for the shortcut).  `packed`: NULL (the weights are packed into ws,
 * air_conv2d_dgrad_s2_pair_prepack_bytes(p) bytes, in front of the launch) or the buffer air_conv2d_dgrad_s2_pair_prepack
 * wrote under the same dispatch options.  _prepack_bytes 0 / AIR_EUNSUPPORTED: not this shape, or option CONV_S2 bit 2
 * is off - call air_conv2d_dgrad twice.  air_conv2d_dgrad itself takes the same kernel for a lone 3x3 / stride 2. */
size_t air_conv2d_dgrad_s2_pair_prepack_bytes(const AirConv2d* p);
int air_conv2d_dgrad_s2_pair_prepack(const AirConv2d* p, const float* w, const float* w_sc, void* out, size_t out_bytes,
                                     air_stream_t stream);
int air_conv2d_dgrad_s2_pair(const AirConv2d* p, const float* dy, const float* w, const float* dy_sc, const float* w_sc,
                             const void* packed, float* dx, const float* accumulate, void* ws, size_t ws_bytes,
                             air_stream_t stream);
/* dw = correlation(act(x), dy); same prologue as fwd so the activated tensor
 * is never materialised. */
int air_conv2d_wgrad(const AirConv2d* p, const float* x, const float* dy, float* dw,
                     const float* in_scale, const float* in_shift, int relu,
                     void* ws, size_t ws_bytes, air_stream_t stream);

/* ------------------------------------------- adversarial channel head ----
 * model.ChannelClassifier (model.py:976-1023: GRL -> Linear -> Dropout(0.3) -> ReLU -> Linear ->
 * ReLU) and nn.CrossEntropyLoss (main_train.py:251) of the --ADV_AUG branch
 * (main_train.py:377-403, :420-453).  The two Linear layers use air_linear_fwd / air_linear_bwd. */
/* keep[i] = (u_i >= p) / (1 - p), u from Philox4x32-10(seed, offset + i/4): nn.Dropout's scaled mask. */
int air_dropout_mask(float* keep, size_t n, float p, uint64_t seed, uint64_t offset, air_stream_t stream);
/* y = relu(x * keep); keep NULL = eval mode. */
int air_mask_relu_fwd(const float* x, const float* keep, size_t n, float* y, air_stream_t stream);
/* dx = alpha * dy * keep * (y > 0). */
int air_mask_relu_bwd(const float* dy, const float* y, const float* keep, size_t n, float alpha, float* dx,
                      air_stream_t stream);
/* x *= alpha (gradient reversal: alpha = -lambda, model.py:990-995). */
int air_scale(float* x, size_t n, float alpha, air_stream_t stream);
/* probs = softmax(logits (B,C)); loss = mean_b -log probs[b][labels[b]]; *correct = #(argmax == label). */
int air_softmax_ce_fwd(const float* logits, const long long* labels, int B, int C, float* probs, float* loss,
                       int* correct_or_null, air_stream_t stream);
/* dlogits = g * (probs - onehot(labels)) / B, g = *gscale_or_null (device scalar) or 1. */
int air_softmax_ce_bwd(const float* probs, const long long* labels, int B, int C, const float* gscale_or_null,
                       float* dlogits, air_stream_t stream);

/* ------------------------------------------------- channel augmentation --
 * On-the-fly IR convolution ahead of the LFCC kernel (BASELINE.json configs[4]).  Replaces the
 * OFFLINE augmentation of channel_simulation/simulated_device.py:16-61 and
 * simulated_device_channel.py:6-56, which shell out to the un-vendored idiap/acoustic-simulator
 * tool (PARITY UNPINNED: spec = oracle/channel.py, checked against scipy.signal.fftconvolve).
 * y[b] = (x[b] * irs[ir_idx[b]])[:L]; ir_idx[b] < 0 copies the utterance unchanged; ir_idx NULL
 * uses IR 0 for all.  normalize != 0 rescales each augmented utterance to its input peak
 * (max|y| = max|x|).  irs is (n_ir, H) fp32, rows zero-padded to H taps.  x and y must not alias. */
size_t air_ir_convolve_ws_bytes(int B);
/* Round 5: with a workspace of air_ir_convolve_ws_bytes_ex(B, n_ir, H) bytes (>= the above), impulse responses of
 * 128 .. 1025 taps are convolved by overlap-save FFT (option IR_FFT, default 1; same result to fp32 FFT rounding, ~10 x
 * faster at 1024 taps); with the smaller workspace, or any other H, the direct FIR runs. */
size_t air_ir_convolve_ws_bytes_ex(int B, int n_ir, int H);
int air_ir_convolve(const float* x, int B, int L, const float* irs, int n_ir, int H, const int* ir_idx,
                    int normalize, float* y, void* ws, size_t ws_bytes, air_stream_t stream);

/* --------------------------------------------------------------- conv1d --
 * nn.Conv1d (stride 1) as used by ecapa_tdnn.py:39,46,55,111,118,140,143, with the
 * conv -> ReLU -> BN ordering of ecapa_tdnn.py:67-69 supported by bias / ReLU epilogues.
 * x (B, Cin, T), w (Cout, Cin, K), y (B, Cout, T).  Supported: K=1; K=3 with
 * dilation 2..4 and pad = dilation; K=5 with pad 2.  x_bstride / y_bstride are batch
 * strides in floats (0 = contiguous) so channel groups of a wider tensor (the Res2
 * split, the (x1,x2,x3) concat) are addressed in place.
 */
typedef struct AirConv1d {
  int B, Cin, T, Cout, K, dil, pad;
  size_t x_bstride, y_bstride;
} AirConv1d;

size_t air_conv1d_ws_bytes(const AirConv1d* p);
/* y = relu?(conv1d(x, w) + bias[co] + bias_bc[b][co]); bias, bias_bc may be NULL. */
int air_conv1d_fwd(const AirConv1d* p, const float* x, const float* w, const float* bias,
                   const float* bias_bc, int relu, float* y, void* ws, size_t ws_bytes,
                   air_stream_t stream);
int air_conv1d_dgrad(const AirConv1d* p, const float* dy, const float* w, float* dx,
                     const float* accumulate, void* ws, size_t ws_bytes, air_stream_t stream);
int air_conv1d_wgrad(const AirConv1d* p, const float* x, const float* dy, float* dw, void* ws,
                     size_t ws_bytes, air_stream_t stream);

/* bf16-compute variants of the pointwise (K = 1) layers (BASELINE.json configs[2], "ECAPA-TDNN-512
 * bf16 train"): ecapa_tdnn.py:39,55,118,140,143.  Same tensors (fp32 in HBM), same epilogues;
 * both operands are rounded to bf16 (nearest even) as they are staged, products accumulate in
 * fp32 on v_mfma_f32_32x32x16_bf16 - the arithmetic of torch.autocast(bfloat16) for nn.Conv1d.
 * air_conv1d_bf16_supported(p, pass): pass 0 forward (Cout % 128 == 0, Cin % 32 == 0),
 * 1 dgrad (Cin % 128 == 0, Cout % 32 == 0), 2 wgrad (Cout % 128 == 0, Cin % 128 == 0); 1 = yes.
 * Other layers (K = 3 dilated, K = 5) stay on the fp32 entry points above. */
int air_conv1d_bf16_supported(const AirConv1d* p, int pass);
size_t air_conv1d_bf16_ws_bytes(const AirConv1d* p);
int air_conv1d_fwd_bf16(const AirConv1d* p, const float* x, const float* w, const float* bias,
                        const float* bias_bc, int relu, float* y, void* ws, size_t ws_bytes,
                        air_stream_t stream);
/* Same, and y is also written as bf16 (nearest even) into y_bf16[(b*Cout + c) * air_conv1d_bf16_tp(T) + t], the
 * operand layout of air_conv1d_wgrad_bf16_pre (from the GEMM's epilogue for the wide layers, by a conversion
 * pass otherwise); NULL = air_conv1d_fwd_bf16. */
int air_conv1d_fwd_bf16_ex(const AirConv1d* p, const float* x, const float* w, const unsigned short* w_packed,
                           const float* bias, const float* bias_bc, int relu, float* y, unsigned short* y_bf16,
                           void* ws, size_t ws_bytes, air_stream_t stream);
/* K = 3 layers: the kernels read the weights as bf16 [tap][m][k] (dgrad: transposed, taps flipped).  The
 * _bf16 entry points pack them on every call; air_conv1d_tap_pack_bf16 packs n_layers equally shaped layers
 * (weights w_stride floats apart, e.g. the seven Res2 branch convs of a Bottle2neck inside the parameter arena)
 * in ONE launch, layer i at out + i * air_conv1d_tap_pack_elems(Cout, Cin); pass that block as w_packed (w may
 * then be NULL).  w_packed must be NULL for K = 1 layers. */
size_t air_conv1d_tap_pack_elems(int Cout, int Cin);
int air_conv1d_tap_pack_bf16(const float* w, size_t w_stride, int n_layers, int Cout, int Cin, int transpose,
                             unsigned short* out, air_stream_t stream);
int air_conv1d_dgrad_bf16(const AirConv1d* p, const float* dy, const float* w, float* dx,
                          const float* accumulate, void* ws, size_t ws_bytes, air_stream_t stream);
/* Forward (dgrad = 0: y = W . x) or data gradient (dgrad = 1: dx = W^T . dy + accumulate) of a K = 1 layer whose INPUT
 * operand the caller already holds as a bf16 copy in the weight-gradient layout [b][channel][Tp] (xb, batch stride
 * xb_bstride in elements, 0 = dense): the 256 x 256 GEMM reads it K-major (ds_read_b64_tr_b16), so neither a
 * conversion nor a transposed copy is made - the SAME copy serves this layer's forward, its data gradient's
 * counterpart and the weight gradients around it.  Needs Cout % 256 == 0 (forward) / Cin % 256 == 0 (dgrad), the K
 * dimension a multiple of 64 and Tp % 256 == 0; AIR_EUNSUPPORTED otherwise.  accumulate / accumulate2 (either may be
 * NULL) with their batch strides in floats (0 = the output's), as in air_conv1d_dgrad_bf16_ex; y_bf16 as in
 * air_conv1d_fwd_bf16_ex. */
int air_conv1d_pointwise_bf16_kmajor(const AirConv1d* p, const unsigned short* xb, size_t xb_bstride, const float* w,
                                     int dgrad, const float* bias, const float* bias_bc, int relu,
                                     const float* accumulate, size_t acc_bstride, const float* accumulate2,
                                     size_t acc2_bstride, float* y, unsigned short* y_bf16, void* ws, size_t ws_bytes,
                                     air_stream_t stream);
/* Same with up to two accumulate operands, each with its own batch stride in floats (0 = dx's): dx = dgrad +
 * accumulate + accumulate2.  ECAPA's block input gradient = dgrad(conv1) + d(block output) [the residual,
 * ecapa_tdnn.py:93] + the (B, 1536, T) concat gradient's slice for the previous block [:170] in one epilogue.
 * AIR_EUNSUPPORTED for the K = 3 and the wide-layer paths when a second operand or a foreign stride is given. */
int air_conv1d_dgrad_bf16_ex(const AirConv1d* p, const float* dy, const float* w, const unsigned short* w_packed,
                             float* dx, const float* accumulate, size_t acc_bstride, const float* accumulate2,
                             size_t acc2_bstride, void* ws, size_t ws_bytes, air_stream_t stream);
int air_conv1d_wgrad_bf16(const AirConv1d* p, const float* x, const float* dy, float* dw, void* ws,
                          size_t ws_bytes, air_stream_t stream);
/* The weight-gradient GEMM runs on bf16 copies of its operands, [b][channel][Tp] with
 * Tp = air_conv1d_bf16_tp(T) frames per row and zeros for t >= T; air_conv1d_wgrad_bf16 makes them in its
 * workspace, one HBM pass per operand.  A caller that already holds a copy - written by the tensor's producer
 * (air_bn_bwd_ex2's dx_bf16) or converted once with air_conv1d_cvt_bf16 for several layers (ECAPA's (B, 1536, T)
 * concat feeds layer4 and, as channel slices, two Bottle2neck conv1 layers: ecapa_tdnn.py:118,39) - passes it to
 * air_conv1d_wgrad_bf16_pre: x_bf16 / dy_bf16 with their batch strides in ELEMENTS (0 = dense; a channel slice
 * of a wider copy keeps the wide stride), NULL = convert the fp32 tensor as before.  Same rounding (nearest
 * even) everywhere, so results are bit-identical to air_conv1d_wgrad_bf16. */
int air_conv1d_bf16_tp(int T);
int air_conv1d_cvt_bf16(const float* x, size_t x_bstride, int B, int C, int T, unsigned short* out,
                        air_stream_t stream);
int air_conv1d_wgrad_bf16_pre(const AirConv1d* p, const float* x, const float* dy, const unsigned short* x_bf16,
                              size_t x_bf16_bstride, const unsigned short* dy_bf16, size_t dy_bf16_bstride,
                              float* dw, void* ws, size_t ws_bytes, air_stream_t stream);

/* ------------------------------------------------------- batchnorm/relu --
 * nn.BatchNorm2d/1d (+ F.relu) as used at resnet.py:55-67,132,142 and
 * ecapa_tdnn.py.  x is (B, C, S) with S = H*W (or T).
 */
/* Batch statistics -> mean/invstd, fused scale/shift for the apply, running
 * stat update (momentum 0.1, unbiased var) when running_* non-NULL.
 * stats_in: NULL - the kernel reduces x itself (shifted sums, fp64 two-stage, fixed order) - or the statistics
 * records air_conv2d_fwd wrote for this very tensor (its header carries the group and channel counts, which must
 * match C; x is then not read): a wave per channel merges them in fp64 in a fixed order (Chan's formula on
 * (n, mean, M2)). */
size_t air_bn_ws_bytes(int B, int C, int S);
int air_bn_stats(const float* x, int B, int C, int S, const double* stats_in,
                 const float* gamma, const float* beta, float eps, float momentum,
                 float* running_mean, float* running_var,
                 float* mean, float* invstd, float* scale, float* shift,
                 void* ws, size_t ws_bytes, air_stream_t stream);
/* Eval-mode scale/shift from running stats. */
int air_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* scale, float* shift,
                       air_stream_t stream);
/* y = x*scale[c] + shift[c], optional ReLU. */
int air_bn_apply(const float* x, int B, int C, int S, const float* scale, const float* shift,
                 int relu, float* y, air_stream_t stream);
/* Same, and rowmean[b*C + c] (B*C floats, may be NULL) receives the mean over s of the OUTPUT plane: the SE
 * squeeze (ecapa_tdnn.py:19) of the tensor being written, without a pass that re-reads it.  rowmean needs
 * 32 <= S <= 1024 (one workgroup per plane); AIR_EUNSUPPORTED otherwise.  y_bf16 (may be NULL): y also as bf16 at
 * y_bf16[(b*C + c) * y_bf16_tp + s], the operand layout of air_conv1d_wgrad_bf16_pre. */
int air_bn_apply_ex(const float* x, int B, int C, int S, const float* scale, const float* shift,
                    int relu, float* y, float* rowmean, unsigned short* y_bf16, int y_bf16_tp, air_stream_t stream);
/* Backward of y = relu?(bn(x)) in training mode.  dy: grad wrt y.  relu bit 0: a ReLU
 * follows the BN (resnet.py:64); bit 1: the BN input is itself a ReLU output
 * (conv -> ReLU -> BN, ecapa_tdnn.py:67-69), so dx is masked where x == 0 and is the
 * gradient w.r.t. the pre-ReLU conv output.
 * dx_accum: if non-zero dx += result (gradient joining a residual branch).
 * dgamma/dbeta: (C,) written. */
int air_bn_bwd(const float* x, const float* dy, int B, int C, int S,
               const float* mean, const float* invstd, const float* gamma, const float* beta,
               int relu, float* dx, int dx_accum, float* dgamma, float* dbeta,
               void* ws, size_t ws_bytes, air_stream_t stream);
/* Same with three fusions for the conv -> ReLU -> BN layers of ecapa_tdnn.py:
 *  - dy may be a channel-slice view (dy_bstride = batch stride in floats, 0 = dense) and a second gradient dy2
 *    (same shape, own batch stride, may be NULL) is added to it on the fly: the Res2 chain's
 *    "d(sp_i) = d(cat slice) + d(next branch input)" join (ecapa_tdnn.py:78-83) without a pass that forms the sum;
 *  - dy_rowbias (B*C) or NULL: the incoming gradient is dy + rowbias_scale * dy_rowbias[b][c] - the SE
 *    squeeze's mean-over-time gradient (ecapa_tdnn.py:19: d mean / d x = 1/T) without a pass that adds it;
 *  - dbias (C) or NULL (needs relu bit 1): gradient of the conv bias, sum_{b,s} dx, obtained in closed
 *    form from three extra sums of the statistics pass - no pass over dx. */
int air_bn_bwd_ex(const float* x, const float* dy, size_t dy_bstride, const float* dy2, size_t dy2_bstride,
                  const float* dy_rowbias, float rowbias_scale, int B, int C, int S, const float* mean,
                  const float* invstd, const float* gamma, const float* beta, int relu, float* dx, int dx_accum,
                  float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes, air_stream_t stream);
/* Same, and dx is also written as bf16 (nearest even) into dx_bf16[(b*C + c) * dx_bf16_tp + s] - the operand
 * layout of air_conv1d_wgrad_bf16_pre (dx_bf16_tp = air_conv1d_bf16_tp(S); the caller keeps the frames
 * s >= S zero).  dx_bf16 NULL = air_bn_bwd_ex. */
int air_bn_bwd_ex2(const float* x, const float* dy, size_t dy_bstride, const float* dy2, size_t dy2_bstride,
                   const float* dy_rowbias, float rowbias_scale, int B, int C, int S, const float* mean,
                   const float* invstd, const float* gamma, const float* beta, int relu, float* dx, int dx_accum,
                   float* dgamma, float* dbeta, float* dbias, unsigned short* dx_bf16, int dx_bf16_tp, void* ws,
                   size_t ws_bytes, air_stream_t stream);
/* air_bn_bwd_ex2 with the two per-channel sums taken by the convolution that produced dy (air_conv2d_dgrad_bn,
 * `sums`): the first pass over (x, dy) is skipped, a workgroup per channel merges the records in fp64.  sums_in NULL =
 * air_bn_bwd_ex2.  Only for plain relu(batchnorm(x)) with one dense gradient (dy2, dy_rowbias, dbias NULL, relu = 1). */
int air_bn_bwd_ex3(const float* x, const float* dy, size_t dy_bstride, const float* dy2, size_t dy2_bstride,
                   const float* dy_rowbias, float rowbias_scale, int B, int C, int S, const float* mean,
                   const float* invstd, const float* gamma, const float* beta, int relu, float* dx, int dx_accum,
                   float* dgamma, float* dbeta, float* dbias, unsigned short* dx_bf16, int dx_bf16_tp,
                   const void* sums_in, void* ws, size_t ws_bytes, air_stream_t stream);

/* ------------------------------------------------------------- pooling ---
 * SelfAttention.forward (resnet.py:23-46) on x (B, C, T) (the squeezed conv5
 * output, i.e. BEFORE the permute at resnet.py:185): w_t = <x[:, :, t], a>,
 * alpha = softmax_T(tanh(w)), weighted = x*alpha, out = [sum_T weighted ,
 * unbiased std_T(weighted + noise)] -> (B, 2C).
 * noise: NULL (none) or (B, T, C) already scaled by 1e-5 (reference layout).
 */
int air_selfatt_pool_fwd(const float* x, int B, int C, int T, const float* att_w,
                         const float* noise, float* out, float* alpha_save, air_stream_t stream);
int air_selfatt_pool_bwd(const float* x, int B, int C, int T, const float* att_w,
                         const float* noise, const float* alpha, const float* out,
                         const float* dout, float* dx, float* datt_partial /*(B,C)*/,
                         air_stream_t stream);

/* out[n] = sum_m x[m][n]: folds the per-utterance datt partials over the batch. */
int air_sum_rows(const float* x, int M, int N, float* out, air_stream_t stream);
/* scale * N(0,1) from a Philox4x32-10 counter stream (seed, offset): the on-device
 * replacement for the host-side ``1e-5*torch.randn`` of resnet.py:38. */
int air_randn(float* out, size_t n, uint64_t seed, uint64_t offset, float scale,
              air_stream_t stream);
/* The same draw with the stream offset held in DEVICE memory: reads *counter as the offset and advances it by
 * ceil(n / 4) behind the draw (a second, 1-thread launch on the same stream).  resnet.py:38 draws fresh noise on
 * every call; inside a captured hipGraph a host-side offset would be frozen into the graph, a device-side one is
 * not - eager launches and graph replays walk the same (seed, offset) sequence.  counter: 8-byte aligned. */
int air_randn_ctr(float* out, size_t n, uint64_t seed, uint64_t* counter, float scale, air_stream_t stream);

/* --------------------------------------------------------------- linear ---
 * nn.Linear (resnet.py:143-144,187-189; ecapa_tdnn.py:148-149): y = x W^T + b.
 * x (M,K), w (N,K), y (M,N).
 */
int air_linear_fwd(const float* x, const float* w, const float* b, int M, int K, int N, float* y,
                   air_stream_t stream);
int air_linear_relu_fwd(const float* x, const float* w, const float* b, int M, int K, int N,
                        float* y, air_stream_t stream); /* y = max(0, x W^T + b) */
int air_linear_bwd(const float* x, const float* w, const float* dy, int M, int K, int N,
                   float* dx, float* dw, float* db, air_stream_t stream);

/* ------------------------------------------------------ ECAPA small ops ---
 * (B, C, T) fp32, time contiguous; *_bstride = batch stride in floats (0 = contiguous).
 */
/* out = a (+ b): copies / joins channel groups that are views of wider tensors
 * (torch.split / torch.cat / "sp + spx[i]" of ecapa_tdnn.py:71-83). */
int air_add_strided(float* out, size_t out_bstride, const float* a, size_t a_bstride,
                    const float* b, size_t b_bstride, int B, int C, int S, air_stream_t stream);
/* Res2 chain step (ecapa_tdnn.py:78-83): v = x*scale[c] + shift[c]; y1 = v (a channel slice of the
 * concat tensor, batch stride y1_bstride); optional y2 = v + add (dense; add is a channel slice with
 * batch stride add_bstride): the next branch's input "sp + spx[i+1]".  add and y2 both NULL or both set. */
int air_res2_bn_apply(const float* x, int B, int C, int S, const float* scale, const float* shift, float* y1,
                      size_t y1_bstride, const float* add, size_t add_bstride, float* y2, air_stream_t stream);
/* Both with a bf16 copy of the result (out / y1) in the weight-gradient operand layout of
 * air_conv1d_wgrad_bf16_pre: element (b, c, s) at bf16[b * bstride + c * tp + s], bstride in elements (0 = C * tp),
 * tp = air_conv1d_bf16_tp(S); the seven branch outputs and the pass-through group of a Bottle2neck together fill
 * the bf16 copy of its concat (conv3's X operand).  NULL = the plain entry points. */
int air_add_strided_ex(float* out, size_t out_bstride, const float* a, size_t a_bstride, const float* b,
                       size_t b_bstride, int B, int C, int S, unsigned short* out_bf16, size_t out_bf16_bstride,
                       int out_bf16_tp, air_stream_t stream);
int air_res2_bn_apply_ex(const float* x, int B, int C, int S, const float* scale, const float* shift, float* y1,
                         size_t y1_bstride, const float* add, size_t add_bstride, float* y2, unsigned short* y1_bf16,
                         size_t y1_bf16_bstride, int y1_bf16_tp, air_stream_t stream);
/* out[c] = sum_{b,s} x[b][c][s]: conv bias gradients (fp64 partials, fixed order). */
size_t air_channel_sum_ws_bytes(int B, int C);
int air_channel_sum(const float* x, int B, int C, int S, size_t bstride, float* out, void* ws,
                    size_t ws_bytes, air_stream_t stream);
/* mean_T and sqrt(clamp(var_T unbiased, clamp_min)) per (b,c) row: SE squeeze
 * (ecapa_tdnn.py:19) and the context statistics (:178).  std may be NULL. */
int air_row_stats(const float* x, int B, int C, int T, float* mean, float* std_or_null,
                  float clamp_min, air_stream_t stream);
/* dx (+)= dmean/T + dstd * (x - mean) / ((T-1) std).  relu_mask != 0: x is a ReLU output and the result
 * is zeroed where x == 0 (the stand-alone ReLU after layer4, ecapa_tdnn.py:173, folded in);
 * rowsum (B*C) or NULL: sum over time of each result row (summed over b it is the conv bias gradient). */
int air_row_stats_bwd(const float* x, int B, int C, int T, const float* mean, const float* std_,
                      const float* dmean, const float* dstd, float clamp_min, float* dx,
                      int accumulate, int relu_mask, float* rowsum_or_null, air_stream_t stream);
/* Same, and the result is also written as bf16 into dx_bf16[(b*C + c) * dx_bf16_tp + t] (operand layout of
 * air_conv1d_wgrad_bf16_pre; NULL = air_row_stats_bwd). */
int air_row_stats_bwd_ex(const float* x, int B, int C, int T, const float* mean, const float* std_,
                         const float* dmean, const float* dstd, float clamp_min, float* dx,
                         int accumulate, int relu_mask, float* rowsum_or_null, unsigned short* dx_bf16,
                         int dx_bf16_tp, air_stream_t stream);
/* dx *= (y > 0): backward of the stand-alone ReLU after layer4 (ecapa_tdnn.py:173). */
int air_relu_mask(float* dx, const float* y, size_t n, air_stream_t stream);
/* out[b][c] = sum_t x[b][c][t]: gradient of a per-utterance bias. */
int air_row_sum(const float* x, int B, int C, int T, float* out, air_stream_t stream);
/* SEModule gate + block residual (ecapa_tdnn.py:27-29,:93): out = x*sigmoid(z[b][c]) + res. */
int air_se_scale_fwd(const float* x, const float* z, const float* res, size_t res_bstride, int B,
                     int C, int T, float* out, size_t out_bstride, air_stream_t stream);
/* Same, with a bf16 copy of out at out_bf16[b * bstride + c * tp + t] (bstride in elements, 0 = C * tp): a block's
 * output is a channel slice of the (B, 1536, T) concat, and this fills the matching slice of the concat's bf16 copy. */
int air_se_scale_fwd_ex(const float* x, const float* z, const float* res, size_t res_bstride, int B,
                        int C, int T, float* out, size_t out_bstride, unsigned short* out_bf16,
                        size_t out_bf16_bstride, int out_bf16_tp, air_stream_t stream);
int air_se_scale_bwd(const float* x, const float* z, const float* dout, size_t dout_bstride, int B,
                     int C, int T, float* dx, float* dz, air_stream_t stream);
/* Attentive statistics pooling (ecapa_tdnn.py:143-185): softmax over T of the attention
 * logits (overwritten with the weights w), mu = sum x w, sg = sqrt(clamp(sum x^2 w - mu^2,
 * 1e-4)); out (B, 2C) = [mu | sg].  bwd overwrites w with d(logits). */
int air_asp_fwd(const float* x, float* logits_to_w, int B, int C, int T, float* out,
                air_stream_t stream);
/* rowsum_or_null (B*C): sum over time of each d(logits) row (summed over b: attention.3's bias gradient). */
int air_asp_bwd(const float* x, float* w_to_dlogits, int B, int C, int T, const float* out,
                const float* dout, float* dx, int accumulate, float* rowsum_or_null, air_stream_t stream);
/* Same, and d(logits) is also written as bf16 into dlogits_bf16[(b*C + c) * dlogits_bf16_tp + t] (NULL = air_asp_bwd). */
int air_asp_bwd_ex(const float* x, float* w_to_dlogits, int B, int C, int T, const float* out,
                   const float* dout, float* dx, int accumulate, float* rowsum_or_null,
                   unsigned short* dlogits_bf16, int dlogits_bf16_tp, air_stream_t stream);

/* ------------------------------------------- bf16-resident activations ---
 * BASELINE.json configs[2] ("ECAPA-TDNN-512 bf16 train"), round 3: every (B, C, T) activation between the layers
 * of ecapa_tdnn.py:64-95,152-187 lives in HBM as bf16 rows - element (b, c, t) at p[b * bs + c * Tp + t] with
 * Tp = air_h_tp(T) frames per row (a multiple of 256) and the frames T .. Tp - 1 ZERO; bs = batch stride in
 * ELEMENTS (0 = C * Tp), so channel slices of wider tensors (the Res2 split, the concats) are addressed in place.
 * Every entry point reads bf16, computes in fp32 and rounds each stored value once (nearest even) - the tensors
 * torch.autocast(bfloat16) would hold in bf16; what is rounded where is stated by oracle/ecapa.py
 * (bf16 = "resident").  Statistics, per-channel / per-row vectors, parameters and their gradients stay fp32.
 * Every writer keeps the padding zero; the GEMMs read the rows as operands without masks. */
int air_h_tp(int T);
/* K = 1 Conv1d (ecapa_tdnn.py:39,55,118,140,143) on bf16 rows, 256 x 256 bf16-MFMA GEMM reading x K-major:
 * dgrad = 0: y = relu?(W x + bias[co] + bias_bc[b][co]) (+ acc + acc2); dgrad = 1: dx = W^T dy + acc + acc2
 * (w is always the layer's (Cout, Cin, 1) fp32 weight; packed to bf16 into ws).  acc / acc2: bf16 rows of the
 * OUTPUT's channel count with their own batch strides, or NULL.  K (= Cin, or Cout for dgrad) % 64 == 0. */
size_t air_h_conv1d_ws_bytes(int Cout, int Cin);
int air_h_conv1d_pointwise(int B, int Cin, int Cout, int T, int Tp, const unsigned short* x, size_t x_bs, const float* w,
                           int dgrad, const float* bias, const float* bias_bc, int relu, const unsigned short* acc,
                           size_t acc_bs, const unsigned short* acc2, size_t acc2_bs, unsigned short* y, size_t y_bs,
                           void* ws, size_t ws_bytes, air_stream_t stream);
/* Same, and the epilogue that stores y also leaves the BatchNorm statistics of the STORED tensor (the conv -> ReLU ->
 * BatchNorm1d pairs of ecapa_tdnn.py:67-69,87-89,159-161,148-150) as {sum, sum of squares} per (channel, 64-frame
 * segment) in `stats` (air_h_conv1d_pointwise_stats_bytes(B, Cout, Tp) bytes, 8-byte aligned; NULL = the plain call);
 * air_h_bn_stats_ex takes it as stats_in.  Forward launches only (dgrad = 0). */
size_t air_h_conv1d_pointwise_stats_bytes(int B, int Cout, int Tp);
int air_h_conv1d_pointwise_ex(int B, int Cin, int Cout, int T, int Tp, const unsigned short* x, size_t x_bs,
                              const float* w, int dgrad, const float* bias, const float* bias_bc, int relu,
                              const unsigned short* acc, size_t acc_bs, const unsigned short* acc2, size_t acc2_bs,
                              unsigned short* y, size_t y_bs, void* stats, void* ws, size_t ws_bytes,
                              air_stream_t stream);
/* dw[co][ci] = sum_{b,t} dy x (fp32 out, split-K in fixed order); ws >= air_conv1d_bf16_ws_bytes of the layer. */
int air_h_conv1d_wgrad(int B, int Cin, int Cout, int T, int Tp, const unsigned short* x, size_t x_bs,
                       const unsigned short* dy, size_t dy_bs, float* dw, void* ws, size_t ws_bytes, air_stream_t stream);
/* Dilated K = 3 conv of a Res2 branch (ecapa_tdnn.py:46), forward (relu?(W * x + bias)) or data gradient
 * (dgrad = 1: w_packed from air_conv1d_tap_pack_bf16(transpose = 1)). */
int air_h_conv1d_tap(int B, int Cin, int Cout, int T, int Tp, int dil, const unsigned short* x, size_t x_bs,
                     const unsigned short* w_packed, int dgrad, const float* bias, int relu, unsigned short* y, size_t y_bs,
                     air_stream_t stream);
/* Same, and the epilogue that stores y also leaves the BatchNorm statistics of the STORED tensor (ecapa_tdnn.py:47-48,
 * 79-81: conv -> ReLU -> BatchNorm1d of a Res2 branch) as {sum, sum of squares} per (channel, 32-frame segment) in
 * `stats` (air_h_conv1d_tap_stats_bytes(B, Cout, Tp) bytes, 8-byte aligned; NULL = air_h_conv1d_tap).  Hand it to
 * air_h_bn_stats_ex as stats_in and the BatchNorm makes no pass over y (round 4: 21 passes per ECAPA step). */
size_t air_h_conv1d_tap_stats_bytes(int B, int Cout, int Tp);
int air_h_conv1d_tap_ex(int B, int Cin, int Cout, int T, int Tp, int dil, const unsigned short* x, size_t x_bs,
                        const unsigned short* w_packed, int dgrad, const float* bias, int relu, unsigned short* y,
                        size_t y_bs, void* stats, air_stream_t stream);
/* Same, for the DATA-GRADIENT launches of the Res2 chain (ecapa_tdnn.py:79-85 backward): y = d(input of branch i) is
 * the second half of the gradient that enters the BatchNorm of branch i - 1 (the first half, bn_dy, is that branch's
 * slice of the concat gradient; bn_x is the BatchNorm's input, the branch's ReLU output).  The epilogue also leaves
 * the five sums of that BatchNorm's backward pass - sum g, sum g xhat, and over bn_x > 0: sum g, the count, sum xhat
 * (g = bn_dy + y as stored, xhat = (bn_x - mean) invstd) - per (channel, 32-frame segment) in bn_sums
 * (air_h_conv1d_tap_bwd_sums_bytes(B, C, Tp) bytes, 16-byte aligned; bn_x / bn_dy 16-byte aligned rows).  Hand it to
 * air_h_bn_bwd_ex as sums_in: the BatchNorm backward then makes no first pass over (bn_x, bn_dy, y).  bn_sums NULL =
 * air_h_conv1d_tap_ex. */
size_t air_h_conv1d_tap_bwd_sums_bytes(int B, int C, int Tp);
int air_h_conv1d_tap_ex2(int B, int Cin, int Cout, int T, int Tp, int dil, const unsigned short* x, size_t x_bs,
                         const unsigned short* w_packed, int dgrad, const float* bias, int relu, unsigned short* y,
                         size_t y_bs, void* stats, const unsigned short* bn_x, size_t bn_x_bs,
                         const unsigned short* bn_dy, size_t bn_dy_bs, const float* bn_mean, const float* bn_invstd,
                         void* bn_sums, air_stream_t stream);
/* Round 6: the Res2 chain's elementwise passes folded into the staging of the conv that consumes them
 * (ecapa_tdnn.py:78-83: `sp = sp + spx[i]; sp = convs[i](sp); sp = bns[i](relu(sp))` - the BatchNorm of branch i - 1
 * needs batch statistics, so it cannot go into the launch that produces r_{i-1}; it goes into the NEXT launch's
 * operand read instead).  Same arithmetic and the same rounding points as air_h_res2_bn_apply / air_h_bn_bwd followed
 * by air_h_conv1d_tap_ex2: every stored tensor is bit-identical (tests/test_ecapa_bf16_gpu.py).  64 -> 64 channels only
 * (air_h_conv1d_tap_pro_ok); pro NULL or kind 0 = air_h_conv1d_tap_ex2.
 *   kind 1 (forward, dgrad = 0): x = r_{i-1} (raw ReLU output of the previous branch).  y1 = bf16(x * pa[c] + pb[c])
 *     -> side1 (its slice of the concat); operand t = bf16(y1 + g0) (g0 = o1's slice i) -> side0 (kept for the weight
 *     gradient).  pa / pb = the BatchNorm's scale / shift.
 *   kind 2 (data gradient, dgrad = 1): x = r_i (the BatchNorm input of THIS branch), g = g0 (+ g1) the gradient of its
 *     output; operand dc = bf16(pc pb (g - pd / N - xhat pe / N)), xhat = (x - pa) pb, zero where x <= 0 -> side0 (the
 *     weight gradient's dy).  pa / pb / pc / pd / pe = mean / invstd / gamma / dbeta / dgamma (air_h_bn_bwd_ex with
 *     dx = NULL leaves dgamma / dbeta / dbias without the apply pass).
 * side0 / side1 must not alias x, g0 or g1 (those are read with a halo of `dil` frames by neighbouring workgroups). */
typedef struct AirTapPrologue {
  int kind;
  const unsigned short* g0; size_t g0_bs;
  const unsigned short* g1; size_t g1_bs;
  const float* pa; const float* pb; const float* pc; const float* pd; const float* pe;
  unsigned short* side0; size_t side0_bs;
  unsigned short* side1; size_t side1_bs;
} AirTapPrologue;
int air_h_conv1d_tap_pro_ok(int Cin, int Cout);
int air_h_conv1d_tap_pro(int B, int Cin, int Cout, int T, int Tp, int dil, const unsigned short* x, size_t x_bs,
                         const unsigned short* w_packed, int dgrad, const float* bias, int relu, unsigned short* y,
                         size_t y_bs, void* stats, const unsigned short* bn_x, size_t bn_x_bs,
                         const unsigned short* bn_dy, size_t bn_dy_bs, const float* bn_mean, const float* bn_invstd,
                         void* bn_sums, const AirTapPrologue* pro, air_stream_t stream);
/* Weight gradients of the n_branches dilated K = 3 convs of one Res2 block (ecapa_tdnn.py:46, W -> W channels,
 * padding = dilation) in ONE launch: dw[i] (W, W, 3) fp32 = sum_{b,t} dy[i][b][co][t] x[i][b][ci][t + (k - 1) dil],
 * bf16 MFMA over the resident operands (products exact, fp32 sums, fixed-order split over the utterances) - the
 * fp32 contraction of the widened operands without the widened copies.  x / dy / dw: HOST arrays of n_branches
 * device pointers, x_bs / dy_bs their batch strides in elements (NULL or 0 = W * Tp).  W % 64 == 0, dil in 2..4,
 * n_branches <= 16, Tp % 64 == 0; ws >= air_h_conv1d_tap_wgrad_ws_bytes(). */
size_t air_h_conv1d_tap_wgrad_ws_bytes(int n_branches, int B, int W);
int air_h_conv1d_tap_wgrad(int n_branches, int B, int W, int T, int Tp, int dil, const unsigned short* const* x,
                           const size_t* x_bs, const unsigned short* const* dy, const size_t* dy_bs, float* const* dw,
                           void* ws, size_t ws_bytes, air_stream_t stream);
/* BatchNorm1d, training mode: statistics of the bf16 tensor (fp64 two-stage sums, running-stat update) ... */
size_t air_h_bn_ws_bytes(int B, int C);
int air_h_bn_stats(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                   float* mean, float* invstd, float* scale, float* shift, void* ws, size_t ws_bytes,
                   air_stream_t stream);
/* stats_in / stats_bytes: the records air_h_conv1d_tap_ex wrote for x (merged in fp64, a workgroup per channel; x is
 * not read), or NULL / 0 = air_h_bn_stats. */
int air_h_bn_stats_ex(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, const void* stats_in,
                      size_t stats_bytes, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                      void* ws, size_t ws_bytes, air_stream_t stream);
/* ... y = bf16(x * scale[c] + shift[c]); rowmean (B*C, may be NULL) = mean over t of the STORED y (the SE squeeze) */
int air_h_bn_apply(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, const float* scale,
                   const float* shift, unsigned short* y, size_t y_bs, float* rowmean, air_stream_t stream);
/* ... and its backward for conv -> ReLU -> BN (x = the BN input, a ReLU output; relu_in masks dx where x == 0):
 * incoming gradient dy + dy2 + rowbias_scale * dy_rowbias[b][c] as in air_bn_bwd_ex; dx bf16 (may alias dy);
 * dgamma / dbeta / dbias (C) fp32. */
int air_h_bn_bwd(const unsigned short* x, size_t x_bs, const unsigned short* dy, size_t dy_bs, const unsigned short* dy2,
                 size_t dy2_bs, const float* dy_rowbias, float rowbias_scale, int B, int C, int T, int Tp,
                 const float* mean, const float* invstd, const float* gamma, int relu_in, unsigned short* dx,
                 size_t dx_bs, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                 air_stream_t stream);
/* (round 6) dx NULL: the sums / dgamma / dbeta / dbias only - the apply pass is the prologue of air_h_conv1d_tap_pro. */
/* sums_in / sums_bytes: the records air_h_conv1d_tap_ex2 wrote for this BatchNorm (merged in fp64, a workgroup per
 * channel; the first pass over x / dy / dy2 is skipped), or NULL / 0 = air_h_bn_bwd.  Needs relu_in = 1, no row bias. */
int air_h_bn_bwd_ex(const unsigned short* x, size_t x_bs, const unsigned short* dy, size_t dy_bs, const unsigned short* dy2,
                    size_t dy2_bs, const float* dy_rowbias, float rowbias_scale, int B, int C, int T, int Tp,
                    const float* mean, const float* invstd, const float* gamma, int relu_in, unsigned short* dx,
                    size_t dx_bs, float* dgamma, float* dbeta, float* dbias, const void* sums_in, size_t sums_bytes,
                    void* ws, size_t ws_bytes, air_stream_t stream);
/* Res2 chain step (ecapa_tdnn.py:78-83): y1 = bf16(x * scale + shift); y2 = bf16(y1 + add) (add, y2 both or neither) */
int air_h_res2_bn_apply(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, const float* scale,
                        const float* shift, unsigned short* y1, size_t y1_bs, const unsigned short* add, size_t add_bs,
                        unsigned short* y2, size_t y2_bs, air_stream_t stream);
/* SE gate + block residual (ecapa_tdnn.py:27-29,:93): out = bf16(x * sigmoid(z[b][c]) + res), and its backward
 * dx = bf16(dout * sigmoid(z)), dz = s (1 - s) sum_t dout x. */
int air_h_se_scale_fwd(const unsigned short* x, size_t x_bs, const float* z, const unsigned short* res, size_t res_bs,
                       int B, int C, int T, int Tp, unsigned short* out, size_t out_bs, air_stream_t stream);
int air_h_se_scale_bwd(const unsigned short* x, size_t x_bs, const float* z, const unsigned short* dout, size_t dout_bs,
                       int B, int C, int T, int Tp, unsigned short* dx, size_t dx_bs, float* dz, air_stream_t stream);
/* Context statistics (ecapa_tdnn.py:178) of a dense bf16 tensor and their backward folded into dx
 * (dx = bf16(dx + ...), ReLU mask of layer4's output, rowsum of the stored values): air_row_stats[_bwd]. */
int air_h_row_stats(const unsigned short* x, int B, int C, int T, int Tp, float* mean, float* std_or_null,
                    float clamp_min, air_stream_t stream);
int air_h_row_stats_bwd(const unsigned short* x, int B, int C, int T, int Tp, const float* mean, const float* std_,
                        const float* dmean, const float* dstd, float clamp_min, unsigned short* dx, int accumulate,
                        int relu_mask, float* rowsum_or_null, air_stream_t stream);
/* Attentive statistics pooling (ecapa_tdnn.py:143-185): logits (bf16) -> w = bf16(softmax_T) in place; [mu | sg]
 * from the STORED w.  bwd: dx written (bf16), w overwritten with bf16(d logits), rowsum of the stored d logits. */
int air_h_asp_fwd(const unsigned short* x, unsigned short* logits_to_w, int B, int C, int T, int Tp, float* out,
                  air_stream_t stream);
int air_h_asp_bwd(const unsigned short* x, unsigned short* w_to_dlogits, int B, int C, int T, int Tp, const float* out,
                  const float* dout, unsigned short* dx, float* rowsum_or_null, air_stream_t stream);
/* Edges of the bf16 region: rows -> dense (B, C, T) fp32; rows -> rows (channel-slice copies). */
int air_h_to_f32(const unsigned short* x, size_t x_bs, int B, int C, int T, int Tp, float* y, air_stream_t stream);
int air_h_copy(const unsigned short* x, size_t x_bs, int B, int C, int Tp, unsigned short* y, size_t y_bs,
               air_stream_t stream);
/* The K = 5 first layer (ecapa_tdnn.py:111) as a pointwise GEMM: dense fp32 (B, Cin, T) -> bf16 rows (B, rows, Tp),
 * row ci K + k = x[ci][t + k dil - pad] (zeros outside [0, T), behind T, and in rows >= Cin K), so that
 * air_h_conv1d_pointwise / air_h_conv1d_wgrad with the (Cout, Cin, K) weight seen as (Cout, Cin K [+ zero columns])
 * are the layer's forward and weight gradient. */
int air_h_unfold(const float* x, size_t x_bs, int B, int Cin, int T, int Tp, int K, int dil, int pad, int rows,
                 unsigned short* y, size_t y_bs, air_stream_t stream);

/* ----------------------------------------------------------- OC-Softmax ---
 * AngularIsoLoss.forward == OCSoftmax.forward (loss.py:73-97, :187-206).
 * x (B,D), center (1,D), labels (B,) int64.  loss: scalar; neg_scores (B,).
 * bwd: dx (B,D), dcenter (1,D) for d(loss*gscale).
 * One workgroup each (the loss is summed in a fixed order): B <= 4096, else AIR_EUNSUPPORTED.
 */
int air_ocsoftmax_fwd(const float* x, const float* center, const int64_t* labels, int B, int D,
                      float r_real, float r_fake, float alpha, float* loss, float* neg_scores,
                      air_stream_t stream);
int air_ocsoftmax_bwd(const float* x, const float* center, const int64_t* labels, int B, int D,
                      float r_real, float r_fake, float alpha, const float* gscale_dev,
                      float* dx, float* dcenter, air_stream_t stream);

/* ------------------------------------------------------------ optimiser ---
 * torch.optim.Adam as configured at main_train.py:175-176 (coupled L2 weight
 * decay) and torch.optim.SGD(lr) (main_train.py:272), over flat fp32 buffers.
 * step is the 1-based step count.  grad_scale multiplies g first (1/world).
 */
int air_adam_step(float* p, const float* g, float* m, float* v, size_t n, int step, float lr,
                  float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  air_stream_t stream);
int air_sgd_step(float* p, const float* g, size_t n, float lr, float grad_scale,
                 air_stream_t stream);

/* Test instrumentation: occupy `nblocks` compute units for `milliseconds` (<= 200) on `stream` - one 64-thread
 * workgroup per block holding `lds_bytes` of LDS and spinning on the wall clock.  Stands in for collective (RCCL)
 * kernels that are resident on some CUs while the persistent convolution kernels are dispatched
 * (tests/test_cu_mask_gpu.py); the reference has no counterpart (main_train.py:174 is a commented-out DataParallel). */
int air_debug_cu_hog(int nblocks, int lds_bytes, double milliseconds, air_stream_t stream);

/* Measurement instrumentation: one wave on `stream` that writes n_samples pairs {wall-clock ticks (100 MHz), core-clock
 * counter (s_memtime)} into out[2 * n_samples], one pair every interval_us (>= 1; n_samples * interval_us <= 5 s).
 * Launched on a side stream next to the training step it gives the clock the compute units run at under that kernel
 * mix (bench.py "core_clock"): the MFMA-dense kernels are power-capped below the 2.4 GHz the peak figures assume.
 * No reference counterpart. */
int air_debug_clock_probe(unsigned long long* out, int n_samples, double interval_us, air_stream_t stream);

/* ------------------------------------------------- bench instrumentation ---
 * Opt-in HIP-event timing of the dominant kernels on their launch stream, used
 * by bench.py's roofline leg only (off by default).  kid indexes the kernel
 * template instance (air_prof_kernel_name).  work = algorithmic FLOPs (conv) or
 * bytes (LFCC) summed over the recorded launches. */
int air_prof_enable(int on);
int air_prof_kernel_count(void);
const char* air_prof_kernel_name(int kid);
int air_prof_collect(int kid, int* launches, double* total_ms, double* total_work);
/* The same plus total_issued: the FLOPs those launches sent to the matrix pipe - equal to
 * total_work for the direct kernels, fewer for the Winograd kernels (36 or 30 multiplies per
 * 4x4 / 3x4 output tile, padded tiles included) - and total_bytes: their algorithmic HBM bytes
 * (every operand read once, every result written once; 0 for kernels that do not state it). */
int air_prof_collect2(int kid, int* launches, double* total_ms, double* total_work,
                      double* total_issued, double* total_bytes);

/* ------------------------------------------------------------- utility ---- */
int air_add_inplace(float* y, const float* x, size_t n, air_stream_t stream); /* y += x */

#ifdef __cplusplus
}
#endif
#endif /* AIR_HIP_H */
