/*
 * pgv_hip.h — C ABI of the MI355X (gfx950) conv-VAE train-step hot path.
 *
 * The reference (gwendal-lv/preset-gen-vae) has no FFI: its hot path is stock torch.nn ops called from
 * Python (model/layer.py, model/encoder.py, model/decoder.py, model/VAE.py, model/loss.py, utils/audio.py,
 * train.py:203-248).  This header is the boundary the build introduces *below* that Python surface: every
 * entry point names the reference op(s) it replaces.  Conventions (all entry points):
 *
 *   - plain C: raw device pointers, ints, floats; no torch / C++ types;
 *   - tensors are fp32, contiguous, NCHW (the reference's layout);
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, nothing synchronises, nothing
 *     allocates (graph-capture safe); scratch comes from caller-provided workspaces;
 *   - return 0 on success, a negative PGV_E_* code otherwise (never throws across the ABI);
 *     pgv_last_error() returns a thread-local message for the last failure;
 *   - re-entrant; the only process-wide state is the kernel-selection policy (pgv_set_kernel_policy, a test aid).
 *
 * "big"/"small" naming for convolutions: a stride-s convolution maps a big tensor [B,Cb,Hb,Wb] to a small
 * one [B,Cs,Hs,Ws]; its transpose maps small to big.  The weight buffer is always [Cs][Cb][kh][kw], which is
 * torch's Conv2d layout [Cout,Cin,kh,kw] when big is the input and torch's ConvTranspose2d layout
 * [Cin,Cout,kh,kw] when small is the input, so both layer kinds share kernels and weights are never repacked.
 */
#ifndef PGV_HIP_H
#define PGV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGV_OK 0
#define PGV_E_INVALID (-1)   /* bad argument / unsupported shape */
#define PGV_E_LAUNCH (-2)    /* hipLaunch / runtime failure      */
#define PGV_E_WORKSPACE (-3) /* workspace too small              */

#define PGV_ACT_NONE 0
#define PGV_ACT_LEAKY_RELU 1 /* nn.LeakyReLU(slope)  encoder.py:239-240, decoder.py:203-204 */
#define PGV_ACT_HARDTANH 2   /* nn.Hardtanh(-1,1)    decoder.py:98,219 */

typedef struct pgv_conv_desc {
  int32_t B;               /* batch */
  int32_t Cb, Hb, Wb;      /* big tensor   [B,Cb,Hb,Wb] */
  int32_t Cs, Hs, Ws;      /* small tensor [B,Cs,Hs,Ws] */
  int32_t kh, kw;          /* kernel */
  int32_t stride, pad;     /* same on both axes (reference uses [2,2]/2 or [1,1]/0) */
  int32_t flags;           /* PGV_PREZEROED: the accumulated outputs of the call (stats / gw) already hold zeros */
  const void* w_shadow;    /* optional (may be NULL): the weight shadow of the call's weight tensor for THIS descriptor's
                              flags, written by pgv_conv_weight_shadow and still current.  PGV_COMPUTE_BF16: the rounded
                              weights, channel-innermost - the deep layers then run their bf16-native kernels (ABI v11).
                              PGV_COMPUTE_F32_SPLIT: three bf16 planes per direction in the kernels' fragment order -
                              forward / input-gradient calls of the layers listed at the flag run the six-instruction
                              kernels with it and the native fp32 kernels without it (ABI v13, large planes v14) */
} pgv_conv_desc;

/* Reduction outputs (BN statistics, weight / bias gradients, BN-backward projections) are accumulated with atomics.
 * By default every call clears its output first (one memset node per call).  With PGV_PREZEROED the caller promises
 * the buffer is zero - e.g. one arena cleared once per step, or the zero_grad'ed flat gradient buffer
 * (train.py:208) - and the call only accumulates. */
#define PGV_PREZEROED 1
/* bf16 compute / fp32 storage (BASELINE config 2): the operands of every product (activations after the folded
 * BatchNorm affine, weights, gradients) are rounded to bfloat16 (RNE) and multiplied on the bf16 matrix cores
 * (v_mfma_f32_16x16x32_bf16) with fp32 accumulation; tensors in HBM, BatchNorm, losses and Adam stay fp32. */
#define PGV_COMPUTE_BF16 2
/* fp32 products evaluated as SIX bf16 matrix instructions (fp32 mode only; ignored together with PGV_COMPUTE_BF16):
 * every operand value x is held as three bfloat16 terms x1 + x2 + x3 (exact), the product as the six largest cross terms
 * (v_mfma_f32_16x16x32_bf16, smallest term first) with fp32 accumulation - the dropped terms are below 2^-23 of the
 * product, the measured error against float64 is at or below that of v_mfma_f32_16x16x4_f32 (tests/test_gpu_kernels.py:
 * within 1.25 x + 1e-7 of it on every kernel).  Only the layers with a kernel for it change - every k4 s2 p2 layer of the
 * reference stacks with more than one input channel:
 *   8 <-> 16 channels on 129x174, 16 <-> 32 on 65x88, 32 <-> 64 on 33x45   forward, fused input gradient (pgv_bwd_fuse;
 *                                                                          the transposed direction without class sums),
 *                                                                          weight gradient
 *                                                                          (conv_big_split.hip, conv_wgrad_split.hip)
 *   64 <-> 128 on 17x23, 128 <-> 256 on 9x12, 256 <-> 512 on 5x7           forward, input gradient, weight gradient
 *   1x1 layers on 3x4 planes                                               forward, input gradient, weight gradient
 *                                                                          (conv_deep_split.hip, conv_deep_bf16.hip)
 * Forward / input-gradient calls need the layer's split weight shadow (pgv_conv_weight_shadow_bytes > 0 under this flag:
 * 12 bytes per weight) in pgv_conv_desc.w_shadow and compute natively without it; the weight gradient needs none (both
 * operands are activations, split in the kernel's loader) but needs the workspace of pgv_conv_wgrad_workspace_bytes.
 * Every other call computes as without the flag.  The library default is off (native fp32 instruction); bench.py times
 * the step with it on and says so in its config. */
#define PGV_COMPUTE_F32_SPLIT 8
/* The BatchNorm statistics output of a forward conv call (`stats`) is PGV_CLS_COPIES partial copies [copies][2C] of
 * doubles, zeroed by the caller, and a workgroup may add into any of them (the wave-specialised kernels use the copy of
 * their XCD: 256 workgroups finishing together on ONE copy serialise on its 2C addresses, 7-8 us per launch); they are
 * added up by whoever finalizes: pgv_bn_src.stats_copies / pgv_bn_finalize_src. */
#define PGV_STATS_COPIES 4

/* ---- library info ------------------------------------------------------------------------------ */
int pgv_abi_version(void);
const char* pgv_last_error(void);
/* 0 = prefer tuned kernels (default), 1 = force the generic one-thread-per-output kernels, 2 = tuned kernels but
 * without the shape-specialised band kernels of the reference layer shapes, 3 = policy 0 without the wave-specialised
 * second-generation kernels (1-3 are test / A-B timing aids; process-wide).  1-3 also keep the first-generation front-end
 * kernel in pgv_stft / pgv_stft_mel (0 runs the 16-frame-group kernel for hop 256 with a mel projection). */
int pgv_set_kernel_policy(int policy);

/* ---- convolutions (layer.Conv2D / layer.TConv2D bodies, model/layer.py:10-46) -------------------- */
/* small = act(conv_{stride,pad}(in') + bias[Cs]),  in' = big*in_scale[c]+in_shift[c] inside the image and 0
 * in the zero padding (in_scale/in_shift may be NULL: identity).  The per-channel affine is the *producer's*
 * BatchNorm2d folded into this consumer's load (layer.py:21-26 puts BN after the activation).
 * Replaces nn.Conv2d forward (layer.py:19-20) and, called with a gradient as `big`, ConvTranspose2d dgrad.
 * bias may be NULL.  stats (may be NULL) points to 2*Cs DOUBLES that are overwritten with the sum and the sum of
 * squares of the written outputs per channel (BatchNorm2d batch statistics, fused into the epilogue). */
int pgv_conv_down(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                  const float* w, const float* bias, int act, float slope, float* small, double* stats,
                  void* stream);

/* big = act(conv_transpose_{stride,pad}(in') + bias[Cb]); output_padding is implied by Hb/Wb.
 * Replaces nn.ConvTranspose2d forward (layer.py:38-40, decoder.py:218) and nn.Conv2d dgrad. */
int pgv_conv_up(const pgv_conv_desc* d, const float* small, const float* in_scale, const float* in_shift,
                const float* w, const float* bias, int act, float slope, float* big, double* stats,
                void* stream);

/* bf16 weight shadow (PGV_COMPUTE_BF16; nn.Conv2d / nn.ConvTranspose2d weights of the deep layers, model/encoder.py:64-69,249-255,
 * model/decoder.py:72-75,205-210): the weight tensor rounded to bfloat16 and laid out channel-innermost, once for the forward
 * and once for the transposed direction, so that the kernels stream half the bytes and copy weight slabs to LDS as they
 * are.  pgv_conv_weight_shadow_bytes: bytes of the shadow of this layer, 0 when the layer has no bf16-native kernels
 * (every call then behaves as without a shadow).  pgv_conv_weight_shadow writes it (16-byte aligned, caller-owned; one
 * launch, to be repeated whenever w changes: once per optimizer step); calls pass it in pgv_conv_desc.w_shadow. */
int64_t pgv_conv_weight_shadow_bytes(const pgv_conv_desc* d);
int pgv_conv_weight_shadow(const pgv_conv_desc* d, const float* w, void* shadow, void* stream);
/* The shadows of n <= 8 layers in one launch (the layers of one conv stack: a launch per layer costs 4-6 us of latency each);
 * every layer must have a shadow (pgv_conv_weight_shadow_bytes > 0).  ABI v12. */
int pgv_conv_weight_shadows(int n, const pgv_conv_desc* const* descs, const float* const* ws, void* const* shadows,
                            void* stream);

/* Optional fusion for input-gradient calls (backward of model/layer.py:21-26 under train.py:246).  The product of the
 * call is g, the gradient w.r.t. the BatchNorm output o of the next-lower block; with a pgv_bwd_fuse it is never
 * stored: the epilogue applies that block's BatchNorm + activation backward and writes the gradient w.r.t. its
 * pre-activation convolution output instead,
 *     out = g_y = act'(a) * (coef[c] * g + coef[C + c] * a + coef[2C + c]),
 * and accumulates the block's bias gradient gbias[c] += sum g_y (float atomics; the caller clears gbias).
 * coef ([3C] floats) comes from pgv_bn_bwd_coef (train-mode BatchNorm), or is (scale, 0, 0) for an eval-mode
 * BatchNorm and (1, 0, 0) for a block without BatchNorm.  act' is recovered from the saved activated tensor a.
 * Kernel families without the fused epilogue write g and run pgv_act_bwd_coef in place. */
#define PGV_CLS_COPIES 8
typedef struct pgv_bwd_fuse {
  const float* a;    /* saved activated output of the lower block, same shape as the output tensor */
  const float* coef; /* [3C] */
  float* gbias;      /* [C], accumulated into; may be NULL */
  int32_t act;       /* activation of the lower block (PGV_ACT_*) */
  float slope;
  float* cls;        /* optional: g_y summed by (row parity, column parity) class, cls[copy][c][2 * (row & 1) + (col & 1)],
                        accumulated into PGV_CLS_COPIES partial copies of [C][4] (the caller clears them; a workgroup
                        adds into the copy of its XCD - 256 workgroups finishing together on ONE copy serialise on its
                        addresses, ~80 ns per atomic) that the consumers add up: pgv_conv_tap_sums / pgv_coef_req.cls
                        with the output gradient of a stride-2 ConvTranspose2d */
  int32_t gbias_copies; /* 0: gbias is [C].  PGV_CLS_COPIES: gbias is that many zeroed partial copies of [C] (a workgroup
                           adds into the copy of its XCD - the same-address atomics of 256 workgroups finishing together
                           cost the fused kernels 3-6 us each); the sums reach the bias gradient through the pgv_bias_req
                           of the block's own weight-gradient call (pgv_conv_wgrad_ex), which runs next */
} pgv_bwd_fuse;
int pgv_conv_down_fused(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                        const float* w, const float* bias, int act, float slope, float* small, double* stats,
                        const pgv_bwd_fuse* fuse, void* stream);
int pgv_conv_up_fused(const pgv_conv_desc* d, const float* small, const float* in_scale, const float* in_shift,
                      const float* w, const float* bias, int act, float slope, float* big, double* stats,
                      const pgv_bwd_fuse* fuse, void* stream);

/* BatchNorm backward of block l WITHOUT a pass over the gradient g of its output o = a*scale + shift: both projections
 * follow from what block l+1 (the consumer of o; descriptor d, weights w) has already produced.  Since g = W^T gy,
 *     sum_p g[c,p] * o[c,p] = sum_{c',tap} w * gw        over the weight slice of channel c   (gw = pgv_conv_wgrad of block
 *                                                          l+1, evaluated with the same folded affine on o), and
 *     sum_p g[c,p]          = sum_{c',tap} w * T[c'][tap]  with T = pgv_conv_tap_sums of gy, the other wgrad operand.
 * lower_is_big != 0: o is the big tensor of d (block l+1 is a Conv2d), else the small one (ConvTranspose2d).
 * Outputs: coef[3C] for pgv_bwd_fuse / pgv_act_bwd_coef with n = B*H*W elements per channel,
 *     coef[c] = scale, coef[C+c] = -scale*rstd*S2/n, coef[2C+c] = -scale*(S1 - mean*rstd*S2)/n,
 *     S1 = sum g, S2 = sum g*a_hat = (sum g*o - beta*S1)/gamma  (beta = shift + mean*scale, gamma = scale/rstd),
 * and the BatchNorm parameter gradients ggamma[c] = S2, gbeta[c] = S1 (either may be NULL).  A channel whose scale is
 * exactly 0 gets S2 = 0.  flags: PGV_COMPUTE_BF16 = w is read rounded to bfloat16, as the products saw it. */
int pgv_bn_bwd_coef(const pgv_conv_desc* d, int lower_is_big, const float* w, const float* gw, const double* T,
                    const float* scale, const float* shift, const float* mean, const float* rstd, int64_t n,
                    float* coef, float* ggamma, float* gbeta, void* stream);
/* pgv_conv_tap_sums (border form when cls is given) followed by pgv_bn_bwd_coef, as one call: gy is the other operand
 * of the consumer's weight gradient (its output gradient).  T: [C_gy * kh*kw + 1] doubles of scratch holding zeros
 * (flags & PGV_PREZEROED) or cleared by the call.  (A single-launch form - the last workgroup of the tap-sum kernel
 * computing the coefficients - was measured at 43-164 us against 5 + 5: one workgroup streaming the whole weight
 * gradient is far slower than a launch.) */
int pgv_bn_bwd_coef_from_gy(const pgv_conv_desc* d, int lower_is_big, const float* gy, const float* cls, double* T,
                            const float* w, const float* gw, const float* scale, const float* shift, const float* mean,
                            const float* rstd, int64_t n, float* coef, float* ggamma, float* gbeta, int flags,
                            void* stream);
/* T[c][kh][kw] = sum over the batch and over the positions of gy[:,c] that kernel tap (kh,kw) pairs with a position
 * inside the OTHER tensor of d (what a channel of ones there would receive as weight gradient).  gy_is_big != 0: gy is
 * the big tensor [B,Cb,Hb,Wb] (gradient of a ConvTranspose2d output), else the small one.  T: [C][kh*kw] doubles,
 * overwritten (flags & PGV_PREZEROED: accumulated into).
 * cls (may be NULL) = the class sums of gy (pgv_conv_class_sums, or the producer's own: for the small tensor simply the
 * per-channel sum, i.e. the bias gradient of the block that owns gy): with them only the few border rows and columns of
 * gy that some tap cannot pair are read, T = class sum - unpaired border positions; without them gy is read in full.
 * Layout of cls: gy_is_big (m = stride classes per axis) - PGV_CLS_COPIES partial copies of [C][m*m], added up here
 * (pgv_bwd_fuse.cls; pgv_conv_class_sums fills copy 0); else [C], one copy. */
int pgv_conv_tap_sums(const pgv_conv_desc* d, int gy_is_big, const float* gy, const float* cls, double* T, int flags,
                      void* stream);
/* cls[c][(r mod m)*m + (w mod m)] = sum over the batch of gy[:,c,r,w], m = stride when gy is the big tensor of d (a tap
 * of a transposed convolution reaches one residue class of rows / columns), m = 1 (plain channel sums) when it is the
 * small one.  cls: gy_is_big - [PGV_CLS_COPIES][C][m*m] floats (copy 0 receives the sums, the other copies are cleared),
 * else [C]; overwritten (flags & PGV_PREZEROED: accumulated into). */
int pgv_conv_class_sums(const pgv_conv_desc* d, int gy_is_big, const float* gy, float* cls, int flags, void* stream);
/* g_y = act'(a) * (coef[c]*g + coef[C+c]*a + coef[2C+c]), gbias[c] += sum g_y: the unfused form of pgv_bwd_fuse
 * (in place allowed).  gbias may be NULL; flags: PGV_PREZEROED refers to gbias. */
int pgv_act_bwd_coef(const float* g, const float* a, const float* coef, int B, int C, int HW, int act, float slope,
                     float* g_y, float* gbias, int flags, void* stream);

/* gw[cs][cb][kh][kw] = sum_{b,oh,ow} small'[b,cs,oh,ow] * big'[b,cb,oh*s-p+kh,ow*s-p+kw]
 * (autograd of both layer kinds).  Either operand may carry a folded BN affine.
 * gw is overwritten (with PGV_PREZEROED in d->flags: added to).  workspace: pgv_conv_wgrad_workspace() bytes, 16-byte
 * aligned, private to the call until it has completed on `stream` (the wave-specialised kernels keep one partial
 * gradient per workgroup / wave there and a second launch adds them up); with a null or smaller workspace the call
 * falls back to kernels that flush with float atomics. */
int64_t pgv_conv_wgrad_workspace(const pgv_conv_desc* d);
int pgv_conv_wgrad(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                   const float* small, const float* small_scale, const float* small_shift, float* gw,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* pgv_conv_wgrad and, behind it, the BatchNorm-backward coefficients of the block BELOW (pgv_bn_bwd_coef_from_gy with gy =
 * the operand of this call that is the block's output gradient - `small` when lower_is_big, else `big` - and gw = this
 * call's result), as one call.  With the wave-specialised weight-gradient kernels the partial-gradient reduce pass and
 * the border tap sums (independent of each other) share ONE launch, followed by the coefficient kernel: two dependent
 * launches behind the weight-gradient kernel instead of three; otherwise the pieces run one after the other.
 * scratch: zeroed doubles, >= max(C_gy*kh*kw, 1024) + 1, private to the call (the tap sums of few channels are spread
 * over partial copies).  With PGV_PREZEROED gw must hold zeros (the
 * coefficients are those of THIS call's gradient). */
typedef struct pgv_coef_req {
  int32_t lower_is_big;
  const float* cls;   /* class sums of gy (see pgv_conv_tap_sums); required */
  const float* w;     /* weights of this block, [Cs][Cb][kh][kw] */
  const float *scale, *shift, *mean, *rstd; /* of the lower block's BatchNorm */
  int64_t n;          /* elements per channel of the lower block's output */
  float *coef, *ggamma, *gbeta;
  double* scratch;    /* max(C_gy*kh*kw, 1024) + 1 doubles, zeroed: receives the tap sums of gy */
  int32_t cls_copies; /* 0: by the rule of pgv_conv_tap_sums (copies for the big tensor, one [C] for the small one);
                         PGV_CLS_COPIES: cls is partial copies also for the small tensor (a bias gradient kept as
                         pgv_bwd_fuse.gbias_copies) */
} pgv_coef_req;
int pgv_conv_wgrad_coef(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                        const float* small, const float* small_scale, const float* small_shift, float* gw,
                        void* workspace, int64_t workspace_bytes, const pgv_coef_req* req, void* stream);
/* ... and the bias gradient of the block itself from partial copies (pgv_bwd_fuse.gbias_copies), as one more role of the
 * reduce launch: gbias[c] = (accumulate ? gbias[c] : 0) + sum over the PGV_CLS_COPIES copies of copies[.][c].  req / bias
 * may be NULL. */
typedef struct pgv_bias_req {
  const float* copies; /* [PGV_CLS_COPIES][C] */
  float* gbias;        /* [C] */
  int32_t C;
  int32_t accumulate;
} pgv_bias_req;
int pgv_conv_wgrad_ex(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                      const float* small, const float* small_scale, const float* small_shift, float* gw, void* workspace,
                      int64_t workspace_bytes, const pgv_coef_req* req, const pgv_bias_req* bias, void* stream);

/* ---- BatchNorm (nn.BatchNorm2d / BatchNorm1d train mode, layer.py:21-26, encoder.py:86-87) ------- */
/* stats[0:C] = sum, stats[C:2C] = sum of squares over (B,HW) of a[B,C,HW]. Overwrites stats.
 * Cross-workgroup accumulation is in float64 (one double atomic per workgroup): the per-channel sums feed
 * E[x^2]-E[x]^2 and the BN-backward projections, which cancel heavily (numerics). */
int pgv_bn_stats(const float* a, int B, int C, int HW, double* stats, void* stream);
/* From stats: mean/biased var -> scale=gamma*rstd, shift=beta-mean*scale; saves mean,rstd;
 * running_mean/var momentum update with the unbiased variance (torch semantics); any of running_* may be NULL. */
int pgv_bn_finalize(const double* stats, int C, int64_t n, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* scale, float* shift, float* mean, float* rstd, void* stream);
/* (num_batches_tracked, may be NULL, is incremented by one: nn.BatchNorm's counter, same launch) */
/* The arguments of pgv_bn_finalize as a value: the BatchNorm whose statistics a producer kernel has just accumulated and
 * whose affine the NEXT kernel applies to its input.  pgv_conv_down_bn / pgv_conv_up_bn / pgv_dropout_fwd_bn =
 * pgv_bn_finalize(src...) followed by the plain call with in_scale = src->scale, in_shift = src->shift - as ONE launch
 * where the kernel that serves the shape evaluates the (per-channel, float64) finalize arithmetic in its prologue: every
 * workgroup for itself, the first one also writing scale / shift / mean / rstd and the running statistics for the
 * backward pass.  A ~5 us dependent launch less per BatchNorm layer. */
typedef struct pgv_bn_src {
  const double* stats; /* [2C] sums and sums of squares (pgv_bn_stats layout) */
  int64_t n;           /* elements per channel */
  const float *gamma, *beta; /* nullable */
  float eps, momentum;
  float *running_mean, *running_var; /* nullable */
  int64_t* num_batches_tracked;      /* nullable */
  float *scale, *shift, *mean, *rstd; /* [C] each, outputs; scale and shift required */
  int32_t stats_copies;               /* 0 / 1: stats is [2C]; PGV_CLS_COPIES: partial copies (PGV_STATS_COPIES) */
} pgv_bn_src;
/* pgv_bn_finalize with its arguments as a pgv_bn_src (adds up the partial copies of the statistics, if any). */
int pgv_bn_finalize_src(const pgv_bn_src* src, int C, void* stream);
int pgv_conv_down_bn(const pgv_conv_desc* d, const float* big, const pgv_bn_src* in_bn, const float* w, const float* bias,
                     int act, float slope, float* small_out, double* stats, void* stream);
int pgv_conv_up_bn(const pgv_conv_desc* d, const float* small_in, const pgv_bn_src* in_bn, const float* w,
                   const float* bias, int act, float slope, float* big_out, double* stats, void* stream);
/* pgv_conv_up (in_bn null: in_scale / in_shift, both may be null) or pgv_conv_up_bn of an OUTPUT block (no statistics) plus the
 * squared-error reconstruction criterion where the output is produced (ABI v16; train.py:104,222 MSELoss / L2Loss against
 * the minibatch, model/decoder.py:218-219 the output layer): with an upstream gradient of exactly 1 for the criterion's value,
 *   g_y = act'(big_out) * 2 * scale * (big_out - target),  gbias[Cb] += sum g_y,  *loss_acc += scale * sum (big_out - target)^2,
 *   cls[PGV_CLS_COPIES][4] += sums of g_y by (row parity, column parity) (single-channel outputs; may be null)
 * - what pgv_sqerr_act_bwd_cls computes from big_out in a pass of its own.  *fused = 1: done; *fused = 0: this shape / mode
 * has no fused kernel and NOTHING was launched (the caller runs pgv_conv_up[_bn] and pgv_sqerr_act_bwd[_cls]).  Accumulators
 * are added to (the caller clears them). */
int pgv_conv_up_sqerr(const pgv_conv_desc* d, const float* small_in, const pgv_bn_src* in_bn, const float* in_scale,
                      const float* in_shift, const float* w, const float* bias, int act, float slope, float* big_out,
                      const float* target, float scale, float* g_y, float* gbias, float* loss_acc, float* cls, int* fused,
                      void* stream);
/* Eval-mode BN folded to an affine: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale
 * (validation forward, train.py:261-291). */
int pgv_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int C, float* scale, float* shift, void* stream);
/* o = a*scale[c]+shift[c]  (materialised BN output; in place allowed). */
int pgv_affine_nchw(const float* a, const float* scale, const float* shift, int B, int C, int HW, float* o,
                    void* stream);
/* nn.BatchNorm1d over x[B][C] (encoder.py:86-87: the 'bn' latent regularisation), train mode, one launch per direction
 * (the tensor is [256][128]: statistics / finalize / apply as separate launches cost more in launch latency than in
 * work).  Forward: batch statistics in float64, y = x*scale + shift, saves scale / mean / rstd for the backward,
 * momentum update of the running statistics with the unbiased variance, num_batches_tracked += 1 (any of those
 * pointers may be NULL).  Backward: gx = scale*(g - mean(g) - x_hat*mean(g*x_hat)), ggamma = sum g*x_hat, gbeta = sum g. */
int pgv_bn1d_fwd(const float* x, int B, int C, const float* gamma, const float* beta, float eps, float momentum,
                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y, float* scale,
                 float* mean, float* rstd, void* stream);
int pgv_bn1d_bwd(const float* g, const float* x, const float* scale, const float* mean, const float* rstd, int B, int C,
                 float* gx, float* ggamma, float* gbeta, void* stream);
/* red[0:C] = sum g_o, red[C:2C] = sum g_o * a_hat, a_hat=(a-mean)*rstd. Overwrites red (flags: PGV_PREZEROED). */
int pgv_bn_bwd_reduce(const float* g_o, const float* a, const float* mean, const float* rstd, int B, int C,
                      int HW, double* red, int flags, void* stream);
/* Backward through [activation -> BN]: given g_o (grad of BN output) produces g_y (grad of the pre-activation
 * conv output): g_a = scale[c]*(g_o - red[c]/n - a_hat*red[C+c]/n); g_y = g_a * act'(a).
 * With scale==NULL (block without BN) g_a = g_o; with red==NULL (eval-mode BN) g_a = scale[c]*g_o.
 * Also emits gbias[c] = sum g_y (bias gradient, may be NULL; flags: PGV_PREZEROED applies to it), and when BN is
 * present writes ggamma[c] = red[C+c], gbeta[c] = red[c] as float32 (either may be NULL).
 * act' is recovered from the saved activated tensor a (sign for LeakyReLU; |a|<1 for Hardtanh). */
int pgv_act_bn_bwd(const float* g_o, const float* a, const float* scale, const float* mean, const float* rstd,
                   const double* red, int B, int C, int HW, int act, float slope, float* g_y, float* gbias,
                   float* ggamma, float* gbeta, int flags, void* stream);
/* pgv_bn_bwd_reduce followed by pgv_act_bn_bwd (train-mode BatchNorm) as ONE launch for small planes - a workgroup per
 * channel keeps the channel's values in registers between the reduction and the application: same results (the sums are
 * float64 across the workgroup), 3 instead of 5 passes over the tensors.  pgv_bn_act_bwd_fusable: 1 when (B, C, HW) is
 * served (the deepest blocks of the 8-layer stack: at most 8 k values per channel, at least 96 channels), else the two
 * calls above are the way.  In place allowed (g_y == g_o); gbias as in pgv_act_bn_bwd.  ABI v13. */
int pgv_bn_act_bwd_fusable(int B, int C, int HW);
int pgv_bn_act_bwd_fused(const float* g_o, const float* a, const float* scale, const float* mean, const float* rstd, int B,
                         int C, int HW, int act, float slope, float* g_y, float* gbias, float* ggamma, float* gbeta,
                         int flags, void* stream);
/* Output block under a squared-error criterion, backward in one pass: with g = 2 * scale * g_loss[0] * (a - x) (the
 * gradient of scale * sum (a - x)^2, train.py:222 / loss.py:15-43, never materialised) it writes
 * g_y = act'(a) * g and accumulates gbias[c] += sum g_y - pgv_sqerr_bwd followed by pgv_act_bn_bwd without BatchNorm.
 * loss_acc (optional): += scale * sum (a - x)^2, the criterion's value as a by-product of the same pass (the caller
 * provides a zeroed scalar).  flags: PGV_PREZEROED refers to gbias. */
int pgv_sqerr_act_bwd(const float* a, const float* x, const float* g_loss, float scale, int B, int C, int HW, int act,
                      float slope, float* g_y, float* gbias, float* loss_acc, int flags, void* stream);
/* The same with the class sums of g_y as a by-product (see pgv_bwd_fuse.cls): planes of W columns, cls
 * [PGV_CLS_COPIES][C][4] floats accumulated into (the caller clears them); single-channel tensors with planes of >= 16384 elements only (the
 * spectrogram output layer). */
int pgv_sqerr_act_bwd_cls(const float* a, const float* x, const float* g_loss, float scale, int B, int C, int HW, int W,
                          int act, float slope, float* g_y, float* gbias, float* loss_acc, float* cls, int flags,
                          void* stream);

/* ---- fully-connected (nn.Linear, encoder.py:85, decoder.py:64) ----------------------------------- */
/* C[M,N] = alpha * op(A)[M,K] @ op(B)[K,N] + beta_bias: generic strided fp32 GEMM on f32 MFMA.
 * A element (m,k) at A[m*sam + k*sak]; B element (k,n) at B[k*sbk + n*sbn]; C row-major ldc.
 * bias_n (len N) optional (NULL).  Overwrites C.  flags: PGV_COMPUTE_BF16 rounds both operands to bfloat16 (fp32
 * accumulation), 0 = fp32; PGV_PREZEROED: C holds zeros on entry (e.g. a slice of the zero_grad'ed gradient / scratch
 * buffer) - a split-K product then accumulates straight into it without a clearing launch.  workspace is reserved
 * (split-K accumulates in C). */
int64_t pgv_gemm_workspace(int M, int N, int K);
int pgv_gemm(int M, int N, int K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
             int64_t sbn, float* C, int64_t ldc, const float* bias_n, int flags, void* workspace,
             int64_t workspace_bytes, void* stream);
/* out[n] (+)= sum_m x[m*ld + n]  (bias gradient of Linear).  flags: PGV_PREZEROED = out already holds zeros (or a
 * partial sum to add to); 0 = the call clears it first. */
int pgv_colsum(const float* x, int M, int N, int64_t ld, float* out, int flags, void* stream);

/* ---- dropout / reparameterisation / losses -------------------------------------------------------- */
/* Counter-based RNG (Philox4x32-10). rng_state: device uint64[2] = {seed, offset}; kernels only read it. */
/* mask[i] = (u_i >= p) ? 1/(1-p) : 0  (nn.Dropout train mode, encoder.py:85, decoder.py:65). */
int pgv_dropout_mask(const uint64_t* rng_state, uint64_t stream_id, float p, int64_t n, float* mask, void* stream);
/* nn.Dropout forward in one pass: y = x * mask with the mask of pgv_dropout_mask (same state, stream and counters)
 * drawn on the fly and stored for the backward product (encoder.py:85, decoder.py:65). */
int pgv_dropout_apply(const uint64_t* rng_state, uint64_t stream_id, float p, int64_t n, const float* x, float* y,
                      float* mask, void* stream);
/* nn.Dropout (train mode) without a stored mask, the form the train step uses: y = x' * mask with the mask
 * pgv_dropout_mask draws from the same state / stream; saved_state[2] receives a copy of the generator state it was drawn
 * from, and pgv_dropout_bwd regenerates the mask from that copy (gx = gy * mask; in place allowed).  x is [B][C][HW];
 * scale / shift (nullable, [C]): x' = x*scale[c] + shift[c] - the BatchNorm of the encoder's last conv block folded into
 * this pass (encoder.py:85) - else x' = x. */
int pgv_dropout_fwd(const uint64_t* rng_state, uint64_t stream_id, float p, const float* x, int64_t B, int C, int64_t HW,
                    const float* scale, const float* shift, float* y, uint64_t* saved_state, void* stream);
/* pgv_dropout_fwd with the affine given as the BatchNorm it comes from (pgv_bn_src above: finalize in the same launch) */
int pgv_dropout_fwd_bn(const uint64_t* rng_state, uint64_t stream_id, float p, const float* x, int64_t B, int C, int64_t HW,
                       const pgv_bn_src* bn, float* y, uint64_t* saved_state, void* stream);
int pgv_dropout_bwd(const uint64_t* saved_state, uint64_t stream_id, float p, int64_t n, const float* gy, float* gx,
                    void* stream);
/* pgv_dropout_bwd over a [M][N] gradient plus colsum[n] (+)= sum_m gx[m][n] in the same pass - the bias gradient of the
 * nn.Linear whose output the Dropout follows (decoder.py:64-65).  PGV_PREZEROED: colsum already holds zeros. */
/* pgv_dropout_bwd (gx = g_d * regenerated mask, [B][C][HW]) and pgv_bn_bwd_reduce over gx against the saved activation a
 * (red[0:C] += sum gx, red[C:2C] += sum gx * a_hat) as ONE pass: the Dropout in front of the encoder's Linear sits on
 * top of the last conv block's BatchNorm, whose backward needs these projections first. */
int pgv_dropout_bwd_bn_reduce(const uint64_t* saved_state, uint64_t stream_id, float p, const float* g_d, const float* a,
                              const float* mean, const float* rstd, int B, int C, int HW, float* gx, double* red, int flags,
                              void* stream);
int pgv_dropout_bwd_colsum(const uint64_t* saved_state, uint64_t stream_id, float p, int M, int N, const float* gy,
                           float* gx, float* colsum, int flags, void* stream);
/* eps ~ N(0,1) i.i.d. (VAE.py:54-55). */
int pgv_normal(const uint64_t* rng_state, uint64_t stream_id, int64_t n, float* out, void* stream);
/* rng_state[1] += inc (device side, keeps graph replays advancing). */
int pgv_rng_advance(uint64_t* rng_state, uint64_t inc, void* stream);
/* y = x * m (mask multiply; used forward and backward). */
int pgv_mul(const float* x, const float* m, int64_t n, float* y, void* stream);

/* BasicVAE.forward reparameterisation (VAE.py:49-56) + GaussianDkl (loss.py:57-66):
 *   mu = ml[b,0,:], lv = ml[b,1,:]; z = mu + exp(lv/2)*eps ; kl = 0.5*sum(exp(lv)+mu^2-lv-1) * kl_scale
 * kl_scale = 1/B or 1/(B*D).  kl (device scalar) is overwritten.  eps NULL => z = mu (eval mode, VAE.py:57-58). */
int pgv_reparam_kl_fwd(const float* ml, const float* eps, int B, int D, float kl_scale, float* z, float* kl,
                       void* stream);
/* g_ml = d/d(ml) [ <g_z, z> + g_kl * kl ] ; g_z may be NULL, g_kl is a device scalar pointer (may be NULL). */
/* The same with eps drawn inside: eps_out[i] = element i of what pgv_normal draws from the same state / stream (stored
 * for pgv_reparam_kl_bwd).  PGV_PREZEROED: kl already holds zero (no clearing launch). */
int pgv_reparam_kl_fwd_rng(const float* ml, const uint64_t* rng_state, uint64_t stream_id, int B, int D, float kl_scale,
                           float* z, float* eps_out, float* kl, int flags, void* stream);
/* The encoder head in one launch per direction: nn.BatchNorm1d (train mode, pgv_bn1d_fwd) over the Linear output
 * x[B][2D], then pgv_reparam_kl_fwd_rng over y = [mu | log-variance]; backward: pgv_reparam_kl_bwd (+ g_y, a gradient that
 * reaches y directly; g_z / g_kl / g_y nullable), pgv_bn1d_bwd over it, and colsum[c] = sum_b gx[b][c] (nullable; the bias
 * gradient of the Linear in front; PGV_PREZEROED: added to what colsum holds).  Same values as the separate calls. */
int pgv_bn1d_reparam_fwd(const float* x, int B, int D, const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y, float* scale,
                         float* mean, float* rstd, const uint64_t* rng_state, uint64_t stream_id, float kl_scale,
                         float* z, float* eps_out, float* kl, int flags, void* stream);
int pgv_bn1d_reparam_bwd(const float* g_z, const float* g_kl, const float* g_y, const float* y, const float* eps,
                         const float* x, const float* scale, const float* mean, const float* rstd, int B, int D,
                         float kl_scale, float* gx, float* ggamma, float* gbeta, float* colsum, int flags,
                         void* stream);
int pgv_reparam_kl_bwd(const float* ml, const float* eps, const float* g_z, const float* g_kl, int B, int D,
                       float kl_scale, float* g_ml, void* stream);

/* loss = scale * sum((xhat-x)^2)  (nn.MSELoss('mean'): scale=1/numel, train.py:104,222;
 * loss.L2Loss: scale=1/B[/numel-per-item], loss.py:37-43).  loss (device scalar) overwritten. */
int pgv_sqerr_fwd(const float* xhat, const float* x, int64_t n, float scale, float* loss, void* stream);
/* g[i] = g_loss * 2*scale*(xhat-x) * 1[|xhat|<1 if hardtanh]  — gradient w.r.t. the *pre-Hardtanh* tensor
 * when hardtanh!=0 (xhat is the clamped output; torch passes zero grad at/after the bounds). */
int pgv_sqerr_bwd(const float* xhat, const float* x, const float* g_loss, int64_t n, float scale, int hardtanh,
                  float* g, void* stream);

/* ---- optimizer (torch.optim.Adam, coupled L2, train.py:166-167) ---------------------- */
/* hyper: device float[4] = {lr, bias_correction1 = 1-b1^t, bias_correction2 = 1-b2^t, grad_scale}. */
int pgv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                  float beta2, float eps, float weight_decay, void* stream);
/* Device-side step counter: pows (device double[3], initialised to {1,1,0}): pows[0] *= beta1, pows[1] *= beta2,
 * pows[2] += 1 (the number of steps taken: the powers underflow - beta1^t after ~7000 steps - and cannot be inverted);
 * hyper[1] = 1-pows[0], hyper[2] = 1-pows[1].  Keeps the bias corrections advancing under hipGraph replay. */
int pgv_adam_tick(double* pows, float* hyper, float beta1, float beta2, void* stream);
/* pgv_adam_tick plus the rest of a train step's single-thread bookkeeping in the same launch: the generator offset
 * rng_state[1] += rng_inc (pgv_rng_advance; rng_state nullable) and the reported loss total[0] = loss_a[0] +
 * loss_b[0] * weight_b[0] (+ loss_c[0]) (train.py:227,246; total / loss_c nullable).  finite (nullable, ABI v15): finite[0]
 * = 1 if every loss term and the total are finite, else 0 - the harness's check_nan_values / ModelConvergenceError test
 * (utils/exception.py:13-23, train.py:245) as a device flag that costs no launch and no synchronisation of its own.
 * Placed in front of pgv_adam_step. */
int pgv_step_tick(double* pows, float* hyper, float beta1, float beta2, uint64_t* rng_state, uint64_t rng_inc,
                  const float* loss_a, const float* loss_b, const float* weight_b, const float* loss_c, float* total,
                  float* finite, void* stream);

/* ---- STFT -> mel -> dB front-end (utils/audio.py:20-92, data/abstractbasedataset.py:129-131) -------- */
/* wav[B][n_samples] fp32 -> out[B][n_mels][n_frames]; n_fft=1024 only, hop any, centre zero padding.
 * mel CSR: row_ptr[n_mels+1], col[nnz], val[nnz] (Slaney basis built on the host).  n_mels==0 => linear
 * spectrogram with n_fft/2+1 rows.  out = affine_a * 20*log10(max(mag, floor)) + affine_b. */
int pgv_stft_mel(const float* wav, int B, int64_t n_samples, int n_fft, int hop, int n_frames,
                 const float* window, float norm, const int32_t* mel_row_ptr, const int32_t* mel_col,
                 const float* mel_val, int n_mels, float floor_lin, float affine_a, float affine_b, float* out,
                 void* stream);

/* The same transform with a choice of output (Spectrogram surface of utils/audio.py:24-61):
 *   PGV_STFT_DB       out[B][rows][n_frames] = affine_a * 20*log10(max(mag, floor_lin)) + affine_b   (= pgv_stft_mel;
 *                     Spectrogram.__call__ with log_scale=True :42-54, MelSpectrogram.__call__ :80-87)
 *   PGV_STFT_LINEAR   out[B][rows][n_frames] = mag = |STFT| / norm (mel-projected when n_mels > 0), no clamp, no
 *                     affine (Spectrogram(log_scale=False).__call__, :42-50)
 *   PGV_STFT_COMPLEX  out[B][n_fft/2+1][n_frames][2] = (re, im) of the un-normalised one-sided STFT
 *                     (Spectrogram.get_stft :33-40 = torch.stft(center=True, pad_mode='constant')); n_mels must be 0. */
#define PGV_STFT_DB 0
#define PGV_STFT_LINEAR 1
#define PGV_STFT_COMPLEX 2
int pgv_stft(const float* wav, int B, int64_t n_samples, int n_fft, int hop, int n_frames, const float* window,
             float norm, const int32_t* mel_row_ptr, const int32_t* mel_col, const float* mel_val, int n_mels,
             int out_mode, float floor_lin, float affine_a, float affine_b, float* out, void* stream);

/* ---- preset-parameter losses and metrics (model/loss.py:72-315) ------------------------ */
/* Tables of a PresetIndexesHelper, all DEVICE pointers (built once per helper by the caller):
 *   num_idx[n_num]        learnable columns of the numerical parameters (get_numerical_learnable_indexes)
 *   cat_idx[n_groups][K]  columns of every one-hot group, padded with -1 (get_categorical_learnable_indexes)
 *   rule_trig[n_rules]    useless-parameter rules (data/preset.py:259-281): rule r fires for a row when
 *                         u_in[row][rule_trig[r]] < 1e-3 (a Dexed operator at zero output level); n_rules <= 32
 *   num_rules[n_num], cat_rules[n_groups]   bit r set: the column / group is useless when rule r fires. */
typedef struct pgv_params_tables {
  int32_t n_num;
  const int32_t* num_idx;
  const uint32_t* num_rules;
  int32_t n_groups, K;
  const int32_t* cat_idx;
  const uint32_t* cat_rules;
  int32_t n_rules;
  const int32_t* rule_trig;
} pgv_params_tables;
#define PGV_PARAMS_CCE 0         /* cat_bce=False, cat_softmax=False: -log of the target's probability */
#define PGV_PARAMS_CCE_SOFTMAX 1 /* cat_softmax=True: softmax(q / softmax_t) first (loss.py:166-167) */
#define PGV_PARAMS_BCE 2         /* cat_bce=True: F.binary_cross_entropy(mean) / 8 (loss.py:172-174) */
/* SynthParamsLoss.__call__ (loss.py:118-183) and its gradient in one launch: loss[0] = numerical term (nn.MSELoss
 * 'mean' if normalize else L2Loss = sum / B, useless columns zeroed on both sides) + cat_factor * sum over groups of the
 * categorical term over the group's useful rows (/ n_groups if normalize); grad[B][L] (nullable) = d loss / d u_out,
 * zero in columns that no term reads.  u_in is not modified (the reference zeroes useless columns in place).
 * workspace: 8 * (1 + min(B, 1024)) bytes whose FIRST 8 BYTES ARE ZERO before the first call (an arrival counter that
 * every call leaves at zero); private to the call until it completes. */
int pgv_params_loss(const float* u_out, const float* u_in, int B, int L, const pgv_params_tables* t, int mode,
                    float softmax_t, int normalize, float cat_factor, float* loss, float* grad, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* The column pairs the two metrics compare, one item per VST parameter (tables are device pointers):
 *   PGV_PARAMS_COL_QUANTIZED     in = u_in[first], out = round(u_out[first] * (card-1)) / (card-1) if card > 0 else
 *                                u_out[first]                        (QuantizedNumericalParamsLoss, loss.py:231-240)
 *   PGV_PARAMS_COL_ONEHOT_VALUE  argmax over idx[first .. first+len) / (len-1) on both sides          (loss.py:242-252)
 *   PGV_PARAMS_COL_CLASS         round(u[first] * (card-1)) on both sides       (CategoricalParamsAccuracy, :287-297)
 *   PGV_PARAMS_COL_ONEHOT_CLASS  argmax over idx[first .. first+len) on both sides                          (:299-306)
 * in_cols / out_cols [B][n_items] (nullable), match[n_items] (nullable) = fraction of rows with in == out. */
#define PGV_PARAMS_COL_QUANTIZED 0
#define PGV_PARAMS_COL_ONEHOT_VALUE 1
#define PGV_PARAMS_COL_CLASS 2
#define PGV_PARAMS_COL_ONEHOT_CLASS 3
int pgv_params_columns(const float* u_out, const float* u_in, int B, int L, int n_items, const int32_t* kind,
                       const int32_t* first, const int32_t* len, const float* card, const int32_t* idx, float* in_cols,
                       float* out_cols, float* match, void* stream);

/* ---- misc -------------------------------------------------------------------------------------------- */
int pgv_fill(float* p, int64_t n, float v, void* stream);
/* y = a*x + y */
int pgv_axpy(int64_t n, float a, const float* x, float* y, void* stream);
/* Streaming copy used for bandwidth calibration in bench.py. */
int pgv_copy(const float* src, float* dst, int64_t n, void* stream);
/* Measured-peak probes for bench.py (SURVEY.md 8d asks for the roofline against measured AND nominal peaks; not part
 * of the train step, nothing in the reference corresponds to them).  pgv_probe_mfma: a dependency-free stream of
 * v_mfma_f32_16x16x4_f32 (bf16 = 0) or v_mfma_f32_16x16x32_bf16 (bf16 = 1), one wave per SIMD on every CU, `iters` x 32
 * instructions per wave; *flops (host pointer, nullable) receives the floating-point operations of the launch.
 * pgv_probe_read: a 16-byte-per-lane grid-stride read reduction over n floats (16-byte aligned). `sink`: one device
 * float that is never written in practice. */
int pgv_probe_mfma(int bf16, int iters, float* sink, int64_t* flops, void* stream);
int pgv_probe_read(const float* buf, int64_t n, float* sink, void* stream);

/* ---- tuning knobs ----------------------------------------------------------------------------------------
 * NOT part of the interface a binding should use: process-wide A/B switches of the timing scripts under scratch/ (and of
 * one test that covers a kept-alive alternative path).  None changes a result beyond summation order; each returns the
 * previous value.  pgv_dbg_set_deep_bf16_stamps takes a device buffer for in-kernel clock stamps of the bf16-native deep
 * kernels (NULL = off, the default). */
int pgv_dbg_set_gemm_tiles(int v);          /* gemm.hip: 1 = the bf16 MFMA tile kernels instead of gemm_frag.hip */
int pgv_dbg_set_gemm_variant(int v);        /* gemm_frag.hip: pipeline stages / grid cap / split-K selection */
int pgv_dbg_set_v2_down_variant(int v);     /* conv_v2_down.hip */
int pgv_dbg_set_wgrad_bf16_variant(int v);  /* conv_v2_wgrad.hip: bit 0 = band partial kernels instead of conv_wgrad_bf16.hip */
int pgv_dbg_set_deep_bf16_variant(int v);   /* conv_deep_bf16.hip */
void pgv_dbg_set_deep_bf16_stamps(void* device_buf);

#ifdef __cplusplus
}
#endif
#endif /* PGV_HIP_H */
