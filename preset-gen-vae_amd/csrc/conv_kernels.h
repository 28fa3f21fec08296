// Internal prototypes shared by the convolution translation units.
#pragma once
#include "pgv_common.h"

int pgv_conv_down_generic(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                          const float* w, const float* bias, int act, float slope, float* small_out,
                          hipStream_t st);
int pgv_conv_up_generic(const pgv_conv_desc* d, const float* small_in, const float* in_scale,
                        const float* in_shift, const float* w, const float* bias, int act, float slope,
                        float* big_out, hipStream_t st);
int pgv_conv_wgrad_generic(const pgv_conv_desc* d, const float* big, const float* big_scale,
                           const float* big_shift, const float* small_in, const float* small_scale,
                           const float* small_shift, float* gw, hipStream_t st);

// Tuned kernels: return 1 when they handled the call, 0 when the shape is not covered (caller falls back to
// the generic kernel), <0 on error.
int pgv_conv_down_tuned(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                        const float* w, const float* bias, int act, float slope, float* small_out, double* stats,
                        hipStream_t st);
int pgv_conv_up_tuned(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                      const float* w, const float* bias, int act, float slope, float* big_out, double* stats,
                      hipStream_t st);
int pgv_conv_wgrad_tuned(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                         const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                         void* workspace, int64_t workspace_bytes, hipStream_t st);
int64_t pgv_conv_wgrad_tuned_workspace(const pgv_conv_desc* d);

// Shape-specialised band kernels (conv_band.hip): compile-time tile geometry for the reference layer shapes.
int pgv_conv_down_band(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                       const float* w, const float* bias, int act, float slope, float* small_out, double* stats,
                       const pgv_bwd_fuse* fuse, hipStream_t st);

int pgv_conv_up_band(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* big_out, double* stats,
                     const pgv_bwd_fuse* fuse, hipStream_t st);
int pgv_conv_wgrad_band(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                        const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                        hipStream_t st);
int pgv_conv_wgrad_band_partial(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                                const float* small_in, const float* small_scale, const float* small_shift, float* partial,
                                int64_t partial_bytes, int* nparts, hipStream_t st);
// bf16-native weight gradient (conv_wgrad_bf16.hip): same workspace layout, the three stride-2 k=4 layers of the stack
int pgv_conv_wgrad_bf16_partial(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                                const float* small_in, const float* small_scale, const float* small_shift, float* partial,
                                int64_t partial_bytes, int* nparts, hipStream_t st);

// Second-generation kernels (conv_v2.hip): one workgroup per CU, waves split M, weights from registers; tried first.
// (bn != null: in_scale / in_shift are bn->scale / bn->shift, not yet computed - the kernel finalizes the BatchNorm in its
// prologue, pgv_bn_src; 0 is returned, nothing launched, when the form that would serve the call cannot)
int pgv_conv_down_v2(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* small_out, double* stats,
                     const pgv_bwd_fuse* fuse, hipStream_t st, const pgv_bn_src* bn = nullptr);

int pgv_conv_up_v2(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                   const float* w, const float* bias, int act, float slope, float* big_out, double* stats,
                   const pgv_bwd_fuse* fuse, hipStream_t st, const pgv_bn_src* bn = nullptr);

// Third-generation kernels of the same layers (conv_c1_ring.hip): LDS ring of image rows filled by LDS-DMA, fp32 only.
// (sq: also the squared-error criterion against sq->target with an upstream gradient of 1, pgv_conv_up_sqerr)
struct pgv_ring_sq {
  const float* target;   // [B, 1, H, W]
  float k, scale;        // 2 * scale; the criterion's scale
  float* gy;             // [B, 1, H, W] gradient of the block's pre-activation output
  float* gbias;          // [1] += sum gy
  float* loss_acc;       // [1] += scale * sum (out - target)^2, or null
  float* cls;            // [PGV_CLS_COPIES][4] += sums of gy by (row parity, column parity), or null
};
int pgv_conv_up_ring(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                     const pgv_bn_src* bn, const pgv_ring_sq* sq = nullptr);
// Second-generation direct kernels (conv_direct2.hip): four pixels per lane, 16-byte LDS reads and stores.
int pgv_conv_up_direct2(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                        const float* w, const float* bias, int act, float slope, float* out, double* stats,
                        hipStream_t st, const pgv_bn_src* bn = nullptr);
int pgv_conv_down_direct2(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                          const float* w, const float* bias, int act, float slope, float* out, double* stats,
                          const pgv_bwd_fuse* fuse, hipStream_t st);

int64_t pgv_conv_wgrad_v2_workspace(const pgv_conv_desc* d);
// (req != null: returns 3 when the tap sums of the output gradient - req->scratch, as pgv_tap_replicas() partial copies -
// came out of the same launches)
int pgv_tap_replicas(int c_gy, int kk);
// pgv_bn_bwd_coef_from_gy with an explicit number of class-sum copies (pgv_coef_req.cls_copies)
int pgv_bn_bwd_coef_from_gy_cc(const pgv_conv_desc* d, int lower_is_big, const float* gy, const float* cls, int cls_copies,
                               double* T, const float* w, const float* gw, const float* scale, const float* shift,
                               const float* mean, const float* rstd, int64_t n, float* coef, float* ggamma, float* gbeta,
                               int flags, hipStream_t st);
int pgv_bn_bwd_coef_rep(const pgv_conv_desc* d, int lower_is_big, const float* w, const float* gw, const double* T, int trep,
                        const float* scale, const float* shift, const float* mean, const float* rstd, int64_t n,
                        float* coef, float* ggamma, float* gbeta, hipStream_t st);
int pgv_conv_wgrad_v2(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                      const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                      void* workspace, int64_t workspace_bytes, const pgv_coef_req* req, const pgv_bias_req* bias,
                      hipStream_t st);
// (bias != null: the reduce launch also sums the block's bias-gradient copies into their destination, pgv_bias_req)
int pgv_bias_finish(const pgv_bias_req* bias, hipStream_t st);   // the same as a launch of its own (other weight-gradient kernels)

// Direct vector-ALU kernels for the 1 <-> 8 channel 5x5 layers (conv_direct.hip): tried first.
int pgv_conv_down_direct(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                         const float* w, const float* bias, int act, float slope, float* out, double* stats,
                         hipStream_t st);
int pgv_conv_up_direct(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                       const float* w, const float* bias, int act, float slope, float* out, double* stats,
                       hipStream_t st);
int pgv_conv_wgrad_direct(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                          const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                          hipStream_t st);

// Gather-GEMM kernels for the deep (small-plane, many-channel) layers: same return convention as the tuned ones.
int pgv_conv_down_gemm(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                       const float* w, const float* bias, int act, float slope, float* out, double* stats,
                       hipStream_t st);
int pgv_conv_up_gemm(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* out, double* stats,
                     hipStream_t st);
int pgv_conv_wgrad_gemm(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                        const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                        hipStream_t st);

// Raw-plane implicit-GEMM kernels for the deep k4 layers (conv_deep.hip): tried before the gather-GEMM.
int pgv_conv_down_deep(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                       const float* w, const float* bias, int act, float slope, float* out, double* stats,
                       hipStream_t st, const pgv_bn_src* bn = nullptr);

int pgv_conv_up_deep(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* out, double* stats,
                     hipStream_t st, const pgv_bn_src* bn = nullptr);

int pgv_conv_wgrad_deep(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                        const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                        hipStream_t st);

int pgv_bn_stats_impl(const float* a, int B, int C, int HW, double* stats, hipStream_t st);
// g_y = act'(a) * (coef[c]*g + coef[C+c]*a + coef[2C+c]), gbias += sum g_y (no clearing): the separate-pass fallback of
// pgv_bwd_fuse for kernel families without the fused epilogue (in place: g_y == g)
int pgv_act_bwd_coef_impl(const float* g, const float* a, const float* coef, int B, int C, int HW, int act, float slope,
                          float* g_y, float* gbias, hipStream_t st);
// cls[C][4] += sums of gy[B,C,H,W] by (row parity, column parity): pgv_bwd_fuse.cls for kernels without that by-product
int pgv_class_sums2_impl(const float* gy, int B, int C, int H, int W, float* cls, hipStream_t st);

// bf16-native kernels of the deep layers (conv_deep_bf16.hip, conv_deep_wgrad_bf16.hip); weights come from pgv_conv_desc.w_shadow
// (conv_shadow.hip)
bool pgv_k1_bf16_shape(const pgv_conv_desc* d);
bool pgv_deep_bf16_shape(const pgv_conv_desc* d);
int64_t pgv_conv_weight_shadow_bytes_impl(const pgv_conv_desc* d);
int pgv_conv_weight_shadow_impl(const pgv_conv_desc* d, const float* w, void* shadow, hipStream_t st);
int pgv_conv_down_deep_bf16(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                            const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                            const pgv_bn_src* bn);
int pgv_conv_up_deep_bf16(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                          const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                          const pgv_bn_src* bn);
// PGV_COMPUTE_F32_SPLIT kernels of the deep layers (conv_deep_split.hip): fp32 products as six bf16 matrix instructions
bool pgv_deep_split_shape(const pgv_conv_desc* d);
bool pgv_k1_split_shape(const pgv_conv_desc* d);
int pgv_conv_down_deep_split(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                             const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                             const pgv_bn_src* bn);
int pgv_conv_up_deep_split(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                           const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                           const pgv_bn_src* bn);
// ... and of the large-plane layers of the 4-layer stack (conv_big_split.hip): one 512-thread workgroup per CU, persistent
// over (sample, band) units; per unit a matrix phase in which all eight waves multiply (weights in registers, in fragment
// order) and a vector phase in which they split and commit the next unit's band and move the output tile out
bool pgv_big_split_shape(const pgv_conv_desc* d);
bool pgv_big_bf16q_shape(const pgv_conv_desc* d);   // the same kernels with one operand plane: bf16 operand mode
int pgv_conv_down_big_split(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                            const float* bias, int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse,
                            hipStream_t st, const pgv_bn_src* bn);
int pgv_conv_up_big_split(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                          const float* bias, int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse,
                          hipStream_t st, const pgv_bn_src* bn);
// weight gradient of those layers (conv_wgrad_split.hip): per-workgroup partial gradients, same contract as the bf16 form
int pgv_conv_wgrad_split_partial(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                                 const float* small_in, const float* small_scale, const float* small_shift, float* partial,
                                 int64_t partial_bytes, int* nparts, hipStream_t st);
int64_t pgv_conv_wgrad_deep_bf16_workspace(const pgv_conv_desc* d);
int pgv_conv_wgrad_deep_bf16(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                             const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                             void* workspace, int64_t workspace_bytes, hipStream_t st);
int pgv_conv_up_big_bf16(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                         const float* bias, int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse,
                         hipStream_t st, const pgv_bn_src* bn);
int pgv_conv_down_big_bf16(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                           const float* bias, int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse,
                           hipStream_t st, const pgv_bn_src* bn);
int pgv_conv_weight_shadows_impl(int n, const pgv_conv_desc* const* descs, const float* const* ws, void* const* shadows,
                                 hipStream_t st);
