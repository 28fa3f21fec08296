// Weight shadows (pgv_conv_weight_shadow / pgv_conv_weight_shadows): the per-step copies of a layer's weight in the layout and
// precision its matrix kernels read - bf16 [cs][cb/8][taps][8] and its transposed-convolution twin for the bf16-native deep
// kernels (conv_deep_bf16.hip), [m][k] bf16 for the 1x1 layers, and the fragment-order planes (three for PGV_COMPUTE_F32_SPLIT,
// one for bf16 operand mode) of conv_deep_split.hip / conv_big_split.hip.  Split out of conv_deep_bf16.hip in round 6.
#include "conv_tile.h"
#include "conv_deep_common.h"

namespace {

typedef unsigned short u16;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight shadows.  down: D[cs][cb/8][kh*4+kw][8] (M = cs);  up: U[cb][cs/8][phase][th*2+tw][8] (M = cb), phase = 2ph+pw,
// taps kh = ph + 2th, kw = pw + 2tw.  One thread = (cs, channel group of 8 cb, kernel row): 8 x 16-byte reads, 4 x 16-byte
// writes of the down shadow; the up shadow is written by the thread that owns (cb, group of 8 cs, kernel row).
__device__ __forceinline__ void shadow_k4_item(int it, const float* __restrict__ w, int CS, int CB, u16* __restrict__ down,
                                               u16* __restrict__ up) {
  {
    const int kh = it & 3, g = (it >> 2) % (CB / 8), cs = (it >> 2) / (CB / 8);
    f32x4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const f32x4*>(w + ((size_t)(cs * CB + g * 8 + c) * 16 + kh * 4));
    u32x4* dst = reinterpret_cast<u32x4*>(down + ((size_t)(cs * (CB / 8) + g) * 16 + kh * 4) * 8);
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
      dst[kw] = u32x4{pack_bf16x2(v[0][kw], v[1][kw]), pack_bf16x2(v[2][kw], v[3][kw]), pack_bf16x2(v[4][kw], v[5][kw]),
                      pack_bf16x2(v[6][kw], v[7][kw])};
  }
  if (!up) return;
  {
    const int kh = it & 3, cb = (it >> 2) % CB, g = (it >> 2) / CB;   // cb fastest: the reads of a wave are 64-byte pieces
    f32x4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const f32x4*>(w + ((size_t)((g * 8 + c) * CB + cb) * 16 + kh * 4));
    const int ph = kh & 1, th = kh >> 1;
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
      const int pw = kw & 1, tw = kw >> 1;
      u32x4* dst = reinterpret_cast<u32x4*>(up + ((size_t)((cb * (CS / 8) + g) * 4 + 2 * ph + pw) * 4 + 2 * th + tw) * 8);
      *dst = u32x4{pack_bf16x2(v[0][kw], v[1][kw]), pack_bf16x2(v[2][kw], v[3][kw]), pack_bf16x2(v[4][kw], v[5][kw]),
                   pack_bf16x2(v[6][kw], v[7][kw])};
    }
  }
}
__global__ __launch_bounds__(256) void deep_shadow_kernel(const float* __restrict__ w, int CS, int CB,
                                                        u16* __restrict__ down, u16* __restrict__ up) {
  const int items = CS * (CB / 8) * 4;   // (= CB * (CS / 8) * 4: the same item count serves both layouts)
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) shadow_k4_item(it, w, CS, CB, down, up);
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 layers on 3x4 planes (enc8 / dec1: 512 <-> 2048 channels, model/encoder.py:64-69, model/decoder.py:72-75):
// out[b,m,p] = act(bias[m] + sum_k Wt[m][k] * in'[b,k,p]) - both directions are this one product (forward: m = cs, k = cb,
// Wt = the weight; transposed: m = cb, k = cs, Wt = its transpose), the shadow holds both as [m][k] bf16.  One workgroup =
// 128 output channels (16 per wave) x 4 samples (48 pixels = 3 tiles): every wave runs the whole K, no reduction.  Images
// are channel-innermost, 64 channels = 128 bytes per pixel, the 16-byte group g of a pixel stored at g ^ (pixel & 7).
__device__ __forceinline__ void shadow_k1_item(int it, const float* __restrict__ w, int CS, int CB, u16* __restrict__ down,
                                               u16* __restrict__ up) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)it * 8), b = *reinterpret_cast<const f32x4*>(w + (size_t)it * 8 + 4);
  *reinterpret_cast<u32x4*>(down + (size_t)it * 8) =
      u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  // transposed: up[cb][8 consecutive cs]; cb fastest across the lanes (coalesced reads of 8 weight rows)
  const int cb = it % CB, g = it / CB;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = w[(size_t)(g * 8 + c) * CB + cb];
  *reinterpret_cast<u32x4*>(up + (size_t)cb * CS + g * 8) =
      u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}
__global__ __launch_bounds__(256) void k1_shadow_kernel(const float* __restrict__ w, int CS, int CB, u16* __restrict__ down,
                                                        u16* __restrict__ up) {
  const int n8 = CS * CB / 8;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < n8; it += gridDim.x * 256) shadow_k1_item(it, w, CS, CB, down, up);
}

// the shadows of several layers in ONE launch (a conv stack's forward pass: 4 - 6 us of launch latency per layer otherwise)
struct ShadowTable {
  static constexpr int MAXN = 8;
  const float* w[MAXN];
  u16* down[MAXN];
  int CS[MAXN], CB[MAXN], k1[MAXN], items[MAXN], blk0[MAXN + 1];
  int n;
};
__global__ __launch_bounds__(256) void shadow_multi_kernel(ShadowTable t) {
  int e = 0;
  while (e + 1 < t.n && (int)blockIdx.x >= t.blk0[e + 1]) ++e;
  const int it = ((int)blockIdx.x - t.blk0[e]) * 256 + threadIdx.x;
  if (it >= t.items[e]) return;
  u16* up = t.down[e] + (size_t)t.CS[e] * t.CB[e] * (t.k1[e] ? 1 : 16);
  if (t.k1[e] == 3) {   // deep split: the down layout, then the up layout (both in fragment order, three planes each)
    const int nd = t.CS[e] * t.CB[e] * 2;
    if (it < nd)
      shadow_split_down_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e]);
    else
      shadow_split_up_item(it - nd, t.w[e], t.CS[e], t.CB[e], t.down[e] + (size_t)3 * t.CS[e] * t.CB[e] * 16);
  } else if (t.k1[e] == 5) {   // large-plane split (conv_big_split.hip): down fragments, then up fragments
    const int nd = t.CS[e] * t.CB[e] * 2;
    if (it < nd)
      shadow_bigq_down_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e]);
    else
      shadow_bigq_up_item(it - nd, t.w[e], t.CS[e], t.CB[e], t.down[e] + (size_t)3 * t.CS[e] * t.CB[e] * 16);
  } else if (t.k1[e] == 6) {   // bf16 operand mode on the large-plane kernels: the same fragment orders, one plane each
    const int nd = t.CS[e] * t.CB[e] * 2;
    if (it < nd)
      shadow_bigq_down_item<1>(it, t.w[e], t.CS[e], t.CB[e], t.down[e]);
    else
      shadow_bigq_up_item<1>(it - nd, t.w[e], t.CS[e], t.CB[e], t.down[e] + (size_t)t.CS[e] * t.CB[e] * 16);
  } else if (t.k1[e] == 4)
    shadow_split_k1_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e]);
  else if (t.k1[e])
    shadow_k1_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e], up);
  else
    shadow_k4_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e], up);
}

}  // namespace


// bytes of the bf16 weight shadow of a layer (down + up layouts), 0: the layer has no bf16-native kernels
int64_t pgv_conv_weight_shadow_bytes_impl(const pgv_conv_desc* d) {
  if (!(d->flags & PGV_COMPUTE_BF16)) {   // PGV_COMPUTE_F32_SPLIT: 3 bf16 planes; the deep layers hold a down and an up layout
    if (pgv_deep_split_shape(d)) return (int64_t)12 * d->Cs * d->Cb * 16;
    if (pgv_k1_split_shape(d)) return (int64_t)12 * d->Cs * d->Cb;
    return pgv_big_split_shape(d) ? (int64_t)12 * d->Cs * d->Cb * 16 : 0;
  }
  if (pgv_k1_bf16_shape(d)) return (int64_t)4 * d->Cs * d->Cb;
  return (pgv_deep_bf16_shape(d) || pgv_big_bf16q_shape(d)) ? (int64_t)4 * d->Cs * d->Cb * 16 : 0;
}

int pgv_conv_weight_shadow_impl(const pgv_conv_desc* d, const float* w, void* shadow, hipStream_t st) {
  if (!(d->flags & PGV_COMPUTE_BF16) || pgv_big_bf16q_shape(d)) {
    if (pgv_deep_split_shape(d) || pgv_k1_split_shape(d) || pgv_big_split_shape(d) || pgv_big_bf16q_shape(d)) {
      const pgv_conv_desc* one[1] = {d};
      const float* ws[1] = {w};
      void* sh[1] = {shadow};
      return pgv_conv_weight_shadows_impl(1, one, ws, sh, st);
    }
    return 0;
  }
  if (pgv_k1_bf16_shape(d)) {
    u16* down = (u16*)shadow;
    hipLaunchKernelGGL(k1_shadow_kernel, dim3((unsigned)min((d->Cs * d->Cb / 8 + 255) / 256, 2048)), dim3(256), 0, st, w, d->Cs,
                       d->Cb, down, down + (size_t)d->Cs * d->Cb);
    PGV_CHECK_LAUNCH("conv_weight_shadow");
    return 1;
  }
  if (!pgv_deep_bf16_shape(d)) return 0;
  u16* down = (u16*)shadow;
  u16* up = down + (size_t)d->Cs * d->Cb * 16;
  const int items = d->Cs * (d->Cb / 8) * 4;
  hipLaunchKernelGGL(deep_shadow_kernel, dim3((unsigned)min((items + 255) / 256, 2048)), dim3(256), 0, st, w, d->Cs, d->Cb,
                     down, up);
  PGV_CHECK_LAUNCH("conv_weight_shadow");
  return 1;
}

// 1 = launched, 0 = not this kernel family's case (no shadow in the descriptor, shape not covered)
int pgv_conv_weight_shadows_impl(int n, const pgv_conv_desc* const* descs, const float* const* ws, void* const* shadows,
                                 hipStream_t st) {
  ShadowTable t;
  t.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const pgv_conv_desc* d = descs[i];
    const bool bf = (d->flags & PGV_COMPUTE_BF16) != 0, split = pgv_big_split_shape(d), dsplit = pgv_deep_split_shape(d);
    const bool k1 = bf && pgv_k1_bf16_shape(d), k1split = pgv_k1_split_shape(d), bfq = pgv_big_bf16q_shape(d);
    if (!split && !dsplit && !k1split && !bfq && !(bf && (k1 || pgv_deep_bf16_shape(d)))) return 0;
    t.w[i] = ws[i];
    t.down[i] = (u16*)shadows[i];
    t.CS[i] = d->Cs, t.CB[i] = d->Cb;
    // kind: 0 k4 bf16, 1 1x1 bf16, 3 deep split down + up fragments, 4 split 1x1 fragments, 5 large-plane split fragments,
    // 6 large-plane fragments with one plane (bf16 operand mode)
    t.k1[i] = k1split ? 4 : dsplit ? 3 : split ? 5 : bfq ? 6 : (k1 ? 1 : 0);
    t.items[i] = k1split ? d->Cs * d->Cb / 4
                 : (dsplit || split || bfq) ? d->Cs * d->Cb * 4
                                     : (k1 ? d->Cs * d->Cb / 8 : d->Cs * (d->Cb / 8) * 4);
    t.blk0[i] = blocks;
    blocks += (t.items[i] + 255) / 256;
  }
  t.blk0[n] = blocks;
  hipLaunchKernelGGL(shadow_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t);
  PGV_CHECK_LAUNCH("conv_weight_shadows");
  return 1;
}

