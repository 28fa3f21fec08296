// PGV_COMPUTE_F32_SPLIT kernels of the deep k4 s2 p2 layers (17x23, 9x12 and 5x7 planes, 64..512 channels; model/encoder.py:249-255,
// model/decoder.py:205-210): an fp32 product as SIX bf16 matrix instructions.  An fp32 value is exactly the sum of three
// bfloat16 values, x = x1 + x2 + x3; the six largest cross terms of (w1 + w2 + w3)(x1 + x2 + x3), accumulated smallest first
// in the instruction's fp32 accumulator, carry the product to fp32 accuracy (scratch/ubench/bf16x6.hip: relative L2 error
// 7.2e-7 against 9.7e-7 for v_mfma_f32_16x16x4_f32 on K = 4096 dot products) at 6 / 16 of the fp32 instruction's matrix-pipe
// time.  The fp32-image kernels of conv_deep.hip are bound by that pipe (0.65 - 0.76 of its sustained peak), these are not.
//   * weights: a split SHADOW in fragment order (conv_deep_common.h, shadow_split_down_item), read from global memory straight
//     into registers, one slab ahead - with K split over the waves no two waves share a weight element, an LDS copy would
//     only move it twice;
//   * activations: split once, on the way into LDS, into three plane images of 16-byte pixels (8 channels = one slab); the B
//     fragment of output pixel n, kernel row kh, lane group kq = kernel column is the pixel at (2oh+kh, 2ow+kq) of a plane,
//     one ds_read_b128, conflict free with the plane strides of scratch/deep_split_strides.py.
// One workgroup = 64 output channels x NS samples; 8 waves = 2 halves of the channels x 4 kernel rows (K groups); LDS
// reduction rounds over the 4 K groups before the epilogue.
#include "conv_tile.h"
#include "conv_deep_common.h"

namespace {

typedef unsigned short u16;

template <int H_, int W_, int NS_>
struct DownS3 {
  static constexpr int H = H_, W = W_, NS = NS_;
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, HW = H * W, HP = 2 * Hs + 2;
  static constexpr int WP = (H == 5 && W == 7) ? 12 : (H == 9 && W == 12) ? 23 : (H == 17 && W == 23) ? 28 : 2 * Ws + 2;
  static constexpr int PLANE = (H == 5 && W == 7) ? 104 : (H == 9 && W == 12) ? 278 : HP * WP;
  static_assert(WP >= 2 * Ws + 2 && PLANE >= HP * WP, "padded plane");
  static constexpr int N = NS * P, NT = (N + 15) / 16;
  static constexpr int IMG = NS * PLANE * 16;              // bytes of one plane image (one of hi / mid / lo)
  static constexpr int STAGE = 3 * IMG;
  static constexpr int QUADS = (HW + 3) / 4;               // pixel quads of a plane (the last one shifted back)
  static constexpr int ITEMS = NS * 4 * QUADS;             // (sample, channel pair, quad)
  static constexpr int QB = (ITEMS + 511) / 512;
  static constexpr int RED_BYTES = 8 * NT * 1024, OUT_BYTES = NS * 64 * P * 4;
  static constexpr int WORK = (2 * STAGE > RED_BYTES + OUT_BYTES) ? 2 * STAGE : RED_BYTES + OUT_BYTES;
  static_assert(STAGE % 16 == 0 && HW >= 4, "alignment");
};

template <class G>
__global__ __launch_bounds__(512) void deep_down_split_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                              const float* __restrict__ in_scale,
                                                              const float* __restrict__ in_shift,
                                                              const u32x4* __restrict__ wsh, const float* __restrict__ bias,
                                                              int act, float slope, float* __restrict__ out,
                                                              double* __restrict__ stats, int groups, int stat_stride,
                                                              pgv_bn_src in_bn) {
  constexpr int NT = G::NT, HW = G::HW, P = G::P, NS = G::NS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  float* aff = reinterpret_cast<float*>(ldsb + G::WORK);   // [2*CB]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = wave >> 2, kh = wave & 3;
  int mb, grp;
  deep_block(CS / 64, groups, mb, grp);
  const int cs0 = mb * 64, b0 = grp * NS;

  // zero both stages' images once (the data pixels are rewritten every slab, the padding never)
  for (int i = tid; i < 2 * G::STAGE / 16; i += 512) reinterpret_cast<u32x4*>(ldsb)[i] = u32x4{0, 0, 0, 0};
  for (int i = tid; i < CB; i += 512) {
    float sc = 1.f, sh = 0.f;
    if (in_bn.stats)
      pgv_bn_finalize_dev(in_bn, CB, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[CB + i] = sh;
  }

  // ---- loader coordinates (identical for every slab)
  const int nslab = CB / 8;
  int b_src[G::QB], b_dst[G::QB][4], b_cp[G::QB];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 512 * i, G::ITEMS - 1);
    b_ok[i] = tid + 512 * i < G::ITEMS;
    const int si = q / (4 * G::QUADS), rem = q - si * (4 * G::QUADS), cp = rem / G::QUADS, qi = rem - cp * G::QUADS;
    const int p0 = min(4 * qi, HW - 4);
    const int bs = min(b0 + si, B - 1);   // partial last group: duplicate the last sample (masked at the store)
    b_src[i] = (bs * CB + 2 * cp) * HW + p0;
    b_cp[i] = cp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pe = p0 + e, r = pe / G::W, c = pe - r * G::W;
      b_dst[i][e] = (si * G::PLANE + (r + 2) * G::WP + c + 2) * 16 + cp * 4;
    }
  }
  // ---- fragment coordinates
  const u32x4* a_src = wsh + ((size_t)(mb * nslab) * 8 + wave) * 384 + lane;   // + slab * 8 * 384; + (plane * 2 + mt) * 64
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = min(t * 16 + m, G::N - 1);
    const int si = n / P, pix = n - si * P, oh = pix / G::Ws, ow = pix - oh * G::Ws;
    boff[t] = (si * G::PLANE + (2 * oh + kh) * G::WP + 2 * ow + kq) * 16;
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[2][6];   // [set][plane * 2 + mt]
  f4u rb[G::QB][2];
  auto issue_a = [&](int set, int slab) {
#pragma unroll
    for (int i = 0; i < 6; ++i) ra[set][i] = a_src[(size_t)slab * (8 * 384) + i * 64];
  };
  auto issue_b = [&](int slab) {
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const float* p = big + b_src[i] + slab * (8 * HW);
      rb[i][0] = *reinterpret_cast<const f4u*>(p);
      rb[i][1] = *reinterpret_cast<const f4u*>(p + HW);
    }
  };
  auto commit = [&](unsigned char* st, int slab) {
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int c = slab * 8 + 2 * b_cp[i];
      const float s0 = aff[c], s1 = aff[c + 1], h0 = aff[CB + c], h1 = aff[CB + c + 1];
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh, xm, xl, yh, ym, yl;
          pgv_split3(fmaf(rb[i][0][e], s0, h0), xh, xm, xl);
          pgv_split3(fmaf(rb[i][1][e], s1, h1), yh, ym, yl);
          *reinterpret_cast<unsigned*>(st + b_dst[i][e]) = pgv_pack_bf16x2(xh, yh);
          *reinterpret_cast<unsigned*>(st + G::IMG + b_dst[i][e]) = pgv_pack_bf16x2(xm, ym);
          *reinterpret_cast<unsigned*>(st + 2 * G::IMG + b_dst[i][e]) = pgv_pack_bf16x2(xl, yl);
        }
      }
    }
  };
  // the six products of a fragment pair, smallest first: (w1 x3, w3 x1, w2 x2), (w1 x2, w2 x1), w1 x1
  auto six = [&](const u32x4 (&a)[6], int mt, const u32x4 (&b)[3], f32x4 c) {
    c = mfma_bf16_k32(a[0 + mt], b[2], c);
    c = mfma_bf16_k32(a[4 + mt], b[0], c);
    c = mfma_bf16_k32(a[2 + mt], b[1], c);
    c = mfma_bf16_k32(a[0 + mt], b[1], c);
    c = mfma_bf16_k32(a[2 + mt], b[0], c);
    return mfma_bf16_k32(a[0 + mt], b[0], c);
  };
  auto slab_products = [&](const u32x4 (&a)[6], const unsigned char* st, int t0, int t1) {
#pragma unroll
    for (int t = t0; t < t1; ++t) {
      u32x4 bf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const u32x4*>(st + p * G::IMG + boff[t]);
      acc[0][t] = six(a, 0, bf, acc[0][t]);
      acc[1][t] = six(a, 1, bf, acc[1][t]);
    }
  };

  issue_a(0, 0);
  issue_b(0);
  __syncthreads();   // images zeroed, affine staged
  commit(ldsb, 0);
  if (nslab > 1) issue_b(1);
  __syncthreads();
  constexpr int TH = (NT + 1) / 2;
#pragma unroll 1
  for (int s = 0; s < nslab; s += 2) {   // (two slabs per trip: the register sets of the weight fragments alternate)
    {
      const unsigned char* st = ldsb;
      if (s + 1 < nslab) issue_a(1, s + 1);
      slab_products(ra[0], st, 0, TH);
      if (s + 1 < nslab) {   // the next slab goes to the other stage under the running matrix pipe
        commit(ldsb + G::STAGE, s + 1);
        if (s + 2 < nslab) issue_b(s + 2);
      }
      slab_products(ra[0], st, TH, NT);
      __syncthreads();
    }
    if (s + 1 < nslab) {
      const unsigned char* st = ldsb + G::STAGE;
      if (s + 2 < nslab) issue_a(0, s + 2);
      slab_products(ra[1], st, 0, TH);
      if (s + 2 < nslab) {
        commit(ldsb, s + 2);
        if (s + 3 < nslab) issue_b(s + 3);
      }
      slab_products(ra[1], st, TH, NT);
      __syncthreads();
    }
  }

  // ---- the 4 K groups' partial tiles are added up in two rounds (mt): every wave stores its 16 x N partial tile, the 2 NT
  // (half, N tile) sums are dealt over the 8 waves, summed in the fixed order of the K groups - deterministic - and
  // finished: bias, activation, into the [sample][channel][P] output tile; the stages are free after the last barrier
  const pgv_act_params ap = pgv_act_setup(act, slope);
  f32x4* red = reinterpret_cast<f32x4*>(ldsb);
  float* otile = reinterpret_cast<float*>(ldsb + G::RED_BYTES);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int t = 0; t < NT; ++t) red[(wave * NT + t) * 64 + lane] = acc[mt][t];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < (2 * NT + 7) / 8; ++jj) {
      const int j = wave + 8 * jj;
      if (j < 2 * NT) {
        const int hf = j / NT, t = j - hf * NT;
        f32x4 v = red[((hf * 4) * NT + t) * 64 + lane];
#pragma unroll
        for (int u = 1; u < 4; ++u) v += red[((hf * 4 + u) * NT + t) * 64 + lane];
        const int n = t * 16 + m, si = n / P, pix = n - si * P, cl = hf * 32 + mt * 16 + 4 * kq;
        if (n < G::N) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            otile[(si * 64 + cl + i) * P + pix] = pgv_act_apply(v[i] + (bias ? bias[cs0 + cl + i] : 0.f), ap);
        }
      }
    }
    __syncthreads();
  }
  // ---- BatchNorm statistics of the written outputs: 8 lanes per channel over the tile, one pair of atomics per channel
  if (stats) {
    stats += (size_t)(blockIdx.x & (PGV_CLS_COPIES - 1)) * stat_stride;   // PGV_STATS_COPIES: this XCD's partial copy
    const int ch = tid >> 3, part = tid & 7;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
    for (int si = 0; si < NS; ++si) {
      if (b0 + si < B)
        for (int i = part; i < P; i += 8) {
          const float v = otile[(si * 64 + ch) * P + i];
          s1 += v;
          s2 += v * v;
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    if (part == 0) {
      atomicAdd(&stats[cs0 + ch], (double)s1);
      atomicAdd(&stats[CS + cs0 + ch], (double)s2);
    }
  }
#pragma unroll
  for (int si = 0; si < NS; ++si) {
    if (b0 + si < B) {
      float* dst = out + ((int64_t)(b0 + si) * CS + cs0) * P;
      const float* src = otile + si * 64 * P;
      for (int i = tid; i < 64 * P; i += 512) dst[i] = src[i];
    }
  }
}

template <int H, int W, int NS>
int launch_deep_down_split(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                           const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                           const pgv_bn_src* bn) {
  using G = DownS3<H, W, NS>;
  if (d->Cs % 64 || d->Cb % 8 || !d->w_shadow) return 0;
  if ((int64_t)d->B * d->Cb * G::HW * 4 >= (int64_t)1 << 31 || (int64_t)d->Cs * d->Cb * 96 >= (int64_t)1 << 31) return 0;
  const size_t bytes = (size_t)G::WORK + sizeof(float) * (2 * (size_t)d->Cb + 8);
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = deep_down_split_kernel<G>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_down_deep_split");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_deep_split: memset failed");
    return PGV_E_LAUNCH;
  }
  const int groups = (d->B + NS - 1) / NS;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * (d->Cs / 64))), dim3(512), bytes, st, d->B, d->Cb, d->Cs, big, in_scale,
                     in_shift, (const u32x4*)d->w_shadow, bias, act, slope, out, stats, groups,
                     (d->flags & PGV_STATS_COPIES) ? 2 * d->Cs : 0, bn ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("conv_down_deep_split");
  return 1;
}

}  // namespace

bool pgv_deep_split_shape(const pgv_conv_desc* d) {
  return (d->flags & PGV_COMPUTE_F32_SPLIT) && !(d->flags & PGV_COMPUTE_BF16) && d->kh == 4 && d->kw == 4 && d->stride == 2 &&
         d->pad == 2 && d->Cb >= 64 && d->Cb % 16 == 0 && d->Cs % 64 == 0 &&
         ((d->Hb == 17 && d->Wb == 23) || (d->Hb == 9 && d->Wb == 12) || (d->Hb == 5 && d->Wb == 7));
}

// 1 = launched, 0 = not this kernel family's case
int pgv_conv_down_deep_split(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                             const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                             const pgv_bn_src* bn) {
  if (!d->w_shadow || !pgv_deep_split_shape(d)) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_down_split<17, 23, 1>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_down_split<9, 12, 4>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_down_split<5, 7, 8>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  return 0;
}
