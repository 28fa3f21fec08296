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
  const pgv_split_sel sel = pgv_split_sel_make();
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
          unsigned ph, pm, pl;
          pgv_split3_pair(fmaf(rb[i][0][e], s0, h0), fmaf(rb[i][1][e], s1, h1), ph, pm, pl, sel);
          *reinterpret_cast<unsigned*>(st + b_dst[i][e]) = ph;
          *reinterpret_cast<unsigned*>(st + G::IMG + b_dst[i][e]) = pm;
          *reinterpret_cast<unsigned*>(st + 2 * G::IMG + b_dst[i][e]) = pl;
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

// ---------------------------------------------------------------------------------------------------------------
// UP: out[b,cb,ih,iw] = act(bias[cb] + sum_{cs,kh,kw} w[cs,cb,kh,kw] * s'[b,cs,oh,ow]): four 2x2-tap convolutions, one per output
// phase (conv_deep_bf16.hip, UpB).  One workgroup = 32 big channels x NS samples; 8 waves = 4 phases x 2 M halves of 16 rows,
// every wave over the whole K and all pixel tiles of its phase: no weight element is shared between waves (fragments straight
// from the split shadow's up layout) and nothing to reduce.  A slab = 16 small channels = two K = 32 steps per phase.
template <int H_, int W_, int NS_>
struct UpS3 {
  static constexpr int H = H_, W = W_, NS = NS_;
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, HW = H * W;
  static constexpr int SWP = (H == 5 && W == 7) ? 5 : (H == 9 && W == 12) ? 11 : (H == 17 && W == 23) ? 14 : Ws + 1;
  static constexpr int SPLANE = (H == 5 && W == 7) ? 23 : (H == 9 && W == 12) ? 70 : (H == 17 && W == 23) ? 141 : (Hs + 1) * SWP;
  static_assert(SWP >= Ws + 1 && SPLANE >= (Hs + 1) * SWP, "padded plane");
  static constexpr int hu(int p) { return (p >> 1) ? H / 2 : (H + 1) / 2; }
  static constexpr int wu(int p) { return (p & 1) ? W / 2 : (W + 1) / 2; }
  static constexpr int TMAX = (NS * hu(0) * wu(0) + 15) / 16;   // tiles of a wave (phase 0 has the most pixels)
  static constexpr int MT = 32, KS = 2;
  static constexpr int IMG = NS * SPLANE * 16;             // one image: 8 channels of one plane
  static constexpr int STAGE = KS * 3 * IMG;               // [K step][plane]
  static constexpr int QUADS = (P + 3) / 4;
  static constexpr int ITEMS = NS * 4 * KS * QUADS;        // (sample, channel pair, pixel quad)
  static constexpr int QB = (ITEMS + 511) / 512;
  static constexpr int OUT_BYTES = NS * MT * HW * 4;
  static constexpr int WORK = (2 * STAGE > OUT_BYTES) ? 2 * STAGE : OUT_BYTES;
  static_assert(STAGE % 16 == 0 && P >= 4, "alignment");
};

template <class G>
__global__ __launch_bounds__(512) void deep_up_split_kernel(int B, int CB, int CS, const float* __restrict__ small_in,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            const u32x4* __restrict__ wsh, const float* __restrict__ bias,
                                                            int act, float slope, float* __restrict__ out,
                                                            double* __restrict__ stats, int groups, int stat_stride,
                                                            pgv_bn_src in_bn) {
  constexpr int HW = G::HW, P = G::P, NS = G::NS, TMAX = G::TMAX, MT = G::MT, KS = G::KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  float* aff = reinterpret_cast<float*>(ldsb + G::WORK);   // [2*CS]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const pgv_split_sel sel = pgv_split_sel_make();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // phases 0..3 have decreasing pixel counts: the two waves of a SIMD (w, w + 4) take phases p and 3 - p
  const int ph = wave < 4 ? wave : 7 - wave, half = wave >> 2;
  int mb, grp;
  deep_block(CB / MT, groups, mb, grp);
  const int cb0 = mb * MT, b0 = grp * NS;

  for (int i = tid; i < 2 * G::STAGE / 16; i += 512) reinterpret_cast<u32x4*>(ldsb)[i] = u32x4{0, 0, 0, 0};
  for (int i = tid; i < CS; i += 512) {
    float sc = 1.f, sh = 0.f;
    if (in_bn.stats)
      pgv_bn_finalize_dev(in_bn, CS, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[CS + i] = sh;
  }

  // ---- loader coordinates
  int b_src[G::QB], b_dst[G::QB][4], b_cp[G::QB];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 512 * i, G::ITEMS - 1);
    b_ok[i] = tid + 512 * i < G::ITEMS;
    const int si = q / (4 * KS * G::QUADS), rem = q - si * (4 * KS * G::QUADS), cp = rem / G::QUADS, qi = rem - cp * G::QUADS;
    const int p0 = min(4 * qi, P - 4);
    const int bs = min(b0 + si, B - 1);
    b_src[i] = (bs * CS + 2 * cp) * P + p0;
    b_cp[i] = cp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pe = p0 + e, oh = pe / G::Ws, ow = pe - oh * G::Ws;
      b_dst[i][e] = (cp >> 2) * 3 * G::IMG + (si * G::SPLANE + oh * G::SWP + ow) * 16 + (cp & 3) * 4;
    }
  }
  // ---- this wave's tiles: pixels [16 t, +16) of phase ph's list (sample, u, v)
  const int phh = ph >> 1, pww = ph & 1;
  const int hu = phh ? G::H / 2 : (G::H + 1) / 2, wu = pww ? G::W / 2 : (G::W + 1) / 2;
  const int cnt = NS * hu * wu, ntl = (cnt + 15) >> 4;
  const int th = kq >> 1, tw = kq & 1;
  const int ng = CS / 8;
  const u32x4* a_src = wsh + ((size_t)(mb * ng) * 8 + ph * 2 + half) * 192 + lane;   // + g * 8 * 192; + plane * 64
  int boff[TMAX], opix[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const int n = t * 16 + m, nn = min(n, cnt - 1);
    const int si = nn / (hu * wu), rem = nn - si * (hu * wu), u = rem / wu, v = rem - u * wu;
    boff[t] = (si * G::SPLANE + (u + 1 - th) * G::SWP + (v + 1 - tw)) * 16;
    opix[t] = (t < ntl && n < cnt) ? si * MT * HW + (2 * u + phh) * G::W + 2 * v + pww : -1;
  }
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[cb0 + half * 16 + 4 * kq + i] : 0.f;
  f32x4 acc[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[2][KS * 3];   // [set][K step * 3 + plane]
  f4u rb[G::QB][2];
  auto issue_a = [&](int set, int slab) {
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) ra[set][j * 3 + p] = a_src[(size_t)(slab * KS + j) * (8 * 192) + p * 64];
  };
  auto issue_b = [&](int slab) {
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const float* p = small_in + b_src[i] + slab * (8 * KS * P);
      rb[i][0] = *reinterpret_cast<const f4u*>(p);
      rb[i][1] = *reinterpret_cast<const f4u*>(p + P);
    }
  };
  auto commit = [&](unsigned char* st, int slab) {
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int c = slab * 8 * KS + 2 * b_cp[i];
      const float s0 = aff[c], s1 = aff[c + 1], h0 = aff[CS + c], h1 = aff[CS + c + 1];
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned ph, pm, pl;
          pgv_split3_pair(fmaf(rb[i][0][e], s0, h0), fmaf(rb[i][1][e], s1, h1), ph, pm, pl, sel);
          *reinterpret_cast<unsigned*>(st + b_dst[i][e]) = ph;
          *reinterpret_cast<unsigned*>(st + G::IMG + b_dst[i][e]) = pm;
          *reinterpret_cast<unsigned*>(st + 2 * G::IMG + b_dst[i][e]) = pl;
        }
      }
    }
  };
  auto kstep_products = [&](const u32x4 (&a)[KS * 3], int j, const unsigned char* st) {
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < ntl) {
        u32x4 bf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const u32x4*>(st + (j * 3 + p) * G::IMG + boff[t]);
        f32x4 c = acc[t];   // the six products, smallest first
        c = mfma_bf16_k32(a[j * 3 + 0], bf[2], c);
        c = mfma_bf16_k32(a[j * 3 + 2], bf[0], c);
        c = mfma_bf16_k32(a[j * 3 + 1], bf[1], c);
        c = mfma_bf16_k32(a[j * 3 + 0], bf[1], c);
        c = mfma_bf16_k32(a[j * 3 + 1], bf[0], c);
        acc[t] = mfma_bf16_k32(a[j * 3 + 0], bf[0], c);
      }
    }
  };

  const int nslab = CS / (8 * KS);
  issue_a(0, 0);
  issue_b(0);
  __syncthreads();
  commit(ldsb, 0);
  if (nslab > 1) issue_b(1);
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < nslab; s += 2) {   // (two slabs per trip: the register sets of the weight fragments alternate)
    {
      const unsigned char* st = ldsb;
      if (s + 1 < nslab) issue_a(1, s + 1);
      kstep_products(ra[0], 0, st);
      if (s + 1 < nslab) {   // the next slab goes to the other stage under the running matrix pipe
        commit(ldsb + G::STAGE, s + 1);
        if (s + 2 < nslab) issue_b(s + 2);
      }
      kstep_products(ra[0], 1, st);
      __syncthreads();
    }
    if (s + 1 < nslab) {
      const unsigned char* st = ldsb + G::STAGE;
      if (s + 2 < nslab) issue_a(0, s + 2);
      kstep_products(ra[1], 0, st);
      if (s + 2 < nslab) {
        commit(ldsb, s + 2);
        if (s + 3 < nslab) issue_b(s + 3);
      }
      kstep_products(ra[1], 1, st);
      __syncthreads();
    }
  }

  // ---- epilogue: bias, activation, into the [sample][channel][H*W] output tile (the stages are free)
  const pgv_act_params ap = pgv_act_setup(act, slope);
  float* otile = reinterpret_cast<float*>(ldsb);
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (opix[t] >= 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) otile[opix[t] + (half * 16 + 4 * kq + i) * HW] = pgv_act_apply(acc[t][i] + bv[i], ap);
    }
  }
  __syncthreads();
  if (stats) {   // 16 lanes per channel over the tile, one pair of atomics per channel
    stats += (size_t)(blockIdx.x & (PGV_CLS_COPIES - 1)) * stat_stride;
    const int ch = tid >> 4, part = tid & 15;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
    for (int si = 0; si < NS; ++si) {
      if (b0 + si < B)
        for (int i = part; i < HW; i += 16) {
          const float v = otile[(si * MT + ch) * HW + i];
          s1 += v;
          s2 += v * v;
        }
    }
    s1 = group16_sum(s1);
    s2 = group16_sum(s2);
    if (part == 0) {
      atomicAdd(&stats[cb0 + ch], (double)s1);
      atomicAdd(&stats[CB + cb0 + ch], (double)s2);
    }
  }
#pragma unroll
  for (int si = 0; si < NS; ++si) {
    if (b0 + si < B) {
      float* dst = out + ((int64_t)(b0 + si) * CB + cb0) * HW;
      const float* src = otile + si * MT * HW;
      for (int i = tid; i < MT * HW; i += 512) dst[i] = src[i];
    }
  }
}

template <int H, int W, int NS>
int launch_deep_up_split(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                         const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                         const pgv_bn_src* bn) {
  using G = UpS3<H, W, NS>;
  if (d->Cb % G::MT || d->Cs % (16 * G::KS) || !d->w_shadow) return 0;
  if ((int64_t)d->B * d->Cs * G::P * 4 >= (int64_t)1 << 31 || (int64_t)d->Cs * d->Cb * 96 >= (int64_t)1 << 31) return 0;
  const size_t bytes = (size_t)G::WORK + sizeof(float) * (2 * (size_t)d->Cs + 8);
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = deep_up_split_kernel<G>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_up_deep_split");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_deep_split: memset failed");
    return PGV_E_LAUNCH;
  }
  const int groups = (d->B + NS - 1) / NS;
  const u32x4* up = (const u32x4*)d->w_shadow + (size_t)d->Cs * d->Cb * 6;   // (after the down layout: 96 bytes per weight)
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * (d->Cb / G::MT))), dim3(512), bytes, st, d->B, d->Cb, d->Cs, small_in,
                     in_scale, in_shift, up, bias, act, slope, out, stats, groups,
                     (d->flags & PGV_STATS_COPIES) ? 2 * d->Cb : 0, bn ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("conv_up_deep_split");
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 layers on 3x4 planes (enc8 / dec1: 512 <-> 2048 channels): out[b,m,p] = act(bias[m] + sum_k Wt[m][k] * in'[b,k,p]).
// One workgroup = 128 output channels (16 per wave) x 8 samples (96 pixels = 6 tiles), every wave over the whole K: its
// weight fragments come straight from the split shadow (fragment order, one slab ahead), the activation slab (64 channels,
// 128 bytes per pixel, group g of a pixel at g ^ (pixel & 7), three planes) through LDS.
template <int NS_>
struct K1S3 {
  static constexpr int P = 12, NS = NS_, NPX = NS * P, NT = NPX / 16, MT = 128, CK = 64;
  static constexpr int IMG = NPX * 128, STAGE = 3 * IMG;
  static constexpr int ITEMS = NS * 32 * 3, QB = (ITEMS + 511) / 512;   // (sample, channel pair, pixel quad)
  static constexpr int OUT_BYTES = NS * MT * P * 4;
  static_assert(NPX % 16 == 0 && OUT_BYTES <= 2 * STAGE, "tile shapes");
};

template <class G>
__global__ __launch_bounds__(512) void k1_fwd_split_kernel(int B, int M, int K, const float* __restrict__ in,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift, const u32x4* __restrict__ wsh,
                                                           const float* __restrict__ bias, int act, float slope,
                                                           float* __restrict__ out, double* __restrict__ stats, int groups,
                                                           int stat_stride, pgv_bn_src in_bn) {
  constexpr int P = G::P, NS = G::NS, NT = G::NT, MT = G::MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  float* aff = reinterpret_cast<float*>(ldsb + 2 * G::STAGE);   // [2*K]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const pgv_split_sel sel = pgv_split_sel_make();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int mb, grp;
  deep_block(M / MT, groups, mb, grp);
  const int m0 = mb * MT, b0 = grp * NS;

  for (int i = tid; i < K; i += 512) {
    float sc = 1.f, sh = 0.f;
    if (in_bn.stats)
      pgv_bn_finalize_dev(in_bn, K, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[K + i] = sh;
  }
  int b_src[G::QB], b_dst[G::QB][4], b_cp[G::QB];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 512 * i, G::ITEMS - 1), si = q / 96, rem = q - si * 96, cp = rem / 3, qi = rem - cp * 3;
    b_ok[i] = tid + 512 * i < G::ITEMS;
    const int bs = min(b0 + si, B - 1);
    b_src[i] = (bs * K + 2 * cp) * P + 4 * qi;
    b_cp[i] = cp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int px = si * P + 4 * qi + e;
      b_dst[i][e] = px * 128 + (((cp >> 2) ^ (px & 7)) * 16) + (cp & 3) * 4;
    }
  }
  const int nks = K / 32;
  const u32x4* a_src = wsh + ((size_t)(m0 / 16 + wave) * nks) * 192 + lane;   // + K step * 192; + plane * 64
  int b_frag[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int px = t * 16 + m;
      b_frag[t][j] = px * 128 + (((j * 4 + kq) ^ (px & 7)) * 16);
    }
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[m0 + wave * 16 + 4 * kq + i] : 0.f;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[2][6];   // [set][K step * 3 + plane]
  f4u rb[G::QB][2];
  auto issue_a = [&](int set, int slab) {
#pragma unroll
    for (int i = 0; i < 6; ++i) ra[set][i] = a_src[(size_t)slab * 384 + i * 64];
  };
  auto issue_b = [&](int slab) {
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const float* p = in + b_src[i] + slab * (64 * P);
      rb[i][0] = *reinterpret_cast<const f4u*>(p);
      rb[i][1] = *reinterpret_cast<const f4u*>(p + P);
    }
  };
  auto commit = [&](unsigned char* st, int slab) {
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int c = slab * 64 + 2 * b_cp[i];
      const float s0 = aff[c], s1 = aff[c + 1], h0 = aff[K + c], h1 = aff[K + c + 1];
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned ph, pm, pl;
          pgv_split3_pair(fmaf(rb[i][0][e], s0, h0), fmaf(rb[i][1][e], s1, h1), ph, pm, pl, sel);
          *reinterpret_cast<unsigned*>(st + b_dst[i][e]) = ph;
          *reinterpret_cast<unsigned*>(st + G::IMG + b_dst[i][e]) = pm;
          *reinterpret_cast<unsigned*>(st + 2 * G::IMG + b_dst[i][e]) = pl;
        }
      }
    }
  };
  auto kstep_products = [&](const u32x4 (&a)[6], int j, const unsigned char* st) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      u32x4 bf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const u32x4*>(st + p * G::IMG + b_frag[t][j]);
      f32x4 c = acc[t];   // the six products, smallest first
      c = mfma_bf16_k32(a[j * 3 + 0], bf[2], c);
      c = mfma_bf16_k32(a[j * 3 + 2], bf[0], c);
      c = mfma_bf16_k32(a[j * 3 + 1], bf[1], c);
      c = mfma_bf16_k32(a[j * 3 + 0], bf[1], c);
      c = mfma_bf16_k32(a[j * 3 + 1], bf[0], c);
      acc[t] = mfma_bf16_k32(a[j * 3 + 0], bf[0], c);
    }
  };
  const int nslab = K / 64;
  issue_a(0, 0);
  issue_b(0);
  __syncthreads();   // affine staged
  commit(ldsb, 0);
  if (nslab > 1) issue_b(1);
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < nslab; s += 2) {
    {
      const unsigned char* st = ldsb;
      if (s + 1 < nslab) issue_a(1, s + 1);
      kstep_products(ra[0], 0, st);
      if (s + 1 < nslab) {
        commit(ldsb + G::STAGE, s + 1);
        if (s + 2 < nslab) issue_b(s + 2);
      }
      kstep_products(ra[0], 1, st);
      __syncthreads();
    }
    if (s + 1 < nslab) {
      const unsigned char* st = ldsb + G::STAGE;
      if (s + 2 < nslab) issue_a(0, s + 2);
      kstep_products(ra[1], 0, st);
      if (s + 2 < nslab) {
        commit(ldsb, s + 2);
        if (s + 3 < nslab) issue_b(s + 3);
      }
      kstep_products(ra[1], 1, st);
      __syncthreads();
    }
  }
  // ---- epilogue: bias, activation, into the [sample][channel][P] output tile
  const pgv_act_params ap = pgv_act_setup(act, slope);
  float* otile = reinterpret_cast<float*>(ldsb);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 16 + m, s2 = n / P, pix = n - s2 * P;
#pragma unroll
    for (int i = 0; i < 4; ++i) otile[(s2 * MT + wave * 16 + 4 * kq + i) * P + pix] = pgv_act_apply(acc[t][i] + bv[i], ap);
  }
  __syncthreads();
  if (stats) {   // 4 lanes per channel over the tile, one pair of atomics per channel
    stats += (size_t)(blockIdx.x & (PGV_CLS_COPIES - 1)) * stat_stride;
    const int ch = tid >> 2, part = tid & 3;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s3 = 0; s3 < NS; ++s3) {
      if (b0 + s3 < B) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float v = otile[(s3 * MT + ch) * P + part * 3 + i];
          s1 += v;
          s2 += v * v;
        }
      }
    }
    s1 += __shfl_xor(s1, 1);
    s2 += __shfl_xor(s2, 1);
    s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 2);
    if (part == 0) {
      atomicAdd(&stats[m0 + ch], (double)s1);
      atomicAdd(&stats[M + m0 + ch], (double)s2);
    }
  }
#pragma unroll
  for (int s3 = 0; s3 < NS; ++s3) {
    if (b0 + s3 < B) {
      float* dst = out + ((int64_t)(b0 + s3) * M + m0) * P;
      const float* src = otile + s3 * MT * P;
      for (int i = tid; i < MT * P; i += 512) dst[i] = src[i];
    }
  }
}

// up = false: out = small (m = cs, k = cb); up = true: out = big (m = cb, k = cs)
template <int NS>
int launch_k1_fwd_split_ns(const pgv_conv_desc* d, bool up, const float* in, const float* in_scale, const float* in_shift,
                        const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                        const pgv_bn_src* bn) {
  using G = K1S3<NS>;
  const int M = up ? d->Cb : d->Cs, K = up ? d->Cs : d->Cb;
  if ((int64_t)d->B * K * G::P * 4 >= (int64_t)1 << 31 || (int64_t)M * K * 12 >= (int64_t)1 << 31) return 0;
  const size_t bytes = 2 * (size_t)G::STAGE + sizeof(float) * (2 * (size_t)K + 8);
  if (bytes > (size_t)kMaxLds) return 0;
  static bool attr_done = false;
  int rc = raise_lds_limit(k1_fwd_split_kernel<G>, &attr_done, "conv_k1_split");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * M, st) != hipSuccess) {
    pgv_set_error("conv_k1_split: memset failed");
    return PGV_E_LAUNCH;
  }
  const int groups = (d->B + G::NS - 1) / G::NS;
  const u32x4* wsh = (const u32x4*)d->w_shadow + (up ? (size_t)d->Cs * d->Cb * 3 / 8 : 0);
  hipLaunchKernelGGL(k1_fwd_split_kernel<G>, dim3((unsigned)(groups * (M / G::MT))), dim3(512), bytes, st, d->B, M, K, in, in_scale,
                     in_shift, wsh, bias, act, slope, out, stats, groups, (d->flags & PGV_STATS_COPIES) ? 2 * M : 0,
                     bn ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("conv_k1_split");
  return 1;
}

// (8 samples per workgroup halve the weight stream; 4 when 128-row blocks x sample groups would not fill the CUs)
int launch_k1_fwd_split(const pgv_conv_desc* d, bool up, const float* in, const float* in_scale, const float* in_shift,
                        const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                        const pgv_bn_src* bn) {
  const int M = up ? d->Cb : d->Cs;
  if ((M / 128) * ((d->B + 7) / 8) >= 256) return launch_k1_fwd_split_ns<8>(d, up, in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  return launch_k1_fwd_split_ns<4>(d, up, in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
}

}  // namespace

bool pgv_deep_split_shape(const pgv_conv_desc* d) {
  return (d->flags & PGV_COMPUTE_F32_SPLIT) && !(d->flags & PGV_COMPUTE_BF16) && d->kh == 4 && d->kw == 4 && d->stride == 2 &&
         d->pad == 2 && d->Cb >= 64 && d->Cb % 32 == 0 && d->Cs % 64 == 0 &&
         ((d->Hb == 17 && d->Wb == 23) || (d->Hb == 9 && d->Wb == 12) || (d->Hb == 5 && d->Wb == 7));
}

bool pgv_k1_split_shape(const pgv_conv_desc* d) {
  return (d->flags & PGV_COMPUTE_F32_SPLIT) && !(d->flags & PGV_COMPUTE_BF16) && d->kh == 1 && d->kw == 1 && d->stride == 1 &&
         d->pad == 0 && d->Hb == 3 && d->Wb == 4 && d->Cb % 128 == 0 && d->Cs % 128 == 0;
}

// 1 = launched, 0 = not this kernel family's case
int pgv_conv_down_deep_split(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                             const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                             const pgv_bn_src* bn) {
  if (d->w_shadow && pgv_k1_split_shape(d)) return launch_k1_fwd_split(d, false, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (!d->w_shadow || !pgv_deep_split_shape(d)) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_down_split<17, 23, 1>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_down_split<9, 12, 4>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_down_split<5, 7, 8>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  return 0;
}

int pgv_conv_up_deep_split(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                           const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                           const pgv_bn_src* bn) {
  if (d->w_shadow && pgv_k1_split_shape(d)) return launch_k1_fwd_split(d, true, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (!d->w_shadow || !pgv_deep_split_shape(d)) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_up_split<17, 23, 2>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_up_split<9, 12, 4>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_up_split<5, 7, 8>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  return 0;
}
