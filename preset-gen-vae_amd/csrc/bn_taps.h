// Pieces of the pass-free BatchNorm backward shared by bn.hip (pgv_conv_tap_sums / pgv_bn_bwd_coef) and
// conv_v2_wgrad.hip (the fused reduce + tap sums + coefficients launch): tap / position predicates, the border-strip
// description worked out on the host, the border sums of one workgroup, the coefficient arithmetic.  DESIGN.md 3.5.
#pragma once
#include "pgv_common.h"

namespace {

// Does kernel tap k pair row (or column) r of gy with a position inside the other tensor (extent o_ext)?
//   gy small: the big position is r*s - p + k;   gy big: the small position is (r + p - k) / s when that divides.
__device__ __forceinline__ bool tap_hits(bool gy_is_big, int r, int k, int s, int p, int o_ext) {
  if (!gy_is_big) {
    const int rb = r * s - p + k;
    return rb >= 0 && rb < o_ext;
  }
  const int t = r + p - k;
  if (t < 0) return false;
  const int q = t / s;
  return q * s == t && q < o_ext;
}

// Is row (column) r of gy in the residue class that tap k can pair at all?  (gy big: r = q*s - p + k for an integer q;
// gy small: every row.)  cls index of a position = (r mod s) * s + (w mod s) for gy big, 0 for gy small.
__device__ __forceinline__ bool tap_in_class(bool gy_is_big, int r, int k, int s, int p) {
  if (!gy_is_big) return true;
  const int t = r + p - k;
  return ((t % s) + s) % s == 0;
}

// T[c][kh][kw] = cls[c][class of the tap] - sum of gy over the positions of that class the tap does NOT pair with a
// position inside the other tensor.  Those positions lie in a few border rows and columns (the partner index is monotone
// in the row / column: strips [0, ra) and [H - rb, H), likewise for columns): only they are read.  The strips and their
// per-tap masks are worked out on the host (TapBorder, a kernel argument).
// A thread owns border POSITIONS (fixed row / column) and sums them over the planes of its batch range with all loads in
// flight; the per-position sums go to LDS, and thread (tap, slice) adds up the positions of its slice that its tap cannot
// pair - no cross-lane reductions (K*K wave reductions through ds_bpermute cost more than everything else here).
constexpr int kTapStrip = 8;      // border rows / columns per side the fast form handles
constexpr int kTapChunk = 512;    // border positions per workgroup (grid.z walks the chunks)
struct TapBorder {
  int ra, rb, ca, cb;                                     // leading / trailing border rows and columns
  unsigned short rm[2 * kTapStrip], cm[2 * kTapStrip];   // low byte: in-class bits per tap, high byte: "in class, no partner"
};
// The border sums of one workgroup: channel c of gy, batch range by, position chunk bz -> res[tap] (LDS, K*K floats) =
// sum of gy over the unpaired border positions of this workgroup's share, valid in all threads after the call.
template <int K>
__device__ __forceinline__ void tap_border_block(const float* __restrict__ gy, int B, int C, int H, int W, int per,
                                                 int gy_is_big, int s, int p, const TapBorder& tb, int c, int by, int bz,
                                                 float* __restrict__ res) {
  __shared__ float psum[kTapChunk];
  __shared__ unsigned pmask[kTapChunk];   // row mask | column mask << 16 of the position
  __shared__ unsigned short rm[2 * kTapStrip], cm[2 * kTapStrip];
  constexpr int KK = K * K, NSL = 256 / KK;             // slices of positions per tap
  __shared__ float part[NSL][KK];
  const int tid = threadIdx.x;
  const int b0 = by * per, b1 = min(B, b0 + per);
  if (tid < 2 * kTapStrip) rm[tid] = tb.rm[tid], cm[tid] = tb.cm[tid];
  const int ra = tb.ra, ca = tb.ca, nBR = tb.ra + tb.rb, nNR = H - nBR, nBC = tb.ca + tb.cb;
  const int n1 = nBR * W, NE = n1 + nNR * nBC;
  const int nb = b1 - b0;
  const int64_t pstride = (int64_t)C * H * W;
  const int tap = tid % KK, slice = tid / KK, kh = tap / K, kw = tap - kh * K;
  // in-class bits of an interior row / column (not in a strip): every tap of the class finds its partner there
  unsigned ic_all = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) ic_all |= 1u << k;
  __syncthreads();
  float tacc = 0.f;   // this thread's tap over its slices of all chunks
  {
    const int base = bz * kTapChunk;
    const int n = min(kTapChunk, NE - base);
    for (int li = tid; li < n; li += 256) {
      const int i = base + li;
      int r, w;
      unsigned mr, mc;
      if (i < n1) {   // a border row, all columns
        const int ri = i / W;
        r = ri < ra ? ri : H - nBR + ri, w = i - ri * W;
        mr = rm[ri < ra ? ri : kTapStrip + (ri - ra)];
        const int wi = w < ca ? w : (w >= W - tb.cb ? kTapStrip + (w - (W - tb.cb)) : -1);
        // interior column: in class for the taps whose residue it has, never unpaired
        unsigned icc = 0;
        if (wi < 0) {
#pragma unroll
          for (int k = 0; k < K; ++k) icc |= tap_in_class(gy_is_big, w, k, s, p) ? 1u << k : 0u;
        }
        mc = wi >= 0 ? cm[wi] : icc;
      } else {        // a border column of an interior row
        const int j = i - n1, ri = j / nBC, ci = j - ri * nBC;
        r = ra + ri, w = ci < ca ? ci : W - nBC + ci;
        mc = cm[ci < ca ? ci : kTapStrip + (ci - ca)];
        unsigned icr = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) icr |= tap_in_class(gy_is_big, r, k, s, p) ? 1u << k : 0u;
        mr = icr;
      }
      const float* src = gy + ((int64_t)b0 * C + c) * H * W + (int64_t)r * W + w;
      float sv = 0.f;
      int b = 0;
      for (; b + 8 <= nb; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(b + u) * pstride];
#pragma unroll
        for (int u = 0; u < 8; ++u) sv += v[u];
      }
      for (; b < nb; ++b) sv += src[b * pstride];
      psum[li] = sv;
      pmask[li] = mr | (mc << 16);
    }
    __syncthreads();
    if (slice < NSL) {
      for (int li = slice; li < n; li += NSL) {
        const unsigned m = pmask[li];
        const unsigned icr = m & 255u, bdr = (m >> 8) & 255u, icc = (m >> 16) & 255u, bdc = m >> 24;
        const bool on = ((icr >> kh) & (icc >> kw) & ((bdr >> kh) | (bdc >> kw)) & 1u) != 0;
        tacc += on ? psum[li] : 0.f;
      }
    }
    __syncthreads();
  }
  (void)ic_all;
  if (slice < NSL) part[slice][tap] = tacc;
  __syncthreads();
  if (tid < KK) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) t += part[sl][tid];
    res[tid] = t;
  }
  __syncthreads();
}

// Coefficients of channel c of the lower block from S_o = sum w*gw and S_1 = sum w*T over the weight slice of c
// (block-wide: every thread of a 256-thread workgroup calls it; T_of(i) reads tap sum i).
struct CoefArgs {
  const float *w, *gw, *scale, *shift, *mean, *rstd;
  float *coef, *ggamma, *gbeta;
  double inv_n;
  int Cb, Cs, KK, lower_is_big, bf16, C;   // C = channels of the lower block
  int trep = 1;                            // the tap sums arrive as this many partial copies [trep][C_gy * KK] (tap_replicas)
};
// Border tap sums of few channels meet on few addresses (C_gy = 1: 384 workgroups on 25 doubles - 15 us of a launch that
// takes 5 otherwise): they are spread over this many copies of T, added up by the coefficient kernel.
static inline int tap_replicas(int c_gy, int kk) { return max(1, min(16, 1024 / (c_gy * kk))); }
// from S_o = sum g*o and S_1 = sum g of channel c (one thread)
__device__ __forceinline__ void bn_bwd_coef_finish(const CoefArgs& ca, int c, double so, double s1) {
  const double sc = ca.scale[c], sh = ca.shift[c], mu = ca.mean[c], rs = ca.rstd[c];
  // o = gamma*a_hat + beta:  sum g*a_hat = (sum g*o - beta * sum g) / gamma
  const double gamma = sc / rs, beta = sh + mu * sc;
  const double s2 = sc != 0.0 ? (so - beta * s1) / gamma : 0.0;
  ca.coef[c] = (float)sc;
  ca.coef[ca.C + c] = (float)(-sc * rs * s2 * ca.inv_n);
  ca.coef[2 * ca.C + c] = (float)(-sc * (s1 - mu * rs * s2) * ca.inv_n);
  if (ca.ggamma) ca.ggamma[c] = (float)s2;
  if (ca.gbeta) ca.gbeta[c] = (float)s1;
}
template <typename TF>
__device__ __forceinline__ void bn_bwd_coef_channel(const CoefArgs& ca, int c, TF T_of, double* red /* >= 16 doubles */) {
  const int n_other = ca.lower_is_big ? ca.Cs : ca.Cb;
  const int total = n_other * ca.KK;
  double so = 0.0, s1 = 0.0;
  // (one thread per element AND copy of the tap sums: the copies are independent loads, not a chain per thread)
  for (int e2 = threadIdx.x; e2 < total * ca.trep; e2 += 256) {
    const int r = e2 / total, e = e2 - r * total;
    const int co = e / ca.KK, tap = e - co * ca.KK;
    const int64_t idx = ca.lower_is_big ? ((int64_t)co * ca.Cb + c) * ca.KK + tap : ((int64_t)c * ca.Cb + co) * ca.KK + tap;
    const double wv = (double)pgv_opnd(ca.w[idx], ca.bf16 != 0);
    if (r == 0) so = fma(wv, (double)ca.gw[idx], so);
    s1 = fma(wv, T_of(e2), s1);
  }
  so = pgv_block_sum_d(so, red);
  s1 = pgv_block_sum_d(s1, red);
  if (threadIdx.x == 0) bn_bwd_coef_finish(ca, c, so, s1);
}

// host side of TapBorder: masks of one axis; false when a strip is longer than kTapStrip (the caller reads gy in full)
static bool host_tap_hits(bool gy_is_big, int r, int k, int s, int p, int o_ext) {
  if (!gy_is_big) {
    const int rb = r * s - p + k;
    return rb >= 0 && rb < o_ext;
  }
  const int t = r + p - k;
  if (t < 0) return false;
  const int q = t / s;
  return q * s == t && q < o_ext;
}
static bool host_tap_in_class(bool gy_is_big, int r, int k, int s, int p) {
  if (!gy_is_big) return true;
  const int t = r + p - k;
  return ((t % s) + s) % s == 0;
}
static bool tap_axis(bool gy_is_big, int n, int K, int s, int p, int o_ext, int* lead, int* trail, unsigned short* m) {
  auto mask = [&](int r) {
    unsigned ic = 0, bd = 0;
    for (int k = 0; k < K; ++k) {
      const bool in = host_tap_in_class(gy_is_big, r, k, s, p);
      ic |= in ? 1u << k : 0u;
      bd |= (in && !host_tap_hits(gy_is_big, r, k, s, p, o_ext)) ? 1u << k : 0u;
    }
    return (unsigned short)(ic | (bd << 8));
  };
  int a = 0;
  while (a < n && (mask(a) >> 8)) ++a;
  int b = 0;
  while (b < n - a && (mask(n - 1 - b) >> 8)) ++b;
  if (a > kTapStrip || b > kTapStrip) return false;
  for (int r = a; r < n - b; ++r)
    if (mask(r) >> 8) return false;   // (cannot happen: the partner index is monotone)
  for (int i = 0; i < 2 * kTapStrip; ++i) m[i] = 0;
  for (int i = 0; i < a; ++i) m[i] = mask(i);
  for (int i = 0; i < b; ++i) m[kTapStrip + i] = mask(n - b + i);
  *lead = a, *trail = b;
  return true;
}

}  // namespace
