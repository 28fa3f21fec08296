// BatchNorm pieces for [B,C,HW] fp32 tensors (nn.BatchNorm2d / BatchNorm1d, train mode; reference
// model/layer.py:21-26, model/encoder.py:86-87; backward per SURVEY Appendix B).
//
// All kernels use a (channel, batch-split) grid: per-channel constants are block-uniform, planes are read
// with lanes on consecutive hw (coalesced NCHW rows), and every per-channel sum is a wave-shuffle + LDS block
// reduction followed by ONE float atomic per block.  HBM-bound: one read (two for the backward pieces) and at
// most one write per element.
#include "conv_kernels.h"
#include "bn_taps.h"
#include "philox.h"

namespace {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at a 4-byte aligned address

struct Split {
  int nsplit;
  int per;  // batch items per split
};

inline Split pick_split(int B, int C, int HW) {
  // ~8 blocks per CU (2048) if the data allows; at least ~16K elements per block so the tail reduction and the
  // atomic are amortised.
  int64_t want = pgv_cdiv(2048, C);
  int64_t by_work = pgv_cdiv((int64_t)B * HW, 16384);
  int ns = (int)max((int64_t)1, min((int64_t)B, min(want, by_work)));
  Split s;
  s.per = (int)pgv_cdiv(B, ns);
  s.nsplit = (int)pgv_cdiv(B, s.per);
  return s;
}


// Visit every element of channel c in samples [b0, b1): f4(offset) for the 16-byte groups, f1(offset) for the tail of
// planes whose size is not a multiple of 4.  Large planes: lanes walk one plane at a time; small planes (the deep
// layers: 12 .. 391 pixels) are flattened over (sample, group) so that all 256 lanes stay busy.
template <typename F4, typename F1>
__device__ __forceinline__ void for_each_in_channel(int b0, int b1, int C, int c, int HW, F4 f4, F1 f1) {
  const int HW4 = HW >> 2, T = HW - (HW4 << 2);
  if (HW4 >= 256) {
    for (int b = b0; b < b1; ++b) {
      const int64_t base = ((int64_t)b * C + c) * HW;
      for (int i = threadIdx.x; i < HW4; i += blockDim.x) f4(base + 4 * i);
      for (int i = (HW4 << 2) + threadIdx.x; i < HW; i += blockDim.x) f1(base + i);
    }
    return;
  }
  const int nb = b1 - b0;
  if (HW4 > 0) {
    const float inv = 1.0f / (float)HW4;
    for (int e = threadIdx.x; e < nb * HW4; e += blockDim.x) {
      const int bi = (int)(((float)e + 0.5f) * inv), i = e - bi * HW4;  // exact for e < 2^20
      f4(((int64_t)(b0 + bi) * C + c) * HW + 4 * i);
    }
  }
  if (T > 0)
    for (int e = threadIdx.x; e < nb * T; e += blockDim.x) {
      const int bi = e / T, i = e - bi * T;
      f1(((int64_t)(b0 + bi) * C + c) * HW + (HW4 << 2) + i);
    }
}

// Two-input form with the loads of U 16-byte groups per lane issued before the first one is used (2 U loads in flight
// per lane: the one-group-at-a-time loop above leaves these streaming kernels at 45-60 % of the HBM rate).
template <int U, typename F4, typename F1>
__device__ __forceinline__ void for_each_in_channel2(int b0, int b1, int C, int c, int HW, const float* __restrict__ p,
                                                     const float* __restrict__ q, F4 f4, F1 f1) {
  const int HW4 = HW >> 2, T = HW - (HW4 << 2);
  const int nt = blockDim.x;
  if (HW4 >= 256) {
    for (int b = b0; b < b1; ++b) {
      const int64_t base = ((int64_t)b * C + c) * HW;
      for (int i0 = threadIdx.x; i0 < HW4; i0 += nt * U) {
        f4u pv[U], qv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * nt;
          if (i < HW4) {
            pv[u] = *reinterpret_cast<const f4u*>(p + base + 4 * i);
            qv[u] = *reinterpret_cast<const f4u*>(q + base + 4 * i);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * nt;
          if (i < HW4) f4(base + 4 * i, pv[u], qv[u]);
        }
      }
      for (int i = (HW4 << 2) + threadIdx.x; i < HW; i += nt) f1(base + i);
    }
    return;
  }
  const int nb = b1 - b0;
  if (HW4 > 0) {
    const float inv = 1.0f / (float)HW4;
    for (int e0 = threadIdx.x; e0 < nb * HW4; e0 += nt * U) {
      f4u pv[U], qv[U];
      int64_t off[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * nt;
        const int bi = (int)(((float)e + 0.5f) * inv), i = e - bi * HW4;  // exact for e < 2^20
        off[u] = ((int64_t)(b0 + bi) * C + c) * HW + 4 * i;
        if (e < nb * HW4) {
          pv[u] = *reinterpret_cast<const f4u*>(p + off[u]);
          qv[u] = *reinterpret_cast<const f4u*>(q + off[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e0 + u * nt < nb * HW4) f4(off[u], pv[u], qv[u]);
    }
  }
  if (T > 0)
    for (int e = threadIdx.x; e < nb * T; e += nt) {
      const int bi = e / T, i = e - bi * T;
      f1(((int64_t)(b0 + bi) * C + c) * HW + (HW4 << 2) + i);
    }
}

__global__ void bn_stats_kernel(const float* __restrict__ a, int B, int C, int HW, int per,
                                double* __restrict__ stats) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
  for_each_in_channel(
      b0, b1, C, c, HW,
      [&](int64_t off) {
        const f4u v = *reinterpret_cast<const f4u*>(a + off);
        const double v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
        s0 += v0 + v2;
        s1 += v1 + v3;
        q0 = fma(v0, v0, fma(v2, v2, q0));
        q1 = fma(v1, v1, fma(v3, v3, q1));
      },
      [&](int64_t off) {
        const double v0 = a[off];
        s0 += v0;
        q0 = fma(v0, v0, q0);
      });
  const double s = pgv_block_sum_d(s0 + s1, red);
  const double q = pgv_block_sum_d(q0 + q1, red);
  if (threadIdx.x == 0) {
    atomicAdd(&stats[c], s);
    atomicAdd(&stats[C + c], q);
  }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int copies, int C, double inv_n, double unbias,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  if (c >= C) return;
  double sum = stats[c], sq = stats[C + c];
  for (int r = 1; r < copies; ++r) sum += stats[r * 2 * C + c], sq += stats[r * 2 * C + C + c];   // (PGV_STATS_COPIES)
  const double mean = sum * inv_n;
  double var = sq * inv_n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
  if (scale) scale[c] = (float)(g * rstd);
  if (shift) shift[c] = (float)(bt - mean * g * rstd);
  if (mean_out) mean_out[c] = (float)mean;
  if (rstd_out) rstd_out[c] = (float)rstd;
  if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
  if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * unbias);
}

// nn.BatchNorm1d over x[B][C] (model/encoder.py:86-87), train mode, each direction in ONE launch: the tensor is tiny
// ([256][128]), the separate statistics / finalize / apply launches cost a dependent-launch latency each.
// A workgroup owns 16 consecutive channels, 16 row groups of threads walk the batch 8 rows at a time (all loads of a
// batch of rows in flight: the kernel is pure latency).
constexpr int kB1C = 16, kB1R = 16, kB1U = 8;
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, int B, int C,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float momentum, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var,
                                                       int64_t* __restrict__ num_batches_tracked, float* __restrict__ y,
                                                       float* __restrict__ scale_out, float* __restrict__ mean_out,
                                                       float* __restrict__ rstd_out) {
  __shared__ double ps[kB1R][kB1C], pq[kB1R][kB1C];
  __shared__ float sc_s[kB1C], sh_s[kB1C];
  const int cl = threadIdx.x % kB1C, rg = threadIdx.x / kB1C;
  const int c = blockIdx.x * kB1C + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int b0 = rg; b0 < B; b0 += kB1R * kB1U) {
      float v[kB1U];
#pragma unroll
      for (int u = 0; u < kB1U; ++u) {
        const int b = b0 + u * kB1R;
        v[u] = b < B ? x[(int64_t)b * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kB1U; ++u) {
        s += (double)v[u];
        q = fma((double)v[u], (double)v[u], q);
      }
    }
  ps[rg][cl] = s, pq[rg][cl] = q;
  __syncthreads();
  if (rg == 0 && c < C) {
#pragma unroll
    for (int k = 1; k < kB1R; ++k) s += ps[k][cl], q += pq[k][cl];
    const double inv_n = 1.0 / (double)B, mean = s * inv_n;
    double var = q * inv_n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
    const float sc = (float)(g * rstd), sh = (float)(bt - mean * g * rstd);
    sc_s[cl] = sc, sh_s[cl] = sh;
    if (scale_out) scale_out[c] = sc;
    if (mean_out) mean_out[c] = (float)mean;
    if (rstd_out) rstd_out[c] = (float)rstd;
    const double unbias = B > 1 ? (double)B / (double)(B - 1) : 1.0;
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * unbias);
  }
  __syncthreads();
  if (c < C) {
    const float sc = sc_s[cl], sh = sh_s[cl];
    for (int b0 = rg; b0 < B; b0 += kB1R * kB1U) {
      float v[kB1U];
#pragma unroll
      for (int u = 0; u < kB1U; ++u) {
        const int b = b0 + u * kB1R;
        v[u] = b < B ? x[(int64_t)b * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kB1U; ++u) {
        const int b = b0 + u * kB1R;
        if (b < B) y[(int64_t)b * C + c] = fmaf(v[u], sc, sh);
      }
    }
  }
}

__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                       const float* __restrict__ scale, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int B, int C,
                                                       float* __restrict__ gx, float* __restrict__ ggamma,
                                                       float* __restrict__ gbeta) {
  __shared__ double ps[kB1R][kB1C], pq[kB1R][kB1C];
  __shared__ float c1_s[kB1C], c2_s[kB1C];
  const int cl = threadIdx.x % kB1C, rg = threadIdx.x / kB1C;
  const int c = blockIdx.x * kB1C + cl;
  const float mu = c < C ? mean[c] : 0.f, rs = c < C ? rstd[c] : 0.f, sc = c < C ? scale[c] : 0.f;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int b0 = rg; b0 < B; b0 += kB1R * kB1U) {
      float gv[kB1U], xv[kB1U];
#pragma unroll
      for (int u = 0; u < kB1U; ++u) {
        const int b = b0 + u * kB1R;
        gv[u] = b < B ? g[(int64_t)b * C + c] : 0.f;
        xv[u] = b < B ? x[(int64_t)b * C + c] : mu;
      }
#pragma unroll
      for (int u = 0; u < kB1U; ++u) {
        s += (double)gv[u];
        q += (double)(gv[u] * ((xv[u] - mu) * rs));
      }
    }
  ps[rg][cl] = s, pq[rg][cl] = q;
  __syncthreads();
  if (rg == 0 && c < C) {
#pragma unroll
    for (int k = 1; k < kB1R; ++k) s += ps[k][cl], q += pq[k][cl];
    if (ggamma) ggamma[c] = (float)q;
    if (gbeta) gbeta[c] = (float)s;
    c1_s[cl] = (float)(s / (double)B), c2_s[cl] = (float)(q / (double)B);
  }
  __syncthreads();
  if (c < C) {
    const float c1 = c1_s[cl], c2 = c2_s[cl];
    for (int b0 = rg; b0 < B; b0 += kB1R * kB1U) {
      float gv[kB1U], xv[kB1U];
#pragma unroll
      for (int u = 0; u < kB1U; ++u) {
        const int b = b0 + u * kB1R;
        gv[u] = b < B ? g[(int64_t)b * C + c] : 0.f;
        xv[u] = b < B ? x[(int64_t)b * C + c] : mu;
      }
#pragma unroll
      for (int u = 0; u < kB1U; ++u) {
        const int b = b0 + u * kB1R;
        if (b < B) gx[(int64_t)b * C + c] = sc * (gv[u] - c1 - (xv[u] - mu) * rs * c2);
      }
    }
  }
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double rstd = 1.0 / sqrt((double)rv[c] + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
  scale[c] = (float)(g * rstd);
  shift[c] = (float)(bt - (double)rm[c] * g * rstd);
}

__global__ void affine_kernel(const float* __restrict__ a, const float* __restrict__ scale,
                              const float* __restrict__ shift, int B, int C, int HW, int per,
                              float* __restrict__ o) {
  const int c = blockIdx.x;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const float sc = scale[c], sh = shift[c];
  for_each_in_channel(
      b0, b1, C, c, HW,
      [&](int64_t off) {
        const f4u v = *reinterpret_cast<const f4u*>(a + off);
        f4u r;
        r.x = fmaf(v.x, sc, sh);
        r.y = fmaf(v.y, sc, sh);
        r.z = fmaf(v.z, sc, sh);
        r.w = fmaf(v.w, sc, sh);
        *reinterpret_cast<f4u*>(o + off) = r;
      },
      [&](int64_t off) { o[off] = fmaf(a[off], sc, sh); });
}

__global__ void bn_bwd_reduce_kernel(const float* __restrict__ g_o, const float* __restrict__ a,
                                     const float* __restrict__ mean, const float* __restrict__ rstd, int B, int C,
                                     int HW, int per, double* __restrict__ red_out) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const float mu = mean[c], rs = rstd[c];
  double s0 = 0.0, d0 = 0.0;
  // 16 bytes per lane (planes are only 4-byte aligned in NCHW with odd H*W): float partials per quad, double across
  for_each_in_channel2<4>(
      b0, b1, C, c, HW, g_o, a,
      [&](int64_t, const f4u& g, const f4u& v) {
        const float h0 = (v.x - mu) * rs, h1 = (v.y - mu) * rs, h2 = (v.z - mu) * rs, h3 = (v.w - mu) * rs;
        s0 += (double)((g.x + g.y) + (g.z + g.w));
        d0 += (double)fmaf(g.x, h0, fmaf(g.y, h1, fmaf(g.z, h2, g.w * h3)));
      },
      [&](int64_t off) {
        const float g0 = g_o[off];
        s0 += (double)g0;
        d0 += (double)(g0 * ((a[off] - mu) * rs));
      });
  const double s = pgv_block_sum_d(s0, red);
  const double dd = pgv_block_sum_d(d0, red);
  if (threadIdx.x == 0) {
    atomicAdd(&red_out[c], s);
    atomicAdd(&red_out[C + c], dd);
  }
}

// bn_bwd_reduce_kernel over g = (Dropout backward of g_d): the mask is regenerated here (pgv_dropout_bwd), g is written
// to gx on the way - the Dropout backward pass behind the encoder's Linear and the reduce pass of the top conv block's
// BatchNorm backward as ONE pass over the gradient (read g_d, a; write gx) instead of two (read g_d, write gx; read gx, a).
// The 16-byte groups of a plane whose size is not a multiple of 4 start at any element index: a group then takes its
// masks from two Philox blocks.
__global__ void dropout_bwd_bn_reduce_kernel(const uint64_t* __restrict__ saved, uint64_t stream_id, float p,
                                             float keep_scale, const float* __restrict__ g_d,
                                             const float* __restrict__ a, const float* __restrict__ mean,
                                             const float* __restrict__ rstd, int B, int C, int HW, int per,
                                             float* __restrict__ gx, double* __restrict__ red_out) {
  __shared__ double red[16];
  const uint64_t seed = saved[0], off0 = saved[1];
  const int c = blockIdx.x;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const float mu = mean[c], rs = rstd[c];
  double s0 = 0.0, d0 = 0.0;
  for_each_in_channel2<4>(
      b0, b1, C, c, HW, g_d, a,
      [&](int64_t off, const f4u& gd, const f4u& v) {
        const int r = (int)(off & 3);
        float m0[4], m1[4];
        dropout_mask4(off0, (uint64_t)(off >> 2), stream_id, seed, p, keep_scale, m0);
        if (r) dropout_mask4(off0, (uint64_t)(off >> 2) + 1, stream_id, seed, p, keep_scale, m1);
        float m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // element off + k: component (r + k) & 3 of block (r + k) >> 2
          const int e = r + k;
          const float lo = (e & 3) == 0 ? m0[0] : ((e & 3) == 1 ? m0[1] : ((e & 3) == 2 ? m0[2] : m0[3]));
          const float hi = (e & 3) == 0 ? m1[0] : ((e & 3) == 1 ? m1[1] : ((e & 3) == 2 ? m1[2] : m1[3]));
          m[k] = (r && e >= 4) ? hi : lo;
        }
        f4u g;
        g.x = gd.x * m[0], g.y = gd.y * m[1], g.z = gd.z * m[2], g.w = gd.w * m[3];
        *reinterpret_cast<f4u*>(gx + off) = g;
        const float h0 = (v.x - mu) * rs, h1 = (v.y - mu) * rs, h2 = (v.z - mu) * rs, h3 = (v.w - mu) * rs;
        s0 += (double)((g.x + g.y) + (g.z + g.w));
        d0 += (double)fmaf(g.x, h0, fmaf(g.y, h1, fmaf(g.z, h2, g.w * h3)));
      },
      [&](int64_t off) {
        float m[4];
        dropout_mask4(off0, (uint64_t)(off >> 2), stream_id, seed, p, keep_scale, m);
        const int k = (int)(off & 3);
        const float g0 = g_d[off] * (k == 0 ? m[0] : (k == 1 ? m[1] : (k == 2 ? m[2] : m[3])));
        gx[off] = g0;
        s0 += (double)g0;
        d0 += (double)(g0 * ((a[off] - mu) * rs));
      });
  const double s = pgv_block_sum_d(s0, red);
  const double dd = pgv_block_sum_d(d0, red);
  if (threadIdx.x == 0) {
    atomicAdd(&red_out[c], s);
    atomicAdd(&red_out[C + c], dd);
  }
}

template <int ACT, bool HAS_BN>
__global__ void act_bn_bwd_kernel(const float* __restrict__ g_o, const float* __restrict__ a,
                                  const float* __restrict__ scale, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const double* __restrict__ redv, double inv_n, int B,
                                  int C, int HW, int per, int act, float slope, float* __restrict__ g_y,
                                  float* __restrict__ gbias, float* __restrict__ ggamma,
                                  float* __restrict__ gbeta) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  if (redv && blockIdx.y == 0 && threadIdx.x == 0) {  // BatchNorm parameter gradients, float32 copies of the sums
    if (ggamma) ggamma[c] = (float)redv[C + c];
    if (gbeta) gbeta[c] = (float)redv[c];
  }
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  constexpr bool has_bn = HAS_BN;
  float sc = 1.f, mu = 0.f, rs = 1.f, c1 = 0.f, c2 = 0.f;
  if (has_bn) {
    sc = scale[c];
    if (redv) {
      mu = mean[c];
      rs = rstd[c];
      c1 = (float)(redv[c] * inv_n);
      c2 = (float)(redv[C + c] * inv_n);
    }
  }
  float acc = 0.f;
  auto one = [&](float g, float av) -> float {
    if (has_bn) g = sc * (g - c1 - (av - mu) * rs * c2);
    if (ACT == PGV_ACT_LEAKY_RELU)
      g = av > 0.f ? g : slope * g;
    else if (ACT == PGV_ACT_HARDTANH)
      g = (av > -1.f && av < 1.f) ? g : 0.f;
    return g;
  };
  for_each_in_channel2<4>(
      b0, b1, C, c, HW, g_o, a,
      [&](int64_t off, const f4u& g, const f4u& v) {
        f4u r;
        r.x = one(g.x, v.x);
        r.y = one(g.y, v.y);
        r.z = one(g.z, v.z);
        r.w = one(g.w, v.w);
        *reinterpret_cast<f4u*>(g_y + off) = r;
        acc += (r.x + r.y) + (r.z + r.w);
      },
      [&](int64_t off) {
        const float r = one(g_o[off], a[off]);
        g_y[off] = r;
        acc += r;
      });
  if (gbias) {
    const float s = pgv_block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(&gbias[c], s);
  }
}

// ---- BatchNorm backward without a pass over the gradient (pgv_bn_bwd_coef / pgv_conv_tap_sums / pgv_act_bwd_coef) -----
// T[c][kh][kw] += sum over (b in this block's batch range, r, w) of gy[b,c,r,w] * hit(kh, r) * hit(kw, w).
// A thread keeps a fixed column (W >= 256: columns tid + 256 j; narrower planes: 256 / W rows per pass, the thread's
// row advances): per element K conditional adds (row masks), the column masks are applied once at the end.
template <int K, int NJ>
__global__ __launch_bounds__(256) void tap_sums_kernel(const float* __restrict__ gy, int B, int C, int H, int W, int per,
                                                       int gy_is_big, int s, int p, int oH, int oW,
                                                       double* __restrict__ T) {
  __shared__ unsigned rowm[1024];
  __shared__ float wsum[4][K * K];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  for (int r = tid; r < H; r += 256) {
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) m |= tap_hits(gy_is_big, r, k, s, p, oH) ? 1u << k : 0u;
    rowm[r] = m;
  }
  __syncthreads();
  const int RP = NJ == 1 ? max(1, 256 / W) : 1;  // rows per pass
  const int rs = NJ == 1 ? tid / W : 0;
  const int w0 = NJ == 1 ? tid - rs * W : tid;
  const bool active = NJ == 1 ? tid < RP * W : true;
  unsigned cm[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int w = w0 + 256 * j;
    cm[j] = 0;
    if (active && w < W)
#pragma unroll
      for (int k = 0; k < K; ++k) cm[j] |= tap_hits(gy_is_big, w, k, s, p, oW) ? 1u << k : 0u;
  }
  float acc[NJ][K];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[j][k] = 0.f;
  if (active) {
    constexpr int U = 4;  // rows in flight per thread
    for (int b = b0; b < b1; ++b) {
      const float* pl = gy + ((int64_t)b * C + c) * H * W;
      for (int r0 = rs; r0 < H; r0 += RP * U) {
        float v[U][NJ];
        unsigned rm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = r0 + u * RP;
          rm[u] = r < H ? rowm[r] : 0u;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int w = w0 + 256 * j;
            v[u][j] = (r < H && w < W) ? pl[(int64_t)r * W + w] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[j][k] += (rm[u] >> k) & 1u ? v[u][j] : 0.f;
      }
    }
  }
  // T[kh][kw] = sum over threads and columns of colmask(kw) * acc[kh]
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) t += (cm[j] >> kw) & 1u ? acc[j][kh] : 0.f;
      t = pgv_wave_sum(t);
      if (lane == 0) wsum[wave][kh * K + kw] = t;
    }
  __syncthreads();
  if (tid < K * K)
    atomicAdd(&T[(int64_t)c * K * K + tid], (double)wsum[0][tid] + (double)wsum[1][tid] + (double)wsum[2][tid] + (double)wsum[3][tid]);
}

// cls[c][(r mod m) * m + (w mod m)] += sum of gy[:, c, r, w]  (m = M: 1 = plain channel sums).  Streaming pass, 16 bytes
// per lane, one float atomic per class per workgroup.
template <int M>
__global__ __launch_bounds__(256) void class_sums_kernel(const float* __restrict__ gy, int B, int C, int H, int W, int per,
                                                         float* __restrict__ cls) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const int nb = b1 - b0;
  float acc[M * M];
#pragma unroll
  for (int i = 0; i < M * M; ++i) acc[i] = 0.f;
  if (M == 1) {
    const int HW = H * W, HW4 = HW >> 2;
    const float inv_hw4 = 1.0f / (float)max(HW4, 1);
    constexpr int U = 4;
    for (int e0 = threadIdx.x; e0 < nb * HW4; e0 += 256 * U) {
      f4u v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 256;
        const int bi = (int)(((float)e + 0.5f) * inv_hw4), i = e - bi * HW4;   // exact for e < 2^20
        v[u] = e < nb * HW4 ? *reinterpret_cast<const f4u*>(gy + ((int64_t)(b0 + bi) * C + c) * HW + 4 * i)
                            : f4u{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc[0] += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    const int T4 = HW - (HW4 << 2);
    for (int e = threadIdx.x; e < nb * T4; e += 256) {
      const int bi = e / T4;
      acc[0] += gy[((int64_t)(b0 + bi) * C + c) * HW + (HW4 << 2) + (e - bi * T4)];
    }
  } else {
    // rows in 16-byte pieces (4-byte aligned): a piece starts at a column that is a multiple of 4, so its elements' column
    // classes are compile-time; the row class selects the accumulator
    const int QW = W >> 2, TW = W - (QW << 2);
    const int rows = nb * H;
    const float inv_qw = 1.0f / (float)max(QW, 1);
    float part[M];   // this thread's sums by column class for the row in hand
    constexpr int U = 4;
    for (int e0 = threadIdx.x; e0 < rows * QW; e0 += 256 * U) {
      f4u v[U];
      int rr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 256;
        const int row = (int)(((float)e + 0.5f) * inv_qw), q = e - row * QW;   // exact for e < 2^20
        const int bi = row / H, r = row - bi * H;
        rr[u] = r % M;
        v[u] = e < rows * QW ? *reinterpret_cast<const f4u*>(gy + (((int64_t)(b0 + bi) * C + c) * H + r) * W + 4 * q)
                             : f4u{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < M; ++k) part[k] = 0.f;
        if (M == 2) {
          part[0] = vv[0] + vv[2], part[M > 1 ? 1 : 0] = vv[1] + vv[3];
        } else {
          // (columns 4q + j: class (4q + j) mod M depends on q - generic, slow path)
          const int e = e0 + u * 256;
          const int row = (int)(((float)e + 0.5f) * inv_qw), q = e - row * QW;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < M; ++k) part[k] += ((4 * q + j) % M) == k ? vv[j] : 0.f;
        }
#pragma unroll
        for (int rc = 0; rc < M; ++rc)
#pragma unroll
          for (int k = 0; k < M; ++k) acc[rc * M + k] += rr[u] == rc ? part[k] : 0.f;
      }
    }
    for (int e = threadIdx.x; e < rows * TW; e += 256) {   // the last W % 4 columns of every row
      const int row = e / TW, j = e - row * TW;
      const int bi = row / H, r = row - bi * H, w = (QW << 2) + j;
      const float v = gy[(((int64_t)(b0 + bi) * C + c) * H + r) * W + w];
#pragma unroll
      for (int i = 0; i < M * M; ++i) acc[i] += ((r % M) * M + (w % M)) == i ? v : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < M * M; ++i) {
    const float t = pgv_block_sum(acc[i], red);
    if (threadIdx.x == 0) atomicAdd(&cls[c * M * M + i], t);
  }
}

// T[c][kh][kw] = cls[c][class of the tap] - the unpaired border positions (bn_taps.h: tap_border_block)
template <int K>
__global__ __launch_bounds__(256) void tap_border_kernel(const float* __restrict__ gy, int B, int C, int H, int W, int per,
                                                         int gy_is_big, int s, int p, TapBorder tb,
                                                         const float* __restrict__ cls, int cls_copies,
                                                         double* __restrict__ T) {
  constexpr int KK = K * K;
  __shared__ float res[KK];
  const int c = blockIdx.x, tid = threadIdx.x;
  tap_border_block<K>(gy, B, C, H, W, per, gy_is_big, s, p, tb, c, blockIdx.y, blockIdx.z, res);
  if (tid < KK) {
    const int kh = tid / K, kw = tid - kh * K;
    double t = -(double)res[tid];
    if (blockIdx.y == 0 && blockIdx.z == 0) {   // the class total enters once per channel
      const int m = gy_is_big ? s : 1;
      const int rho = gy_is_big ? (((kh - p) % s) + s) % s : 0, kap = gy_is_big ? (((kw - p) % s) + s) % s : 0;
      const int ncopy = cls_copies;   // (partial copies by XCD of the producers: pgv_bwd_fuse.cls)
      for (int r = 0; r < ncopy; ++r) t += (double)cls[(r * C + c) * m * m + rho * m + kap];
    }
    atomicAdd(&T[(int64_t)c * KK + tid], t);
  }
}

// One workgroup per channel c of the lower block.
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(CoefArgs ca, const double* __restrict__ T) {
  __shared__ double red[16];
  bn_bwd_coef_channel(ca, blockIdx.x, [&](int i) { return T[i]; }, red);
}

// g_y = act'(a) * (A*g + Bc*a + Cc), gbias += sum g_y: the separate-pass form of pgv_bwd_fuse
template <int ACT>
__global__ void act_bwd_coef_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                    const float* __restrict__ coef, int B, int C, int HW, int per, float slope,
                                    float* __restrict__ g_y, float* __restrict__ gbias) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const float ka = coef[c], kb = coef[C + c], kc = coef[2 * C + c];
  float acc = 0.f;
  auto one = [&](float gv, float av) -> float {
    float t = fmaf(gv, ka, fmaf(av, kb, kc));
    if (ACT == PGV_ACT_LEAKY_RELU)
      t = av > 0.f ? t : slope * t;
    else if (ACT == PGV_ACT_HARDTANH)
      t = (av > -1.f && av < 1.f) ? t : 0.f;
    return t;
  };
  for_each_in_channel2<4>(
      b0, b1, C, c, HW, g, a,
      [&](int64_t off, const f4u& gv, const f4u& v) {
        f4u r;
        r.x = one(gv.x, v.x);
        r.y = one(gv.y, v.y);
        r.z = one(gv.z, v.z);
        r.w = one(gv.w, v.w);
        *reinterpret_cast<f4u*>(g_y + off) = r;
        acc += (r.x + r.y) + (r.z + r.w);
      },
      [&](int64_t off) {
        const float r = one(g[off], a[off]);
        g_y[off] = r;
        acc += r;
      });
  if (gbias) {
    const float s = pgv_block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(&gbias[c], s);
  }
}

// Output block of a decoder under a squared-error criterion, backward in one pass: g = 2 scale g_loss (a - x) is never
// written - it goes straight through the block's activation backward into g_y, with the bias gradient alongside
// (replaces pgv_sqerr_bwd + pgv_act_bn_bwd of a block without BatchNorm: 3 passes over the tensor instead of 6).
__global__ void sqerr_act_bwd_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                     const float* __restrict__ g_loss, float scale, int B, int C, int HW, int per,
                                     int act, float slope, float* __restrict__ g_y, float* __restrict__ gbias,
                                     float* __restrict__ loss_acc) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const float k = 2.0f * scale * g_loss[0];
  float acc = 0.f, sq = 0.f;
  auto one = [&](float av, float xv) -> float {
    const float d = av - xv;
    sq = fmaf(d, d, sq);
    float g = k * d;
    if (act == PGV_ACT_LEAKY_RELU)
      g = av > 0.f ? g : slope * g;
    else if (act == PGV_ACT_HARDTANH)
      g = (av > -1.f && av < 1.f) ? g : 0.f;
    return g;
  };
  for_each_in_channel2<4>(
      b0, b1, C, c, HW, a, x,
      [&](int64_t off, const f4u& av, const f4u& xv) {
        f4u r;
        r.x = one(av.x, xv.x);
        r.y = one(av.y, xv.y);
        r.z = one(av.z, xv.z);
        r.w = one(av.w, xv.w);
        *reinterpret_cast<f4u*>(g_y + off) = r;
        acc += (r.x + r.y) + (r.z + r.w);
      },
      [&](int64_t off) {
        const float r = one(a[off], x[off]);
        g_y[off] = r;
        acc += r;
      });
  if (gbias) {
    const float s = pgv_block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(&gbias[c], s);
  }
  if (loss_acc) {  // the criterion's value as a by-product: scale * sum (a - x)^2
    const float s = pgv_block_sum(sq, red);
    if (threadIdx.x == 0) atomicAdd(loss_acc, scale * s);
  }
}

// single-channel form of the above (the spectrogram output layer: C = 1): one flat grid-stride pass, so the launch is
// as wide as the tensor is long instead of one workgroup per batch split
__global__ void sqerr_act_bwd_flat_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                          const float* __restrict__ g_loss, float scale, int64_t n, int act,
                                          float slope, float* __restrict__ g_y, float* __restrict__ gbias,
                                          float* __restrict__ loss_acc) {
  __shared__ float red[16];
  const float k = 2.0f * scale * g_loss[0];
  float acc = 0.f, sq = 0.f;
  auto one = [&](float av, float xv) -> float {
    const float d = av - xv;
    sq = fmaf(d, d, sq);
    float g = k * d;
    if (act == PGV_ACT_LEAKY_RELU)
      g = av > 0.f ? g : slope * g;
    else if (act == PGV_ACT_HARDTANH)
      g = (av > -1.f && av < 1.f) ? g : 0.f;
    return g;
  };
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  constexpr int U = 4;  // loads of U groups in flight per lane
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
    f4u av[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        av[u] = *reinterpret_cast<const f4u*>(a + 4 * i);
        xv[u] = *reinterpret_cast<const f4u*>(x + 4 * i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        f4u r;
        r.x = one(av[u].x, xv[u].x);
        r.y = one(av[u].y, xv[u].y);
        r.z = one(av[u].z, xv[u].z);
        r.w = one(av[u].w, xv[u].w);
        *reinterpret_cast<f4u*>(g_y + 4 * i) = r;
        acc += (r.x + r.y) + (r.z + r.w);
      }
    }
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float r = one(a[i], x[i]);
    g_y[i] = r;
    acc += r;
  }
  if (gbias) {
    const float s = pgv_block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(&gbias[0], s);
  }
  if (loss_acc) {
    const float s = pgv_block_sum(sq, red);
    if (threadIdx.x == 0) atomicAdd(loss_acc, scale * s);
  }
}

// The single-channel form with the class sums of g_y as a by-product (pgv_sqerr_act_bwd_cls).  Same aligned 16-byte walk
// over the flat tensor as above (rows of odd width start at 4-byte aligned addresses: row-wise 16-byte accesses would be
// split by the memory pipeline and run at a third of the rate); a workgroup owns kSqChunk consecutive floats, works out
// the (row, column) of its first element once, and every element's position from its distance to it.
constexpr int kSqChunk = 16384;
__global__ __launch_bounds__(256) void sqerr_act_bwd_cls_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                               const float* __restrict__ g_loss, float scale, int64_t n,
                                                               int H, int W, int act, float slope,
                                                               float* __restrict__ g_y, float* __restrict__ gbias,
                                                               float* __restrict__ loss_acc, float* __restrict__ cls) {
  __shared__ float red[16];
  const float k = 2.0f * scale * g_loss[0];
  float sq = 0.f, c4[4] = {0.f, 0.f, 0.f, 0.f};   // [2 * row parity + column parity]
  auto one = [&](float av, float xv) -> float {
    const float d = av - xv;
    sq = fmaf(d, d, sq);
    float g = k * d;
    if (act == PGV_ACT_LEAKY_RELU)
      g = av > 0.f ? g : slope * g;
    else if (act == PGV_ACT_HARDTANH)
      g = (av > -1.f && av < 1.f) ? g : 0.f;
    return g;
  };
  const int64_t start = (int64_t)blockIdx.x * kSqChunk;
  const int len = (int)min((int64_t)kSqChunk, n - start);
  const int HW = H * W;
  const int pos0 = (int)(start % HW), r_start = pos0 / W, w_start = pos0 - r_start * W;
  const float inv_w = 1.0f / (float)W;
  auto add = [&](int rel, float v) {   // element start + rel (rel + W < 2^20; at most one plane boundary inside a chunk)
    const int t = w_start + rel;
    const int dr = (int)(((float)t + 0.5f) * inv_w), w = t - dr * W;
    int r = r_start + dr;
    r = r >= H ? r - H : r;
    const int kc = (r & 1) * 2 + (w & 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) c4[q] += kc == q ? v : 0.f;
  };
  const int len4 = len >> 2;
  constexpr int U = 4;
  for (int i0 = threadIdx.x; i0 < len4; i0 += 256 * U) {
    f4u av[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 256;
      if (i < len4) {
        av[u] = *reinterpret_cast<const f4u*>(a + start + 4 * i);
        xv[u] = *reinterpret_cast<const f4u*>(x + start + 4 * i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 256;
      if (i < len4) {
        f4u r;
        r.x = one(av[u].x, xv[u].x);
        r.y = one(av[u].y, xv[u].y);
        r.z = one(av[u].z, xv[u].z);
        r.w = one(av[u].w, xv[u].w);
        *reinterpret_cast<f4u*>(g_y + start + 4 * i) = r;
        // position of the first element, the other three advance the column (and wrap into the next row / plane)
        const int t = w_start + 4 * i;
        const int dr = (int)(((float)t + 0.5f) * inv_w);
        int w = t - dr * W, rr = r_start + dr;
        rr = rr >= H ? rr - H : rr;
        const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kc = (rr & 1) * 2 + (w & 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) c4[q] += kc == q ? rv[j] : 0.f;
          ++w;
          const bool wrap = w == W;
          w = wrap ? 0 : w;
          rr = wrap ? (rr + 1 == H ? 0 : rr + 1) : rr;
        }
      }
    }
  }
  for (int i = (len4 << 2) + threadIdx.x; i < len; i += 256) {   // (only the last chunk has a tail)
    const float r = one(a[start + i], x[start + i]);
    g_y[start + i] = r;
    add(i, r);
  }
  // the six sums of the workgroup in one pass (4 classes, their total = the bias gradient, the squared error)
  __shared__ float red6[4 * 6];
  const float v6[6] = {c4[0], c4[1], c4[2], c4[3], (c4[0] + c4[1]) + (c4[2] + c4[3]), sq};
  const float r = pgv_block_sums<6>(v6, red6);
  if (threadIdx.x < 4)
    atomicAdd(&cls[(blockIdx.x & (PGV_CLS_COPIES - 1)) * 4 + threadIdx.x], r);   // (the copy of this workgroup's XCD; C = 1)
  else if (threadIdx.x == 4) {
    if (gbias) atomicAdd(&gbias[0], r);
  } else if (threadIdx.x == 5) {
    if (loss_acc) atomicAdd(loss_acc, scale * r);
  }
}

__global__ void colsum_kernel(const float* __restrict__ x, int M, int N, int64_t ld, float* __restrict__ out) {
  // block: 64 columns x 4 row-groups; rows split over blockIdx.y.
  __shared__ float part[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
  float acc = 0.f;
  if (n < N)
    for (int m = m0 + rg; m < m1; m += 4) acc += x[(int64_t)m * ld + n];
  part[rg][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rg == 0 && n < N) atomicAdd(&out[n], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] +
                                               part[3][threadIdx.x]);
}

int zero_async(void* p, size_t bytes, hipStream_t st, const char* who) {
  hipError_t e = hipMemsetAsync(p, 0, bytes, st);
  if (e != hipSuccess) {
    pgv_set_error("%s: memset failed: %s", who, hipGetErrorString(e));
    return PGV_E_LAUNCH;
  }
  return PGV_OK;
}

// pgv_act_bn_bwd for a block WITHOUT BatchNorm on small planes (enc8 of the 8-layer stack: 2048 channels of 3x4): the
// channel-per-workgroup walk of act_bn_bwd_kernel reads 48-byte runs 98 KB apart there (42 us for three passes over 25 MB).
// Here the tensor is walked as it lies in memory - a workgroup per sample, 16 bytes per lane, consecutive lanes on
// consecutive addresses - and the bias gradient is collected per channel in LDS (float atomics), flushed once per workgroup.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ACT>
__global__ __launch_bounds__(512) void act_bwd_flat_kernel(const f32x4* __restrict__ g_o, const f32x4* __restrict__ a, int B,
                                                           int C, int HW4, float slope, f32x4* __restrict__ g_y,
                                                           float* __restrict__ gbias) {
  extern __shared__ float sums[];   // [C]
  const int tid = threadIdx.x;
  for (int i = tid; i < C; i += 512) sums[i] = 0.f;
  __syncthreads();
  const int per = C * HW4;   // 16-byte groups of a sample
  auto one = [&](float g, float av) -> float {
    if (ACT == PGV_ACT_LEAKY_RELU)
      g = av > 0.f ? g : slope * g;
    else if (ACT == PGV_ACT_HARDTANH)
      g = (av > -1.f && av < 1.f) ? g : 0.f;
    return g;
  };
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const size_t base = (size_t)b * per;
    for (int i = tid; i < per; i += 512) {
      const f32x4 g = g_o[base + i], v = a[base + i];
      const f32x4 r = {one(g[0], v[0]), one(g[1], v[1]), one(g[2], v[2]), one(g[3], v[3])};
      g_y[base + i] = r;
      if (gbias) atomicAdd(&sums[i / HW4], (r[0] + r[1]) + (r[2] + r[3]));
    }
  }
  __syncthreads();
  if (gbias)
    for (int i = tid; i < C; i += 512) atomicAdd(&gbias[i], sums[i]);
}

// pgv_bn_bwd_reduce + pgv_act_bn_bwd as ONE launch for small planes: one workgroup per channel holds the channel's gradient
// and activation in registers (16-byte groups of a plane, flattened over (sample, group): NR per thread, plus the planes'
// tails), sums them (float per thread, float64 across the workgroup, as the reduce pass does), then applies the backward to
// the values it holds - 3 tensor passes over HBM and one launch instead of 5 and two (the deep blocks of the 8-layer stack:
// 6-14 MB tensors, where the two launches cost more than their traffic).
template <int ACT, int NR>
__global__ __launch_bounds__(512) void bn_act_bwd_fused_kernel(const float* __restrict__ g_o, const float* __restrict__ a,
                                        const float* __restrict__ scale, const float* __restrict__ mean,
                                        const float* __restrict__ rstd, int B, int C, int HW, float slope,
                                        float* __restrict__ g_y, float* __restrict__ gbias, float* __restrict__ ggamma,
                                        float* __restrict__ gbeta) {
  __shared__ double redd[2][16];
  __shared__ double tot[2];
  __shared__ float redf[16];
  const int c = blockIdx.x, nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (nt + 63) >> 6;
  const int HW4 = HW >> 2, T = HW - 4 * HW4, groups = B * HW4, tails = B * T;
  const float mu = mean[c], rs = rstd[c], sc = scale[c];
  f4u gv[NR], av[NR];
  int off[NR];
  float gt[2] = {0.f, 0.f}, at[2] = {0.f, 0.f};
  int offt[2] = {-1, -1};
  double s0 = 0.0, d0 = 0.0;   // (float partials per 16-byte group, float64 from there on: as pgv_bn_bwd_reduce)
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int e = tid + j * nt;
    off[j] = -1;
    if (e < groups) {
      const int bi = e / HW4, i = e - bi * HW4;
      off[j] = (bi * C + c) * HW + 4 * i;
      gv[j] = *reinterpret_cast<const f4u*>(g_o + off[j]);
      av[j] = *reinterpret_cast<const f4u*>(a + off[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + j * nt;
    if (e < tails) {
      const int bi = e / T, i = e - bi * T;
      offt[j] = (bi * C + c) * HW + 4 * HW4 + i;
      gt[j] = g_o[offt[j]];
      at[j] = a[offt[j]];
    }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    if (off[j] >= 0) {
      const f4u g = gv[j], v = av[j];
      const float h0 = (v.x - mu) * rs, h1 = (v.y - mu) * rs, h2 = (v.z - mu) * rs, h3 = (v.w - mu) * rs;
      s0 += (double)((g.x + g.y) + (g.z + g.w));
      d0 += (double)fmaf(g.x, h0, fmaf(g.y, h1, fmaf(g.z, h2, g.w * h3)));
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (offt[j] >= 0) {
      s0 += (double)gt[j];
      d0 += (double)(gt[j] * ((at[j] - mu) * rs));
    }
  }
  const double ws = pgv_wave_sum_d(s0), wd = pgv_wave_sum_d(d0);
  if (lane == 0) redd[0][wave] = ws, redd[1][wave] = wd;
  __syncthreads();
  if (wave == 0) {
    double r0 = lane < nw ? redd[0][lane] : 0.0, r1 = lane < nw ? redd[1][lane] : 0.0;
    r0 = pgv_wave_sum_d(r0);
    r1 = pgv_wave_sum_d(r1);
    if (lane == 0) tot[0] = r0, tot[1] = r1;
  }
  __syncthreads();
  const double inv_n = 1.0 / ((double)B * HW);
  const float c1 = (float)(tot[0] * inv_n), c2 = (float)(tot[1] * inv_n);
  if (tid == 0) {
    if (ggamma) ggamma[c] = (float)tot[1];
    if (gbeta) gbeta[c] = (float)tot[0];
  }
  auto one = [&](float g, float av_) -> float {
    g = sc * (g - c1 - (av_ - mu) * rs * c2);
    if (ACT == PGV_ACT_LEAKY_RELU)
      g = av_ > 0.f ? g : slope * g;
    else if (ACT == PGV_ACT_HARDTANH)
      g = (av_ > -1.f && av_ < 1.f) ? g : 0.f;
    return g;
  };
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    if (off[j] >= 0) {
      f4u r;
      r.x = one(gv[j].x, av[j].x);
      r.y = one(gv[j].y, av[j].y);
      r.z = one(gv[j].z, av[j].z);
      r.w = one(gv[j].w, av[j].w);
      *reinterpret_cast<f4u*>(g_y + off[j]) = r;
      acc += (r.x + r.y) + (r.z + r.w);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (offt[j] >= 0) {
      const float r = one(gt[j], at[j]);
      g_y[offt[j]] = r;
      acc += r;
    }
  }
  if (gbias) {
    const float s = pgv_block_sum(acc, redf);
    if (tid == 0) atomicAdd(&gbias[c], s);
  }
}

// shapes the fused launch serves: a channel's values fit the registers of one workgroup and there are enough channels
static bool bn_act_bwd_fusable(int B, int C, int HW) {
  const int64_t groups = (int64_t)B * (HW >> 2), tails = (int64_t)B * (HW & 3);
  // (at most 4 groups per thread of 512: with 8 groups per thread of 1024 - the 9x12 planes of a 256-sample batch - the
  // kernel took 111 us against 20 us for the two passes: 128 workgroups, a 128-register budget and 16 loads per thread)
  return B > 0 && C >= 96 && groups > 0 && groups <= 2048 && tails <= 2 * 256 * (groups <= 768 ? 1 : 2) &&
         (int64_t)B * C * HW < ((int64_t)1 << 31);
}

}  // namespace

int pgv_bn_stats_impl(const float* a, int B, int C, int HW, double* stats, hipStream_t st) {
  int rc = zero_async(stats, sizeof(double) * 2 * C, st, "pgv_bn_stats");
  if (rc) return rc;
  if ((int64_t)B * HW == 0) return PGV_OK;
  Split s = pick_split(B, C, HW);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C, s.nsplit), dim3(256), 0, st, a, B, C, HW, s.per, stats);
  PGV_CHECK_LAUNCH("bn_stats");
  return PGV_OK;
}

int pgv_act_bwd_coef_impl(const float* g, const float* a, const float* coef, int B, int C, int HW, int act, float slope,
                          float* g_y, float* gbias, hipStream_t st) {
  if (B == 0) return PGV_OK;
  Split s = pick_split(B, C, HW);
  typedef void (*kern_t)(const float*, const float*, const float*, int, int, int, int, float, float*, float*);
  kern_t kern = act == PGV_ACT_LEAKY_RELU ? (kern_t)act_bwd_coef_kernel<PGV_ACT_LEAKY_RELU>
                                          : (act == PGV_ACT_HARDTANH ? (kern_t)act_bwd_coef_kernel<PGV_ACT_HARDTANH>
                                                                     : (kern_t)act_bwd_coef_kernel<PGV_ACT_NONE>);
  hipLaunchKernelGGL(kern, dim3(C, s.nsplit), dim3(256), 0, st, g, a, coef, B, C, HW, s.per, slope, g_y, gbias);
  PGV_CHECK_LAUNCH("act_bwd_coef");
  return PGV_OK;
}

// class sums of a tensor written by a kernel family without that by-product (accumulates into cls[C][4])
int pgv_class_sums2_impl(const float* gy, int B, int C, int H, int W, float* cls, hipStream_t st) {
  if (B == 0) return PGV_OK;
  Split sp = pick_split(B, C, H * W);
  hipLaunchKernelGGL(class_sums_kernel<2>, dim3(C, sp.nsplit), dim3(256), 0, st, gy, B, C, H, W, sp.per, cls);
  PGV_CHECK_LAUNCH("conv_class_sums");
  return PGV_OK;
}

extern "C" {

int pgv_bn_stats(const float* a, int B, int C, int HW, double* stats, void* stream) {
  PGV_CHECK_ARG(a && stats && B >= 0 && C > 0 && HW > 0, "pgv_bn_stats: bad argument");
  return pgv_bn_stats_impl(a, B, C, HW, stats, pgv_stream(stream));
}

int pgv_bn_finalize(const double* stats, int C, int64_t n, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* scale, float* shift, float* mean, float* rstd, void* stream) {
  PGV_CHECK_ARG(stats && C > 0 && n > 0, "pgv_bn_finalize: bad argument");
  // torch raises for n==1 in train mode ("Expected more than 1 value per channel"); the host mirrors that.
  const double unbias = n > 1 ? (double)n / (double)(n - 1) : 1.0;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)pgv_cdiv(C, 128)), dim3(128), 0, pgv_stream(stream), stats, 1,
                     C, 1.0 / (double)n, unbias, gamma, beta, eps, momentum, running_mean, running_var,
                     num_batches_tracked, scale, shift, mean, rstd);
  PGV_CHECK_LAUNCH("bn_finalize");
  return PGV_OK;
}

int pgv_bn_finalize_src(const pgv_bn_src* s, int C, void* stream) {
  PGV_CHECK_ARG(s && s->stats && s->scale && s->shift && C > 0 && s->n > 0, "pgv_bn_finalize_src: incomplete pgv_bn_src");
  const double unbias = s->n > 1 ? (double)s->n / (double)(s->n - 1) : 1.0;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)pgv_cdiv(C, 128)), dim3(128), 0, pgv_stream(stream), s->stats,
                     s->stats_copies > 1 ? s->stats_copies : 1, C, 1.0 / (double)s->n, unbias, s->gamma, s->beta, s->eps,
                     s->momentum, s->running_mean, s->running_var, s->num_batches_tracked, s->scale, s->shift, s->mean,
                     s->rstd);
  PGV_CHECK_LAUNCH("bn_finalize");
  return PGV_OK;
}

int pgv_bn1d_fwd(const float* x, int B, int C, const float* gamma, const float* beta, float eps, float momentum,
                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y, float* scale,
                 float* mean, float* rstd, void* stream) {
  PGV_CHECK_ARG(x && y && B > 0 && C > 0, "pgv_bn1d_fwd: bad argument");
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((unsigned)pgv_cdiv(C, kB1C)), dim3(256), 0, pgv_stream(stream), x, B, C, gamma,
                     beta, eps, momentum, running_mean, running_var, num_batches_tracked, y, scale, mean, rstd);
  PGV_CHECK_LAUNCH("bn1d_fwd");
  return PGV_OK;
}

int pgv_bn1d_bwd(const float* g, const float* x, const float* scale, const float* mean, const float* rstd, int B, int C,
                 float* gx, float* ggamma, float* gbeta, void* stream) {
  PGV_CHECK_ARG(g && x && scale && mean && rstd && gx && B > 0 && C > 0, "pgv_bn1d_bwd: bad argument");
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((unsigned)pgv_cdiv(C, kB1C)), dim3(256), 0, pgv_stream(stream), g, x, scale,
                     mean, rstd, B, C, gx, ggamma, gbeta);
  PGV_CHECK_LAUNCH("bn1d_bwd");
  return PGV_OK;
}

int pgv_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int C, float* scale, float* shift, void* stream) {
  PGV_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "pgv_bn_eval_affine: bad argument");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((unsigned)pgv_cdiv(C, 128)), dim3(128), 0, pgv_stream(stream), gamma,
                     beta, running_mean, running_var, eps, C, scale, shift);
  PGV_CHECK_LAUNCH("bn_eval_affine");
  return PGV_OK;
}

int pgv_affine_nchw(const float* a, const float* scale, const float* shift, int B, int C, int HW, float* o,
                    void* stream) {
  PGV_CHECK_ARG(a && scale && shift && o && B >= 0 && C > 0 && HW > 0, "pgv_affine_nchw: bad argument");
  if (B == 0) return PGV_OK;
  Split s = pick_split(B, C, HW);
  hipLaunchKernelGGL(affine_kernel, dim3(C, s.nsplit), dim3(256), 0, pgv_stream(stream), a, scale, shift, B, C, HW,
                     s.per, o);
  PGV_CHECK_LAUNCH("affine_nchw");
  return PGV_OK;
}

int pgv_bn_bwd_reduce(const float* g_o, const float* a, const float* mean, const float* rstd, int B, int C, int HW,
                      double* red, int flags, void* stream) {
  PGV_CHECK_ARG(g_o && a && mean && rstd && red && B >= 0 && C > 0 && HW > 0, "pgv_bn_bwd_reduce: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (!(flags & PGV_PREZEROED)) {
    int rc = zero_async(red, sizeof(double) * 2 * C, st, "pgv_bn_bwd_reduce");
    if (rc) return rc;
  }
  if (B == 0) return PGV_OK;
  Split s = pick_split(B, C, HW);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, s.nsplit), dim3(256), 0, st, g_o, a, mean, rstd, B, C, HW, s.per,
                     red);
  PGV_CHECK_LAUNCH("bn_bwd_reduce");
  return PGV_OK;
}

int pgv_dropout_bwd_bn_reduce(const uint64_t* saved_state, uint64_t stream_id, float p, const float* g_d, const float* a,
                              const float* mean, const float* rstd, int B, int C, int HW, float* gx, double* red, int flags,
                              void* stream) {
  PGV_CHECK_ARG(saved_state && g_d && a && mean && rstd && gx && red && B >= 0 && C > 0 && HW > 0 && p >= 0.f && p < 1.f,
                "pgv_dropout_bwd_bn_reduce: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (!(flags & PGV_PREZEROED)) {
    int rc = zero_async(red, sizeof(double) * 2 * C, st, "pgv_dropout_bwd_bn_reduce");
    if (rc) return rc;
  }
  if (B == 0) return PGV_OK;
  Split s = pick_split(B, C, HW);
  hipLaunchKernelGGL(dropout_bwd_bn_reduce_kernel, dim3(C, s.nsplit), dim3(256), 0, st, saved_state, stream_id, p,
                     1.0f / (1.0f - p), g_d, a, mean, rstd, B, C, HW, s.per, gx, red);
  PGV_CHECK_LAUNCH("dropout_bwd_bn_reduce");
  return PGV_OK;
}

int pgv_bn_act_bwd_fusable(int B, int C, int HW) { return bn_act_bwd_fusable(B, C, HW) ? 1 : 0; }

int pgv_bn_act_bwd_fused(const float* g_o, const float* a, const float* scale, const float* mean, const float* rstd, int B,
                         int C, int HW, int act, float slope, float* g_y, float* gbias, float* ggamma, float* gbeta,
                         int flags, void* stream) {
  PGV_CHECK_ARG(g_o && a && g_y && scale && mean && rstd && B >= 0 && C > 0 && HW > 0, "pgv_bn_act_bwd_fused: bad argument");
  PGV_CHECK_ARG(bn_act_bwd_fusable(B, C, HW), "pgv_bn_act_bwd_fused: shape not served (pgv_bn_act_bwd_fusable)");
  PGV_CHECK_ARG(act == PGV_ACT_LEAKY_RELU || act == PGV_ACT_HARDTANH || act == PGV_ACT_NONE, "pgv_bn_act_bwd_fused: activation");
  hipStream_t st = pgv_stream(stream);
  if (gbias && !(flags & PGV_PREZEROED)) {
    int rc = zero_async(gbias, sizeof(float) * C, st, "pgv_bn_act_bwd_fused");
    if (rc) return rc;
  }
  const int groups = B * (HW >> 2);
  const int threads = groups <= 768 ? 256 : 512;
  typedef void (*kern_t)(const float*, const float*, const float*, const float*, const float*, int, int, int, float, float*,
                         float*, float*, float*);
  kern_t kern;
  if (act == PGV_ACT_LEAKY_RELU)
    kern = (kern_t)bn_act_bwd_fused_kernel<PGV_ACT_LEAKY_RELU, 4>;
  else if (act == PGV_ACT_HARDTANH)
    kern = (kern_t)bn_act_bwd_fused_kernel<PGV_ACT_HARDTANH, 4>;
  else
    kern = (kern_t)bn_act_bwd_fused_kernel<PGV_ACT_NONE, 4>;
  hipLaunchKernelGGL(kern, dim3(C), dim3(threads), 0, st, g_o, a, scale, mean, rstd, B, C, HW, slope, g_y, gbias, ggamma, gbeta);
  PGV_CHECK_LAUNCH("bn_act_bwd_fused");
  return PGV_OK;
}

int pgv_act_bn_bwd(const float* g_o, const float* a, const float* scale, const float* mean, const float* rstd,
                   const double* red, int B, int C, int HW, int act, float slope, float* g_y, float* gbias,
                   float* ggamma, float* gbeta, int flags, void* stream) {
  PGV_CHECK_ARG(g_o && a && g_y && B >= 0 && C > 0 && HW > 0, "pgv_act_bn_bwd: bad argument");
  PGV_CHECK_ARG(red == nullptr || (scale && mean && rstd), "pgv_act_bn_bwd: train-mode BN needs scale/mean/rstd");
  hipStream_t st = pgv_stream(stream);
  PGV_CHECK_ARG((ggamma == nullptr && gbeta == nullptr) || red != nullptr, "pgv_act_bn_bwd: ggamma/gbeta need red");
  if (gbias && !(flags & PGV_PREZEROED)) {
    int rc = zero_async(gbias, sizeof(float) * C, st, "pgv_act_bn_bwd");
    if (rc) return rc;
  }
  if (B == 0) return PGV_OK;
  if (!scale && HW % 4 == 0 && HW <= 64 && C >= 256 && C <= 8192 &&
      (((uintptr_t)g_o | (uintptr_t)a | (uintptr_t)g_y) & 15) == 0) {   // no BatchNorm, small planes: flat walk
    typedef void (*flat_t)(const f32x4*, const f32x4*, int, int, int, float, f32x4*, float*);
    flat_t fk = act == PGV_ACT_LEAKY_RELU ? (flat_t)act_bwd_flat_kernel<PGV_ACT_LEAKY_RELU>
                                          : (act == PGV_ACT_HARDTANH ? (flat_t)act_bwd_flat_kernel<PGV_ACT_HARDTANH>
                                                                     : (flat_t)act_bwd_flat_kernel<PGV_ACT_NONE>);
    hipLaunchKernelGGL(fk, dim3((unsigned)min(B, 1024)), dim3(512), sizeof(float) * C, st, (const f32x4*)g_o, (const f32x4*)a, B, C,
                       HW / 4, slope, (f32x4*)g_y, gbias);
    PGV_CHECK_LAUNCH("act_bwd_flat");
    return PGV_OK;
  }
  Split s = pick_split(B, C, HW);
  const double inv_n = 1.0 / ((double)B * HW);
  typedef void (*kern_t)(const float*, const float*, const float*, const float*, const float*, const double*, double,
                         int, int, int, int, int, float, float*, float*, float*, float*);
  // activation and BatchNorm presence are compile-time in the kernel (no branches inside the element loop)
  const bool bn = scale != nullptr;
  kern_t kern;
  if (act == PGV_ACT_LEAKY_RELU)
    kern = bn ? (kern_t)act_bn_bwd_kernel<PGV_ACT_LEAKY_RELU, true> : (kern_t)act_bn_bwd_kernel<PGV_ACT_LEAKY_RELU, false>;
  else if (act == PGV_ACT_HARDTANH)
    kern = bn ? (kern_t)act_bn_bwd_kernel<PGV_ACT_HARDTANH, true> : (kern_t)act_bn_bwd_kernel<PGV_ACT_HARDTANH, false>;
  else
    kern = bn ? (kern_t)act_bn_bwd_kernel<PGV_ACT_NONE, true> : (kern_t)act_bn_bwd_kernel<PGV_ACT_NONE, false>;
  hipLaunchKernelGGL(kern, dim3(C, s.nsplit), dim3(256), 0, st, g_o, a, scale, mean, rstd, red, inv_n, B, C, HW, s.per,
                     act, slope, g_y, gbias, ggamma, gbeta);
  PGV_CHECK_LAUNCH("act_bn_bwd");
  return PGV_OK;
}

int pgv_act_bwd_coef(const float* g, const float* a, const float* coef, int B, int C, int HW, int act, float slope,
                     float* g_y, float* gbias, int flags, void* stream) {
  PGV_CHECK_ARG(g && a && coef && g_y && B >= 0 && C > 0 && HW > 0, "pgv_act_bwd_coef: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (gbias && !(flags & PGV_PREZEROED)) {
    int rc = zero_async(gbias, sizeof(float) * C, st, "pgv_act_bwd_coef");
    if (rc) return rc;
  }
  return pgv_act_bwd_coef_impl(g, a, coef, B, C, HW, act, slope, g_y, gbias, st);
}

int pgv_conv_class_sums(const pgv_conv_desc* d, int gy_is_big, const float* gy, float* cls, int flags, void* stream) {
  PGV_CHECK_ARG(d && gy && cls, "pgv_conv_class_sums: null argument");
  const int C = gy_is_big ? d->Cb : d->Cs, H = gy_is_big ? d->Hb : d->Hs, W = gy_is_big ? d->Wb : d->Ws;
  const int m = gy_is_big ? d->stride : 1;
  PGV_CHECK_ARG(m >= 1 && m <= 3, "pgv_conv_class_sums: stride %d not supported", m);
  hipStream_t st = pgv_stream(stream);
  if (!(flags & PGV_PREZEROED)) {
    int rc = zero_async(cls, sizeof(float) * C * m * m * (gy_is_big ? PGV_CLS_COPIES : 1), st, "pgv_conv_class_sums");
    if (rc) return rc;
  }
  if (d->B == 0) return PGV_OK;
  Split sp = pick_split(d->B, C, H * W);
  typedef void (*kern_t)(const float*, int, int, int, int, int, float*);
  kern_t kern = m == 1 ? (kern_t)class_sums_kernel<1> : (m == 2 ? (kern_t)class_sums_kernel<2> : (kern_t)class_sums_kernel<3>);
  hipLaunchKernelGGL(kern, dim3(C, sp.nsplit), dim3(256), 0, st, gy, d->B, C, H, W, sp.per, cls);
  PGV_CHECK_LAUNCH("conv_class_sums");
  return PGV_OK;
}

static int tap_sums_launch(const pgv_conv_desc* d, int gy_is_big, const float* gy, const float* cls, double* T, int flags,
                           hipStream_t st, int cls_copies = 0) {
  const int C = gy_is_big ? d->Cb : d->Cs, H = gy_is_big ? d->Hb : d->Hs, W = gy_is_big ? d->Wb : d->Ws;
  const int oH = gy_is_big ? d->Hs : d->Hb, oW = gy_is_big ? d->Ws : d->Wb;
  const int K = d->kh;
  if (!(flags & PGV_PREZEROED)) {
    int rc = zero_async(T, sizeof(double) * C * K * K, st, "pgv_conv_tap_sums");
    if (rc) return rc;
  }
  if (d->B == 0) return PGV_OK;
  TapBorder tb;
  if (cls && tap_axis(gy_is_big != 0, H, K, d->stride, d->pad, oH, &tb.ra, &tb.rb, tb.rm) &&
      tap_axis(gy_is_big != 0, W, K, d->stride, d->pad, oW, &tb.ca, &tb.cb, tb.cm)) {
    // border form: ~512 workgroups of up to 16 planes x 512 border positions each
    const int NE = (tb.ra + tb.rb) * W + (H - tb.ra - tb.rb) * (tb.ca + tb.cb);
    const int nz = (int)max((int64_t)1, pgv_cdiv(NE, kTapChunk));
    const int per = (int)max((int64_t)1, min((int64_t)16, pgv_cdiv((int64_t)d->B * C * nz, 512)));
    const int nsplit = (int)pgv_cdiv(d->B, per);
    typedef void (*kern_t)(const float*, int, int, int, int, int, int, int, int, TapBorder, const float*, int, double*);
    kern_t kern = nullptr;
    switch (K) {
      case 1: kern = (kern_t)tap_border_kernel<1>; break;
      case 2: kern = (kern_t)tap_border_kernel<2>; break;
      case 3: kern = (kern_t)tap_border_kernel<3>; break;
      case 4: kern = (kern_t)tap_border_kernel<4>; break;
      default: kern = (kern_t)tap_border_kernel<5>; break;
    }
    hipLaunchKernelGGL(kern, dim3(C, nsplit, nz), dim3(256), 0, st, gy, d->B, C, H, W, per, gy_is_big, d->stride, d->pad,
                       tb, cls, cls_copies > 0 ? cls_copies : (gy_is_big ? PGV_CLS_COPIES : 1), T);
    PGV_CHECK_LAUNCH("conv_tap_sums (border)");
    return PGV_OK;
  }
  // ~2048 workgroups, at least 64 K elements each (the K*K-value tail reduction is amortised over them)
  const int64_t want = pgv_cdiv(2048, C), by_work = pgv_cdiv((int64_t)d->B * H * W, 65536);
  const int ns = (int)max((int64_t)1, min((int64_t)d->B, min(want, by_work)));
  const int per = (int)pgv_cdiv(d->B, ns), nsplit = (int)pgv_cdiv(d->B, per);
  const int NJ = (int)pgv_cdiv(W, 256);
  typedef void (*kern_t)(const float*, int, int, int, int, int, int, int, int, int, int, double*);
  kern_t kern = nullptr;
#define PGV_TS(KV)                                                                                      \
  case KV:                                                                                              \
    kern = NJ == 1 ? (kern_t)tap_sums_kernel<KV, 1>                                                     \
                   : (NJ == 2 ? (kern_t)tap_sums_kernel<KV, 2> : (kern_t)tap_sums_kernel<KV, 4>);       \
    break;
  switch (K) {
    PGV_TS(1) PGV_TS(2) PGV_TS(3) PGV_TS(4) PGV_TS(5)
  }
#undef PGV_TS
  hipLaunchKernelGGL(kern, dim3(C, nsplit), dim3(256), 0, st, gy, d->B, C, H, W, per, gy_is_big, d->stride, d->pad, oH,
                     oW, T);
  PGV_CHECK_LAUNCH("conv_tap_sums");
  return PGV_OK;
}

int pgv_conv_tap_sums(const pgv_conv_desc* d, int gy_is_big, const float* gy, const float* cls, double* T, int flags,
                      void* stream) {
  PGV_CHECK_ARG(d && gy && T, "pgv_conv_tap_sums: null argument");
  PGV_CHECK_ARG(d->kh == d->kw && d->kh >= 1 && d->kh <= 5, "pgv_conv_tap_sums: kernel %dx%d not supported", d->kh, d->kw);
  const int H = gy_is_big ? d->Hb : d->Hs, W = gy_is_big ? d->Wb : d->Ws;
  PGV_CHECK_ARG(H <= 1024 && W <= 1024, "pgv_conv_tap_sums: plane %dx%d too large", H, W);
  return tap_sums_launch(d, gy_is_big, gy, cls, T, flags, pgv_stream(stream));
}

static CoefArgs coef_args(const pgv_conv_desc* d, int lower_is_big, const float* w, const float* gw, const float* scale,
                          const float* shift, const float* mean, const float* rstd, int64_t n, float* coef, float* ggamma,
                          float* gbeta) {
  CoefArgs ca;
  ca.w = w, ca.gw = gw, ca.scale = scale, ca.shift = shift, ca.mean = mean, ca.rstd = rstd;
  ca.coef = coef, ca.ggamma = ggamma, ca.gbeta = gbeta;
  ca.inv_n = 1.0 / (double)n;
  ca.Cb = d->Cb, ca.Cs = d->Cs, ca.KK = d->kh * d->kw, ca.lower_is_big = lower_is_big;
  ca.bf16 = (d->flags & PGV_COMPUTE_BF16) ? 1 : 0;
  ca.C = lower_is_big ? d->Cb : d->Cs;
  return ca;
}

int pgv_bn_bwd_coef_from_gy(const pgv_conv_desc* d, int lower_is_big, const float* gy, const float* cls, double* T,
                            const float* w, const float* gw, const float* scale, const float* shift, const float* mean,
                            const float* rstd, int64_t n, float* coef, float* ggamma, float* gbeta, int flags,
                            void* stream) {
  PGV_CHECK_ARG(d && gy && T && w && gw && scale && shift && mean && rstd && coef && n > 0,
                "pgv_bn_bwd_coef_from_gy: bad argument");
  PGV_CHECK_ARG(d->kh == d->kw && d->kh >= 1 && d->kh <= 5, "pgv_bn_bwd_coef_from_gy: kernel %dx%d not supported", d->kh, d->kw);
  const int gy_is_big = !lower_is_big;
  const int H = gy_is_big ? d->Hb : d->Hs, W = gy_is_big ? d->Wb : d->Ws;
  PGV_CHECK_ARG(H <= 1024 && W <= 1024, "pgv_bn_bwd_coef_from_gy: plane %dx%d too large", H, W);
  hipStream_t st = pgv_stream(stream);
  const CoefArgs ca = coef_args(d, lower_is_big, w, gw, scale, shift, mean, rstd, n, coef, ggamma, gbeta);
  int rc = tap_sums_launch(d, gy_is_big, gy, cls, T, flags, st);
  if (rc || d->B == 0) return rc;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(ca.C), dim3(256), 0, st, ca, (const double*)T);
  PGV_CHECK_LAUNCH("bn_bwd_coef");
  return PGV_OK;
}

extern "C++" {
int pgv_tap_replicas(int c_gy, int kk) { return tap_replicas(c_gy, kk); }
int pgv_bn_bwd_coef_from_gy_cc(const pgv_conv_desc* d, int lower_is_big, const float* gy, const float* cls, int cls_copies,
                               double* T, const float* w, const float* gw, const float* scale, const float* shift,
                               const float* mean, const float* rstd, int64_t n, float* coef, float* ggamma, float* gbeta,
                               int flags, hipStream_t st) {
  const int gy_is_big = !lower_is_big;
  const CoefArgs ca = coef_args(d, lower_is_big, w, gw, scale, shift, mean, rstd, n, coef, ggamma, gbeta);
  int rc = tap_sums_launch(d, gy_is_big, gy, cls, T, flags, st, cls_copies);
  if (rc || d->B == 0) return rc;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(ca.C), dim3(256), 0, st, ca, (const double*)T);
  PGV_CHECK_LAUNCH("bn_bwd_coef");
  return PGV_OK;
}
int pgv_bn_bwd_coef_rep(const pgv_conv_desc* d, int lower_is_big, const float* w, const float* gw, const double* T, int trep,
                        const float* scale, const float* shift, const float* mean, const float* rstd, int64_t n,
                        float* coef, float* ggamma, float* gbeta, hipStream_t st) {
  CoefArgs ca = coef_args(d, lower_is_big, w, gw, scale, shift, mean, rstd, n, coef, ggamma, gbeta);
  ca.trep = trep;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(ca.C), dim3(256), 0, st, ca, T);
  PGV_CHECK_LAUNCH("bn_bwd_coef");
  return PGV_OK;
}
}  // extern "C++"

int pgv_bn_bwd_coef(const pgv_conv_desc* d, int lower_is_big, const float* w, const float* gw, const double* T,
                    const float* scale, const float* shift, const float* mean, const float* rstd, int64_t n,
                    float* coef, float* ggamma, float* gbeta, void* stream) {
  PGV_CHECK_ARG(d && w && gw && T && scale && shift && mean && rstd && coef && n > 0, "pgv_bn_bwd_coef: bad argument");
  const CoefArgs ca = coef_args(d, lower_is_big, w, gw, scale, shift, mean, rstd, n, coef, ggamma, gbeta);
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(ca.C), dim3(256), 0, pgv_stream(stream), ca, T);
  PGV_CHECK_LAUNCH("bn_bwd_coef");
  return PGV_OK;
}

int pgv_sqerr_act_bwd(const float* a, const float* x, const float* g_loss, float scale, int B, int C, int HW, int act,
                      float slope, float* g_y, float* gbias, float* loss_acc, int flags, void* stream) {
  PGV_CHECK_ARG(a && x && g_loss && g_y && B >= 0 && C > 0 && HW > 0, "pgv_sqerr_act_bwd: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (gbias && !(flags & PGV_PREZEROED)) {
    int rc = zero_async(gbias, sizeof(float) * C, st, "pgv_sqerr_act_bwd");
    if (rc) return rc;
  }
  if (B == 0) return PGV_OK;
  if (C == 1) {
    const int64_t n = (int64_t)B * HW;
    const unsigned blocks = (unsigned)max((int64_t)1, min((int64_t)2048, pgv_cdiv(n, 256 * 16)));
    hipLaunchKernelGGL(sqerr_act_bwd_flat_kernel, dim3(blocks), dim3(256), 0, st, a, x, g_loss, scale, n, act, slope,
                       g_y, gbias, loss_acc);
    PGV_CHECK_LAUNCH("sqerr_act_bwd");
    return PGV_OK;
  }
  Split s = pick_split(B, C, HW);
  hipLaunchKernelGGL(sqerr_act_bwd_kernel, dim3(C, s.nsplit), dim3(256), 0, st, a, x, g_loss, scale, B, C, HW, s.per,
                     act, slope, g_y, gbias, loss_acc);
  PGV_CHECK_LAUNCH("sqerr_act_bwd");
  return PGV_OK;
}

int pgv_sqerr_act_bwd_cls(const float* a, const float* x, const float* g_loss, float scale, int B, int C, int HW, int W,
                          int act, float slope, float* g_y, float* gbias, float* loss_acc, float* cls, int flags,
                          void* stream) {
  PGV_CHECK_ARG(a && x && g_loss && g_y && cls && B >= 0 && C == 1 && HW > 0 && W > 0 && HW % W == 0,
                "pgv_sqerr_act_bwd_cls: bad argument (single-channel tensors only)");
  hipStream_t st = pgv_stream(stream);
  if (gbias && !(flags & PGV_PREZEROED)) {
    int rc = zero_async(gbias, sizeof(float) * C, st, "pgv_sqerr_act_bwd_cls");
    if (rc) return rc;
  }
  if (B == 0) return PGV_OK;
  const int H = HW / W;
  PGV_CHECK_ARG(HW >= kSqChunk && W + kSqChunk < (1 << 20), "pgv_sqerr_act_bwd_cls: planes of at least %d elements", kSqChunk);
  const int64_t n = (int64_t)B * HW;
  hipLaunchKernelGGL(sqerr_act_bwd_cls_kernel, dim3((unsigned)pgv_cdiv(n, kSqChunk)), dim3(256), 0, st, a, x, g_loss, scale,
                     n, H, W, act, slope, g_y, gbias, loss_acc, cls);
  PGV_CHECK_LAUNCH("sqerr_act_bwd_cls");
  return PGV_OK;
}

int pgv_colsum(const float* x, int M, int N, int64_t ld, float* out, int flags, void* stream) {
  PGV_CHECK_ARG(x && out && M >= 0 && N > 0 && ld >= N, "pgv_colsum: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (!(flags & PGV_PREZEROED)) {
    int rc = zero_async(out, sizeof(float) * N, st, "pgv_colsum");
    if (rc) return rc;
  }
  if (M == 0) return PGV_OK;
  const int gy = (int)max((int64_t)1, min((int64_t)pgv_cdiv(M, 32), pgv_cdiv(1024, pgv_cdiv(N, 64))));
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)pgv_cdiv(N, 64), gy), dim3(256), 0, st, x, M, N, ld, out);
  PGV_CHECK_LAUNCH("colsum");
  return PGV_OK;
}

}  // extern "C"
