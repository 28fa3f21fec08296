// BatchNorm pieces for [B,C,HW] fp32 tensors (nn.BatchNorm2d / BatchNorm1d, train mode; reference
// model/layer.py:21-26, model/encoder.py:86-87; backward per SURVEY Appendix B).
//
// All kernels use a (channel, batch-split) grid: per-channel constants are block-uniform, planes are read
// with lanes on consecutive hw (coalesced NCHW rows), and every per-channel sum is a wave-shuffle + LDS block
// reduction followed by ONE float atomic per block.  HBM-bound: one read (two for the backward pieces) and at
// most one write per element.
#include "conv_kernels.h"

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

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int C, double inv_n, double unbias,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  if (c >= C) return;
  const double mean = (double)stats[c] * inv_n;
  double var = (double)stats[C + c] * inv_n - mean * mean;
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

int pgv_bn_bwd_reduce_impl(const float* g_o, const float* a, const float* mean, const float* rstd, int B, int C, int HW,
                           double* red, hipStream_t st) {
  if (B == 0) return PGV_OK;
  Split s = pick_split(B, C, HW);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, s.nsplit), dim3(256), 0, st, g_o, a, mean, rstd, B, C, HW, s.per,
                     red);
  PGV_CHECK_LAUNCH("bn_bwd_reduce");
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
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)pgv_cdiv(C, 128)), dim3(128), 0, pgv_stream(stream), stats,
                     C, 1.0 / (double)n, unbias, gamma, beta, eps, momentum, running_mean, running_var,
                     num_batches_tracked, scale, shift, mean, rstd);
  PGV_CHECK_LAUNCH("bn_finalize");
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
