// Element-wise / reduction kernels of the VAE step: dropout masks and eps (counter-based Philox), the
// reparameterisation + KL (model/VAE.py:49-58, model/loss.py:57-66), squared-error loss with the Hardtanh
// gate (train.py:104,222; model/loss.py:37-43; model/decoder.py:98), fused Adam (train.py:166-167).
// All are HBM-bound streaming kernels: 16 B per lane where alignment allows, grid capped at 2048 blocks with a
// grid-stride loop, one atomic per block for reductions.
#include "pgv_common.h"
#include "philox.h"

namespace {

constexpr int kBlock = 256;
inline unsigned grid_for(int64_t n, int per_thread = 4) {
  int64_t blocks = pgv_cdiv(n, (int64_t)kBlock * per_thread);
  return (unsigned)max((int64_t)1, min((int64_t)2048, blocks));
}

__global__ void dropout_mask_kernel(const uint64_t* __restrict__ rng, uint64_t stream_id, float p, float keep_scale,
                                    int64_t n, float* __restrict__ mask) {
  const uint64_t seed = rng[0], off = rng[1];
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const U4 r = philox4x32_10(off + (uint64_t)q, stream_id, seed);
    const float v[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = q * 4 + j;
      if (i < n) mask[i] = v[j] >= p ? keep_scale : 0.f;
    }
  }
}

// nn.Dropout forward in one pass: the keep mask is drawn, applied and stored (backward multiplies by it) - same
// stream / counters as dropout_mask_kernel, so the two paths draw identical masks from identical states.
__global__ void dropout_apply_kernel(const uint64_t* __restrict__ rng, uint64_t stream_id, float p, float keep_scale,
                                     int64_t n, const float* __restrict__ x, float* __restrict__ y,
                                     float* __restrict__ mask, int vec) {
  const uint64_t seed = rng[0], off = rng[1];
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const U4 r = philox4x32_10(off + (uint64_t)q, stream_id, seed);
    const float m[4] = {u01(r.x) >= p ? keep_scale : 0.f, u01(r.y) >= p ? keep_scale : 0.f,
                        u01(r.z) >= p ? keep_scale : 0.f, u01(r.w) >= p ? keep_scale : 0.f};
    if (vec && q * 4 + 4 <= n) {
      const float4 xv = reinterpret_cast<const float4*>(x)[q];
      reinterpret_cast<float4*>(y)[q] = make_float4(xv.x * m[0], xv.y * m[1], xv.z * m[2], xv.w * m[3]);
      reinterpret_cast<float4*>(mask)[q] = make_float4(m[0], m[1], m[2], m[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t i = q * 4 + j;
        if (i < n) {
          y[i] = x[i] * m[j];
          mask[i] = m[j];
        }
      }
    }
  }
}

// nn.Dropout without a stored mask: Philox is counter-based, so backward regenerates the mask from a copy of the
// generator state the forward drew from (two words) instead of reading back 4 bytes per element - 2 instead of 3 tensors
// of traffic per pass.  Optional per-channel affine on the way in (x is [B][C][HW]): the BatchNorm of the encoder's last
// conv block, which has no consumer kernel to fold it into (encoder.py:85: Dropout -> Linear).
// (bn.stats != null, C <= kDropBnMaxC: the affine is the BatchNorm bn, finalized here into an LDS table by every workgroup
// - pgv_bn_src - instead of by a launch of its own)
constexpr int kDropBnMaxC = 512;
// AFF: 0 no affine, 1 scale / shift read from global memory, 2 from an LDS table (C <= kDropBnMaxC) that every workgroup
// fills in its prologue - from the vectors, or from the BatchNorm's statistics (bn.stats).  Compile-time: with the table
// chosen by a run-time pointer the per-quad reads became flat loads and the streaming loop lost its overlap (21 us
// against 10 for the plain kernel).
template <int AFF>
__global__ void dropout_fwd_kernel(const uint64_t* __restrict__ rng, uint64_t stream_id, float p, float keep_scale,
                                   int64_t n, const float* __restrict__ x, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int C, int64_t HW, float* __restrict__ y,
                                   uint64_t* __restrict__ saved, int vec, pgv_bn_src bn) {
  __shared__ float tab[AFF == 2 ? 2 * kDropBnMaxC : 2];
  if constexpr (AFF == 2) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float sc, sh;
      if (bn.stats)
        pgv_bn_finalize_dev(bn, C, c, blockIdx.x == 0, sc, sh);
      else
        sc = scale[c], sh = shift[c];
      tab[c] = sc, tab[kDropBnMaxC + c] = sh;
    }
    __syncthreads();
  }
  auto aff_of = [&](int c, float& sc, float& sh) {
    if constexpr (AFF == 2)
      sc = tab[c], sh = tab[kDropBnMaxC + c];
    else
      sc = scale[c], sh = shift[c];
  };
  const uint64_t seed = rng[0], off = rng[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) saved[0] = seed, saved[1] = off;
  const int64_t n4 = (n + 3) / 4;
  // channel of the quad, kept incrementally along the grid-stride walk (a 64-bit division per quad cost more than the
  // Philox rounds): position (ch, rem) of element 4q inside its sample, advanced by the stride's (channels, remainder)
  const int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  // (32-bit divisions: four 64-bit ones per thread - ~1000 instructions - were half of this kernel's 21 us)
  int ch = 0, rem = 0, d_ch = 0, d_rem = 0;
  if (AFF != 0 && vec) {
    if (n < ((int64_t)1 << 31)) {
      const unsigned e0 = (unsigned)q0 * 4u, es = (unsigned)stride * 4u, hw = (unsigned)HW;
      const unsigned p0 = e0 / hw, ps = es / hw;
      ch = (int)(p0 % (unsigned)C), rem = (int)(e0 - p0 * hw);
      d_ch = (int)(ps % (unsigned)C), d_rem = (int)(es - ps * hw);
    } else {
      ch = (int)((q0 * 4 / HW) % C), rem = (int)(q0 * 4 % HW);
      d_ch = (int)((stride * 4 / HW) % C), d_rem = (int)(stride * 4 % HW);
    }
  }
  for (int64_t q = q0; q < n4; q += stride) {
    const U4 r = philox4x32_10(off + (uint64_t)q, stream_id, seed);
    const float m[4] = {u01(r.x) >= p ? keep_scale : 0.f, u01(r.y) >= p ? keep_scale : 0.f,
                        u01(r.z) >= p ? keep_scale : 0.f, u01(r.w) >= p ? keep_scale : 0.f};
    if (vec && q * 4 + 4 <= n) {   // (vec: 16-byte aligned; with an affine HW >= 4: a quad lies in at most two channels)
      float4 xv = reinterpret_cast<const float4*>(x)[q];
      if constexpr (AFF != 0) {
        // (planes of 17x23 = 391 elements - the encoder's - are not multiples of 4: elements past the plane's end take
        // the next channel's pair)
        const int c = ch, left = (int)HW - rem;   // elements of this quad that still belong to channel c
        rem += d_rem, ch += d_ch;
        if (rem >= (int)HW) rem -= (int)HW, ch += 1;
        if (ch >= C) ch -= C;
        float sc, sh, sc2, sh2;
        aff_of(c, sc, sh);
        aff_of(c + 1 < C ? c + 1 : 0, sc2, sh2);
        xv = make_float4(fmaf(xv.x, sc, sh), left > 1 ? fmaf(xv.y, sc, sh) : fmaf(xv.y, sc2, sh2),
                         left > 2 ? fmaf(xv.z, sc, sh) : fmaf(xv.z, sc2, sh2),
                         left > 3 ? fmaf(xv.w, sc, sh) : fmaf(xv.w, sc2, sh2));
      }
      reinterpret_cast<float4*>(y)[q] = make_float4(xv.x * m[0], xv.y * m[1], xv.z * m[2], xv.w * m[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t i = q * 4 + j;
        if (i < n) {
          float xv = x[i];
          if constexpr (AFF != 0) {
            float sc, sh;
            aff_of((int)((i / HW) % C), sc, sh);
            xv = fmaf(xv, sc, sh);
          }
          y[i] = xv * m[j];
        }
      }
    }
  }
}

__global__ void dropout_bwd_kernel(const uint64_t* __restrict__ saved, uint64_t stream_id, float p, float keep_scale,
                                   int64_t n, const float* __restrict__ gy, float* __restrict__ gx, int vec) {
  const uint64_t seed = saved[0], off = saved[1];
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const U4 r = philox4x32_10(off + (uint64_t)q, stream_id, seed);
    const float m[4] = {u01(r.x) >= p ? keep_scale : 0.f, u01(r.y) >= p ? keep_scale : 0.f,
                        u01(r.z) >= p ? keep_scale : 0.f, u01(r.w) >= p ? keep_scale : 0.f};
    if (vec && q * 4 + 4 <= n) {
      const float4 g = reinterpret_cast<const float4*>(gy)[q];
      reinterpret_cast<float4*>(gx)[q] = make_float4(g.x * m[0], g.y * m[1], g.z * m[2], g.w * m[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t i = q * 4 + j;
        if (i < n) gx[i] = gy[i] * m[j];
      }
    }
  }
}

// four standard normals from one Philox block: Box-Muller, two pairs; u in (0,1] for the log
__device__ __forceinline__ void normal4(const U4 r, float (&v)[4]) {
  const float u0 = 1.0f - u01(r.x), u1 = u01(r.y), u2 = 1.0f - u01(r.z), u3 = u01(r.w);
  const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  v[0] = r0 * c0, v[1] = r0 * s0, v[2] = r1 * c1, v[3] = r1 * s1;
}

// dropout_bwd_kernel over a [M][N] gradient with the column sums of the result on the way (the bias gradient of the
// nn.Linear in front of the Dropout, decoder.py:64-65): a workgroup is 64 quad-columns x 4 row groups, a thread owns four
// columns and R rows (loads in flight per thread = R), the four row groups are added up through LDS: one float atomic per
// column per 4R rows.  N % 4 == 0, 16-byte aligned rows.
template <int R>
__global__ __launch_bounds__(256) void dropout_bwd_colsum_kernel(const uint64_t* __restrict__ saved, uint64_t stream_id,
                                                                 float p, float keep_scale, int M, int N4,
                                                                 const float4* __restrict__ gy, float4* __restrict__ gx,
                                                                 float* __restrict__ colsum) {
  __shared__ float4 part[4][64];
  const uint64_t seed = saved[0], off = saved[1];
  const int ql = threadIdx.x & 63, rgp = threadIdx.x >> 6;
  const int qc = blockIdx.x * 64 + ql;
  const int r0 = (blockIdx.y * 4 + rgp) * R;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (qc < N4) {
    float4 g[R];
#pragma unroll
    for (int u = 0; u < R; ++u)
      if (r0 + u < M) g[u] = gy[(int64_t)(r0 + u) * N4 + qc];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (r0 + u >= M) break;
      const int64_t q = (int64_t)(r0 + u) * N4 + qc;
      const U4 r = philox4x32_10(off + (uint64_t)q, stream_id, seed);
      const float4 o = make_float4(u01(r.x) >= p ? g[u].x * keep_scale : 0.f, u01(r.y) >= p ? g[u].y * keep_scale : 0.f,
                                   u01(r.z) >= p ? g[u].z * keep_scale : 0.f, u01(r.w) >= p ? g[u].w * keep_scale : 0.f);
      gx[q] = o;
      acc.x += o.x, acc.y += o.y, acc.z += o.z, acc.w += o.w;
    }
  }
  part[rgp][ql] = acc;
  __syncthreads();
  if (rgp == 0 && qc < N4) {
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 t = part[k][ql];
      acc.x += t.x, acc.y += t.y, acc.z += t.z, acc.w += t.w;
    }
    atomicAdd(&colsum[4 * qc], acc.x);
    atomicAdd(&colsum[4 * qc + 1], acc.y);
    atomicAdd(&colsum[4 * qc + 2], acc.z);
    atomicAdd(&colsum[4 * qc + 3], acc.w);
  }
}

__global__ void normal_kernel(const uint64_t* __restrict__ rng, uint64_t stream_id, int64_t n,
                              float* __restrict__ out) {
  const uint64_t seed = rng[0], off = rng[1];
  const int64_t n4 = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    float v[4];
    normal4(philox4x32_10(off + (uint64_t)q, stream_id, seed), v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = q * 4 + j;
      if (i < n) out[i] = v[j];
    }
  }
}

__global__ void rng_advance_kernel(uint64_t* rng, uint64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += inc;
}

__global__ void mul_kernel(const float* __restrict__ x, const float* __restrict__ m, int64_t n,
                           float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i] * m[i];
}

// 16-byte-per-lane variants (used when every pointer is 16-byte aligned; the n % 4 tail goes through the scalar kernel)
__global__ void mul4_kernel(const float4* __restrict__ x, const float4* __restrict__ m, int64_t n4,
                            float4* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = x[i], b = m[i];
    y[i] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
  }
}

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

__global__ void axpy_kernel(int64_t n, float a, const float* __restrict__ x, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = fmaf(a, x[i], y[i]);
}

__global__ void copy4_kernel(const float4* __restrict__ s, float4* __restrict__ d, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = s[i];
}

// ---- reparameterisation + KL ------------------------------------------------------------------------------
__global__ void reparam_kl_fwd_kernel(const float* __restrict__ ml, const float* __restrict__ eps, int B, int D,
                                      float kl_scale, float* __restrict__ z, float* __restrict__ kl) {
  __shared__ float red[16];
  float acc = 0.f;
  const int64_t n = (int64_t)B * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / D;
    const int dd = (int)(i - b * D);
    const float mu = ml[(b * 2) * D + dd], lv = ml[(b * 2 + 1) * D + dd];
    const float var = expf(lv);
    acc += var + mu * mu - lv - 1.0f;
    if (z) z[i] = eps ? fmaf(expf(0.5f * lv), eps[i], mu) : mu;
  }
  const float s = pgv_block_sum(acc, red);
  if (threadIdx.x == 0 && kl) atomicAdd(kl, 0.5f * kl_scale * s);
}

// The same with eps drawn here - element i of the draw pgv_normal makes from the same state and stream (one Philox block
// per element instead of per four: B*D is 16 K values) - and stored for backward: one launch for the draw, the
// reparameterisation and the Dkl term instead of three.
__global__ void reparam_kl_fwd_rng_kernel(const float* __restrict__ ml, const uint64_t* __restrict__ rng,
                                          uint64_t stream_id, int B, int D, float kl_scale, float* __restrict__ z,
                                          float* __restrict__ eps_out, float* __restrict__ kl) {
  __shared__ float red[16];
  const uint64_t seed = rng[0], off = rng[1];
  float acc = 0.f;
  const int64_t n = (int64_t)B * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / D;
    const int dd = (int)(i - b * D);
    const float mu = ml[(b * 2) * D + dd], lv = ml[(b * 2 + 1) * D + dd];
    float v[4];
    normal4(philox4x32_10(off + (uint64_t)(i >> 2), stream_id, seed), v);
    const int j = (int)(i & 3);
    const float e = j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : v[3]));
    acc += expf(lv) + mu * mu - lv - 1.0f;
    eps_out[i] = e;
    z[i] = fmaf(expf(0.5f * lv), e, mu);
  }
  const float s = pgv_block_sum(acc, red);
  if (threadIdx.x == 0 && kl) atomicAdd(kl, 0.5f * kl_scale * s);
}

// ---- encoder head: nn.BatchNorm1d (train mode) + reparameterisation + Dkl in ONE launch per direction ------------
// (encoder.py:86-87 -> VAE.py:49-56 -> loss.py:57-66).  x[B][2D] is the Linear output; a workgroup owns ONE latent
// coordinate d = the channels {d, D + d} (mu and log-variance), a thread one row at a time: after the BatchNorm
// statistics of its two channels (float64, block-wide) every thread has both operands of z = mu + exp(lv / 2) * eps for
// its rows.  Two dependent launches of ~5-10 us per direction become one.  (A first form with 8 coordinates per
// workgroup - the layout of bn1d_fwd_kernel - ran 24 / 37 us: 8 workgroups, and every thread walked 16 rows of Philox
// rounds, logarithms and exponentials in sequence.)
// sums of N doubles over the block, result in every thread (smem: N * 4 doubles)
template <int N>
__device__ __forceinline__ void head_block_sums(double (&v)[N], double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = pgv_wave_sum_d(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) smem[k * 4 + wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = (smem[k * 4] + smem[k * 4 + 1]) + (smem[k * 4 + 2] + smem[k * 4 + 3]);
}

__global__ __launch_bounds__(256) void bn1d_reparam_fwd_kernel(
    const float* __restrict__ x, int B, int D, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
    int64_t* __restrict__ num_batches_tracked, float* __restrict__ y, float* __restrict__ scale_out,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, const uint64_t* __restrict__ rng, uint64_t stream_id,
    float kl_scale, float* __restrict__ z, float* __restrict__ eps_out, float* __restrict__ kl) {
  __shared__ double smem[16];
  __shared__ float red[16];
  const int C = 2 * D, d = blockIdx.x, tid = threadIdx.x;
  if (d == 0 && tid == 0 && num_batches_tracked) *num_batches_tracked += 1;
  double st[4] = {0.0, 0.0, 0.0, 0.0};   // sum, sum of squares of the mu channel; the same of the log-variance channel
  for (int b = tid; b < B; b += 256) {
    const float v0 = x[(int64_t)b * C + d], v1 = x[(int64_t)b * C + D + d];
    st[0] += (double)v0, st[1] = fma((double)v0, (double)v0, st[1]);
    st[2] += (double)v1, st[3] = fma((double)v1, (double)v1, st[3]);
  }
  head_block_sums<4>(st, smem);
  float sc[2], sh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = h * D + d;
    const double inv_n = 1.0 / (double)B, mean = st[2 * h] * inv_n;
    double var = st[2 * h + 1] * inv_n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
    sc[h] = (float)(g * rstd), sh[h] = (float)(bt - mean * g * rstd);
    if (tid == 0) {
      if (scale_out) scale_out[c] = sc[h];
      if (mean_out) mean_out[c] = (float)mean;
      if (rstd_out) rstd_out[c] = (float)rstd;
      const double unbias = B > 1 ? (double)B / (double)(B - 1) : 1.0;
      if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
      if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * unbias);
    }
  }
  const uint64_t seed = rng[0], off = rng[1];
  float acc = 0.f;
  for (int b = tid; b < B; b += 256) {
    const float mu = fmaf(x[(int64_t)b * C + d], sc[0], sh[0]), lv = fmaf(x[(int64_t)b * C + D + d], sc[1], sh[1]);
    y[(int64_t)b * C + d] = mu;
    y[(int64_t)b * C + D + d] = lv;
    const int64_t i = (int64_t)b * D + d;
    float nv[4];
    normal4(philox4x32_10(off + (uint64_t)(i >> 2), stream_id, seed), nv);
    const int j = (int)(i & 3);
    const float e = j == 0 ? nv[0] : (j == 1 ? nv[1] : (j == 2 ? nv[2] : nv[3]));
    acc += expf(lv) + mu * mu - lv - 1.0f;
    eps_out[i] = e;
    z[i] = fmaf(expf(0.5f * lv), e, mu);
  }
  const float tot = pgv_block_sum(acc, red);
  if (tid == 0 && kl) atomicAdd(kl, 0.5f * kl_scale * tot);
}

// Backward of the same: the gradient of [mu, lv] from g_z / g_kl (pgv_reparam_kl_bwd; + g_y, a gradient arriving at the
// BatchNorm output directly) formed on the fly, BatchNorm1d backward over it (pgv_bn1d_bwd), and the column sums of the
// result - the bias gradient of the Linear in front (sums of a train-mode BatchNorm backward: rounding noise around 0,
// which is what the reference's autograd hands Adam as well).
__global__ __launch_bounds__(256) void bn1d_reparam_bwd_kernel(
    const float* __restrict__ g_z, const float* __restrict__ g_kl, const float* __restrict__ g_y,
    const float* __restrict__ y, const float* __restrict__ epsv, const float* __restrict__ x,
    const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ rstd, int B, int D,
    float kl_scale, float* __restrict__ gx, float* __restrict__ ggamma, float* __restrict__ gbeta,
    float* __restrict__ colsum, int colsum_accumulate) {
  __shared__ double smem[16];
  const int C = 2 * D, d = blockIdx.x, tid = threadIdx.x;
  const float mu0 = mean[d], rs0 = rstd[d], sc0 = scale[d], mu1 = mean[D + d], rs1 = rstd[D + d], sc1 = scale[D + d];
  const float gk = g_kl ? g_kl[0] * kl_scale : 0.f;
  // gradients of y[b][d] (mu) and y[b][D + d] (log-variance)
  auto grads = [&](int b, float& g0, float& g1) {
    const float ym = y[(int64_t)b * C + d], yl = y[(int64_t)b * C + D + d];
    g0 = gk * ym;
    g1 = gk * 0.5f * (expf(yl) - 1.0f);
    if (g_z) {
      const float gz = g_z[(int64_t)b * D + d];
      g0 += gz;
      g1 = fmaf(gz * epsv[(int64_t)b * D + d], 0.5f * expf(0.5f * yl), g1);
    }
    if (g_y) g0 += g_y[(int64_t)b * C + d], g1 += g_y[(int64_t)b * C + D + d];
  };
  double st[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = tid; b < B; b += 256) {
    float g0, g1;
    grads(b, g0, g1);
    st[0] += (double)g0, st[1] += (double)(g0 * ((x[(int64_t)b * C + d] - mu0) * rs0));
    st[2] += (double)g1, st[3] += (double)(g1 * ((x[(int64_t)b * C + D + d] - mu1) * rs1));
  }
  head_block_sums<4>(st, smem);
  if (tid == 0) {
    if (ggamma) ggamma[d] = (float)st[1], ggamma[D + d] = (float)st[3];
    if (gbeta) gbeta[d] = (float)st[0], gbeta[D + d] = (float)st[2];
  }
  const float c10 = (float)(st[0] / (double)B), c20 = (float)(st[1] / (double)B);
  const float c11 = (float)(st[2] / (double)B), c21 = (float)(st[3] / (double)B);
  double cs[2] = {0.0, 0.0};
  for (int b = tid; b < B; b += 256) {
    float g0, g1;
    grads(b, g0, g1);
    const float o0 = sc0 * (g0 - c10 - (x[(int64_t)b * C + d] - mu0) * rs0 * c20);
    const float o1 = sc1 * (g1 - c11 - (x[(int64_t)b * C + D + d] - mu1) * rs1 * c21);
    gx[(int64_t)b * C + d] = o0;
    gx[(int64_t)b * C + D + d] = o1;
    cs[0] += (double)o0, cs[1] += (double)o1;
  }
  if (colsum) {
    head_block_sums<2>(cs, smem);
    if (tid == 0) {
      colsum[d] = (colsum_accumulate ? colsum[d] : 0.f) + (float)cs[0];
      colsum[D + d] = (colsum_accumulate ? colsum[D + d] : 0.f) + (float)cs[1];
    }
  }
}

__global__ void reparam_kl_bwd_kernel(const float* __restrict__ ml, const float* __restrict__ eps,
                                      const float* __restrict__ g_z, const float* __restrict__ g_kl, int B, int D,
                                      float kl_scale, float* __restrict__ g_ml) {
  const float gk = g_kl ? g_kl[0] * kl_scale : 0.f;
  const int64_t n = (int64_t)B * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / D;
    const int dd = (int)(i - b * D);
    const float mu = ml[(b * 2) * D + dd], lv = ml[(b * 2 + 1) * D + dd];
    float gmu = gk * mu, glv = gk * 0.5f * (expf(lv) - 1.0f);
    if (g_z) {
      const float gz = g_z[i];
      gmu += gz;
      if (eps) glv = fmaf(gz * eps[i], 0.5f * expf(0.5f * lv), glv);
    }
    g_ml[(b * 2) * D + dd] = gmu;
    g_ml[(b * 2 + 1) * D + dd] = glv;
  }
}

// ---- squared error -----------------------------------------------------------------------------------------
__global__ void sqerr_fwd_kernel(const float* __restrict__ xhat, const float* __restrict__ x, int64_t n, float scale,
                                 float* __restrict__ loss) {
  __shared__ float red[16];
  float a0 = 0.f, a1 = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n; i += 2 * stride) {
    const float d0 = xhat[i] - x[i], d1 = xhat[i + stride] - x[i + stride];
    a0 = fmaf(d0, d0, a0);
    a1 = fmaf(d1, d1, a1);
  }
  if (i < n) {
    const float d0 = xhat[i] - x[i];
    a0 = fmaf(d0, d0, a0);
  }
  const float s = pgv_block_sum(a0 + a1, red);
  if (threadIdx.x == 0) atomicAdd(loss, s * scale);
}

__global__ void sqerr_fwd4_kernel(const float4* __restrict__ xhat, const float4* __restrict__ x, int64_t n4, float scale,
                                  float* __restrict__ loss) {
  __shared__ float red[16];
  float a0 = 0.f, a1 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = xhat[i], q = x[i];
    const float d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z, d3 = p.w - q.w;
    a0 = fmaf(d0, d0, fmaf(d1, d1, a0));
    a1 = fmaf(d2, d2, fmaf(d3, d3, a1));
  }
  const float s = pgv_block_sum(a0 + a1, red);
  if (threadIdx.x == 0) atomicAdd(loss, s * scale);
}

__global__ void sqerr_bwd4_kernel(const float4* __restrict__ xhat, const float4* __restrict__ x,
                                  const float* __restrict__ g_loss, int64_t n4, float scale, int hardtanh,
                                  float4* __restrict__ g) {
  const float k = 2.0f * scale * (g_loss ? g_loss[0] : 1.0f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = xhat[i], q = x[i];
    float4 r = make_float4(k * (p.x - q.x), k * (p.y - q.y), k * (p.z - q.z), k * (p.w - q.w));
    if (hardtanh) {
      if (!(p.x > -1.0f && p.x < 1.0f)) r.x = 0.f;
      if (!(p.y > -1.0f && p.y < 1.0f)) r.y = 0.f;
      if (!(p.z > -1.0f && p.z < 1.0f)) r.z = 0.f;
      if (!(p.w > -1.0f && p.w < 1.0f)) r.w = 0.f;
    }
    g[i] = r;
  }
}

__global__ void sqerr_bwd_kernel(const float* __restrict__ xhat, const float* __restrict__ x,
                                 const float* __restrict__ g_loss, int64_t n, float scale, int hardtanh,
                                 float* __restrict__ g) {
  const float k = 2.0f * scale * (g_loss ? g_loss[0] : 1.0f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float xh = xhat[i];
    float v = k * (xh - x[i]);
    if (hardtanh && !(xh > -1.0f && xh < 1.0f)) v = 0.f;
    g[i] = v;
  }
}

// ---- Adam (coupled L2) ---------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, const float* __restrict__ hyper, float beta1,
                            float beta2, float eps, float wd) {
  const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2], gs = hyper[3];
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = fmaf(wd, pi, g[i] * gs);
    const float mi = fmaf(beta1, m[i], (1.0f - beta1) * gi);
    const float vi = fmaf(beta2, v[i], (1.0f - beta2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

__global__ void adam_tick_kernel(double* __restrict__ pows, float* __restrict__ hyper, double beta1, double beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double p1 = pows[0] * beta1, p2 = pows[1] * beta2;
    pows[0] = p1;
    pows[1] = p2;
    pows[2] += 1.0;  // the step count itself: beta1^t underflows after ~7000 steps and cannot be inverted for it
    hyper[1] = (float)(1.0 - p1);
    hyper[2] = (float)(1.0 - p2);
  }
}

// The single-thread bookkeeping of a train step as ONE launch in front of the Adam update (each was a dependent launch of
// ~4.7 us): Adam's step counter, the generator offset (all draws of the step are behind it), the reported total loss.
__global__ void step_tick_kernel(double* __restrict__ pows, float* __restrict__ hyper, double beta1, double beta2,
                                 uint64_t* __restrict__ rng_state, uint64_t rng_inc, const float* __restrict__ la,
                                 const float* __restrict__ lb, const float* __restrict__ wb,
                                 const float* __restrict__ lc, float* __restrict__ total, float* __restrict__ finite) {
  if (blockIdx.x != 0) return;
  if (threadIdx.x == 0) {
    const double p1 = pows[0] * beta1, p2 = pows[1] * beta2;
    pows[0] = p1;
    pows[1] = p2;
    pows[2] += 1.0;
    hyper[1] = (float)(1.0 - p1);
    hyper[2] = (float)(1.0 - p2);
  } else if (threadIdx.x == 1) {
    if (rng_state) rng_state[1] += rng_inc;
  } else if (threadIdx.x == 2) {
    if (total) {
      const float a = la[0], b = lb[0], c = lc ? lc[0] : 0.f, t = fmaf(b, wb[0], a) + c;
      total[0] = t;
      // (x - x is 0 for a finite x and NaN for NaN / +-Inf)
      if (finite) finite[0] = ((a - a) + (b - b) + (c - c) + (t - t)) == 0.f ? 1.f : 0.f;
    }
  }
}

__global__ void adam4_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                             float4* __restrict__ v, int64_t n4, const float* __restrict__ hyper, float beta1, float beta2,
                             float eps, float wd) {
  const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2], gs = hyper[3];
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    gi = fmaf(wd, pi, gi * gs);
    mi = fmaf(beta1, mi, (1.0f - beta1) * gi);
    vi = fmaf(beta2, vi, (1.0f - beta2) * gi * gi);
    pi = pi - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p[i], mm = m[i], vv = v[i];
    const float4 gg = g[i];
    upd(pp.x, gg.x, mm.x, vv.x);
    upd(pp.y, gg.y, mm.y, vv.y);
    upd(pp.z, gg.z, mm.z, vv.z);
    upd(pp.w, gg.w, mm.w, vv.w);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}

inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

int zero_scalar(float* p, hipStream_t st, const char* who) {
  hipError_t e = hipMemsetAsync(p, 0, sizeof(float), st);
  if (e != hipSuccess) {
    pgv_set_error("%s: memset failed: %s", who, hipGetErrorString(e));
    return PGV_E_LAUNCH;
  }
  return PGV_OK;
}

}  // namespace

extern "C" {

int pgv_dropout_mask(const uint64_t* rng_state, uint64_t stream_id, float p, int64_t n, float* mask, void* stream) {
  PGV_CHECK_ARG(rng_state && mask && n >= 0 && p >= 0.f && p < 1.f, "pgv_dropout_mask: bad argument");
  if (n == 0) return PGV_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, 4)), dim3(kBlock), 0, pgv_stream(stream), rng_state,
                     stream_id, p, 1.0f / (1.0f - p), n, mask);
  PGV_CHECK_LAUNCH("dropout_mask");
  return PGV_OK;
}

int pgv_dropout_apply(const uint64_t* rng_state, uint64_t stream_id, float p, int64_t n, const float* x, float* y,
                      float* mask, void* stream) {
  PGV_CHECK_ARG(rng_state && x && y && mask && n >= 0 && p >= 0.f && p < 1.f, "pgv_dropout_apply: bad argument");
  if (n == 0) return PGV_OK;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid_for(n, 4)), dim3(kBlock), 0, pgv_stream(stream), rng_state,
                     stream_id, p, 1.0f / (1.0f - p), n, x, y, mask, aligned16(x, y, mask) ? 1 : 0);
  PGV_CHECK_LAUNCH("dropout_apply");
  return PGV_OK;
}

static int dropout_fwd_launch(const uint64_t* rng_state, uint64_t stream_id, float p, const float* x, int64_t B, int C,
                              int64_t HW, const float* scale, const float* shift, const pgv_bn_src* bn, float* y,
                              uint64_t* saved_state, void* stream) {
  const int64_t n = B * C * HW;
  const int vec = aligned16(x, y, y) && (!scale || HW >= 4) ? 1 : 0;
  // (n == 0 still records the state: backward of an empty batch reads it)
  const int aff = !scale ? 0 : (C <= kDropBnMaxC ? 2 : 1);
  auto kern = aff == 0 ? dropout_fwd_kernel<0> : (aff == 1 ? dropout_fwd_kernel<1> : dropout_fwd_kernel<2>);
  hipLaunchKernelGGL(kern, dim3(grid_for(max(n, (int64_t)1), 4)), dim3(kBlock), 0, pgv_stream(stream), rng_state, stream_id,
                     p, 1.0f / (1.0f - p), n, x, scale, shift, C, HW, y, saved_state, vec,
                     (bn && aff == 2) ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("dropout_fwd");
  return PGV_OK;
}

int pgv_dropout_fwd(const uint64_t* rng_state, uint64_t stream_id, float p, const float* x, int64_t B, int C, int64_t HW,
                    const float* scale, const float* shift, float* y, uint64_t* saved_state, void* stream) {
  PGV_CHECK_ARG(rng_state && x && y && saved_state && B >= 0 && C > 0 && HW > 0 && p >= 0.f && p < 1.f &&
                    (scale == nullptr) == (shift == nullptr),
                "pgv_dropout_fwd: bad argument");
  return dropout_fwd_launch(rng_state, stream_id, p, x, B, C, HW, scale, shift, nullptr, y, saved_state, stream);
}

int pgv_dropout_fwd_bn(const uint64_t* rng_state, uint64_t stream_id, float p, const float* x, int64_t B, int C, int64_t HW,
                       const pgv_bn_src* bn, float* y, uint64_t* saved_state, void* stream) {
  PGV_CHECK_ARG(rng_state && x && y && saved_state && B >= 0 && C > 0 && HW > 0 && p >= 0.f && p < 1.f && bn && bn->stats &&
                    bn->scale && bn->shift && bn->n > 0,
                "pgv_dropout_fwd_bn: bad argument");
  if (C <= kDropBnMaxC)
    return dropout_fwd_launch(rng_state, stream_id, p, x, B, C, HW, bn->scale, bn->shift, bn, y, saved_state, stream);
  int rc = pgv_bn_finalize(bn->stats, C, bn->n, bn->gamma, bn->beta, bn->eps, bn->momentum, bn->running_mean,
                           bn->running_var, bn->num_batches_tracked, bn->scale, bn->shift, bn->mean, bn->rstd, stream);
  if (rc) return rc;
  return dropout_fwd_launch(rng_state, stream_id, p, x, B, C, HW, bn->scale, bn->shift, nullptr, y, saved_state, stream);
}

int pgv_dropout_bwd(const uint64_t* saved_state, uint64_t stream_id, float p, int64_t n, const float* gy, float* gx,
                    void* stream) {
  PGV_CHECK_ARG(saved_state && gy && gx && n >= 0 && p >= 0.f && p < 1.f, "pgv_dropout_bwd: bad argument");
  if (n == 0) return PGV_OK;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n, 4)), dim3(kBlock), 0, pgv_stream(stream), saved_state,
                     stream_id, p, 1.0f / (1.0f - p), n, gy, gx, aligned16(gy, gx, gx) ? 1 : 0);
  PGV_CHECK_LAUNCH("dropout_bwd");
  return PGV_OK;
}

int pgv_dropout_bwd_colsum(const uint64_t* saved_state, uint64_t stream_id, float p, int M, int N, const float* gy,
                           float* gx, float* colsum, int flags, void* stream) {
  PGV_CHECK_ARG(saved_state && gy && gx && colsum && M >= 0 && N > 0 && p >= 0.f && p < 1.f,
                "pgv_dropout_bwd_colsum: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (N % 4 == 0 && aligned16(gy, gx, gx) && M > 0) {
    if (!(flags & PGV_PREZEROED)) {
      if (hipMemsetAsync(colsum, 0, sizeof(float) * N, st) != hipSuccess) {
        pgv_set_error("pgv_dropout_bwd_colsum: hipMemsetAsync failed");
        return PGV_E_LAUNCH;
      }
    }
    constexpr int R = 8;
    // (values whose mask is 0 are multiplied, not selected, in dropout_bwd_kernel; here they are selected - the same
    // numbers unless gy holds infinities or NaNs under a dropped position)
    hipLaunchKernelGGL(dropout_bwd_colsum_kernel<R>, dim3((unsigned)pgv_cdiv(N / 4, 64), (unsigned)pgv_cdiv(M, 4 * R)),
                       dim3(256), 0, st, saved_state, stream_id, p, 1.0f / (1.0f - p), M, N / 4, (const float4*)gy,
                       (float4*)gx, colsum);
    PGV_CHECK_LAUNCH("dropout_bwd_colsum");
    return PGV_OK;
  }
  int rc = pgv_dropout_bwd(saved_state, stream_id, p, (int64_t)M * N, gy, gx, stream);
  if (rc) return rc;
  return pgv_colsum(gx, M, N, N, colsum, flags, stream);
}

int pgv_normal(const uint64_t* rng_state, uint64_t stream_id, int64_t n, float* out, void* stream) {
  PGV_CHECK_ARG(rng_state && out && n >= 0, "pgv_normal: bad argument");
  if (n == 0) return PGV_OK;
  hipLaunchKernelGGL(normal_kernel, dim3(grid_for(n, 4)), dim3(kBlock), 0, pgv_stream(stream), rng_state, stream_id,
                     n, out);
  PGV_CHECK_LAUNCH("normal");
  return PGV_OK;
}

int pgv_rng_advance(uint64_t* rng_state, uint64_t inc, void* stream) {
  PGV_CHECK_ARG(rng_state, "pgv_rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, pgv_stream(stream), rng_state, inc);
  PGV_CHECK_LAUNCH("rng_advance");
  return PGV_OK;
}

int pgv_mul(const float* x, const float* m, int64_t n, float* y, void* stream) {
  PGV_CHECK_ARG(x && m && y && n >= 0, "pgv_mul: bad argument");
  if (n == 0) return PGV_OK;
  int64_t done = 0;
  if (aligned16(x, m, y) && n >= 4) {
    done = n & ~(int64_t)3;
    hipLaunchKernelGGL(mul4_kernel, dim3(grid_for(done, 8)), dim3(kBlock), 0, pgv_stream(stream), (const float4*)x,
                       (const float4*)m, done / 4, (float4*)y);
  }
  if (done < n)
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n - done)), dim3(kBlock), 0, pgv_stream(stream), x + done, m + done,
                       n - done, y + done);
  PGV_CHECK_LAUNCH("mul");
  return PGV_OK;
}

int pgv_reparam_kl_fwd(const float* ml, const float* eps, int B, int D, float kl_scale, float* z, float* kl,
                       void* stream) {
  PGV_CHECK_ARG(ml && B >= 0 && D > 0, "pgv_reparam_kl_fwd: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (kl) {
    int rc = zero_scalar(kl, st, "pgv_reparam_kl_fwd");
    if (rc) return rc;
  }
  if (B == 0) return PGV_OK;
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(grid_for((int64_t)B * D, 1)), dim3(kBlock), 0, st, ml, eps, B, D,
                     kl_scale, z, kl);
  PGV_CHECK_LAUNCH("reparam_kl_fwd");
  return PGV_OK;
}

int pgv_reparam_kl_fwd_rng(const float* ml, const uint64_t* rng_state, uint64_t stream_id, int B, int D, float kl_scale,
                           float* z, float* eps_out, float* kl, int flags, void* stream) {
  PGV_CHECK_ARG(ml && rng_state && z && eps_out && B >= 0 && D > 0, "pgv_reparam_kl_fwd_rng: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (kl && !(flags & PGV_PREZEROED)) {
    int rc = zero_scalar(kl, st, "pgv_reparam_kl_fwd_rng");
    if (rc) return rc;
  }
  if (B == 0) return PGV_OK;
  hipLaunchKernelGGL(reparam_kl_fwd_rng_kernel, dim3(grid_for((int64_t)B * D, 1)), dim3(kBlock), 0, st, ml, rng_state,
                     stream_id, B, D, kl_scale, z, eps_out, kl);
  PGV_CHECK_LAUNCH("reparam_kl_fwd_rng");
  return PGV_OK;
}

int pgv_bn1d_reparam_fwd(const float* x, int B, int D, const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y, float* scale,
                         float* mean, float* rstd, const uint64_t* rng_state, uint64_t stream_id, float kl_scale,
                         float* z, float* eps_out, float* kl, int flags, void* stream) {
  PGV_CHECK_ARG(x && y && rng_state && z && eps_out && B > 0 && D > 0, "pgv_bn1d_reparam_fwd: bad argument");
  hipStream_t st = pgv_stream(stream);
  if (kl && !(flags & PGV_PREZEROED)) {
    int rc = zero_scalar(kl, st, "pgv_bn1d_reparam_fwd");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(bn1d_reparam_fwd_kernel, dim3((unsigned)D), dim3(256), 0, st, x, B, D, gamma, beta, eps,
                     momentum, running_mean, running_var, num_batches_tracked, y, scale, mean, rstd, rng_state, stream_id,
                     kl_scale, z, eps_out, kl);
  PGV_CHECK_LAUNCH("bn1d_reparam_fwd");
  return PGV_OK;
}

int pgv_bn1d_reparam_bwd(const float* g_z, const float* g_kl, const float* g_y, const float* y, const float* eps,
                         const float* x, const float* scale, const float* mean, const float* rstd, int B, int D,
                         float kl_scale, float* gx, float* ggamma, float* gbeta, float* colsum, int flags,
                         void* stream) {
  PGV_CHECK_ARG(y && x && scale && mean && rstd && gx && B > 0 && D > 0 && (!g_z || eps),
                "pgv_bn1d_reparam_bwd: bad argument");
  hipLaunchKernelGGL(bn1d_reparam_bwd_kernel, dim3((unsigned)D), dim3(256), 0, pgv_stream(stream), g_z, g_kl,
                     g_y, y, eps, x, scale, mean, rstd, B, D, kl_scale, gx, ggamma, gbeta, colsum,
                     (flags & PGV_PREZEROED) ? 1 : 0);
  PGV_CHECK_LAUNCH("bn1d_reparam_bwd");
  return PGV_OK;
}

int pgv_reparam_kl_bwd(const float* ml, const float* eps, const float* g_z, const float* g_kl, int B, int D,
                       float kl_scale, float* g_ml, void* stream) {
  PGV_CHECK_ARG(ml && g_ml && B >= 0 && D > 0, "pgv_reparam_kl_bwd: bad argument");
  if (B == 0) return PGV_OK;
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(grid_for((int64_t)B * D, 1)), dim3(kBlock), 0, pgv_stream(stream),
                     ml, eps, g_z, g_kl, B, D, kl_scale, g_ml);
  PGV_CHECK_LAUNCH("reparam_kl_bwd");
  return PGV_OK;
}

int pgv_sqerr_fwd(const float* xhat, const float* x, int64_t n, float scale, float* loss, void* stream) {
  PGV_CHECK_ARG(xhat && x && loss && n >= 0, "pgv_sqerr_fwd: bad argument");
  hipStream_t st = pgv_stream(stream);
  int rc = zero_scalar(loss, st, "pgv_sqerr_fwd");
  if (rc) return rc;
  if (n == 0) return PGV_OK;
  int64_t done = 0;
  if (aligned16(xhat, x) && n >= 4) {
    done = n & ~(int64_t)3;
    hipLaunchKernelGGL(sqerr_fwd4_kernel, dim3(grid_for(done, 16)), dim3(kBlock), 0, st, (const float4*)xhat,
                       (const float4*)x, done / 4, scale, loss);
  }
  if (done < n)
    hipLaunchKernelGGL(sqerr_fwd_kernel, dim3(grid_for(n - done, 8)), dim3(kBlock), 0, st, xhat + done, x + done,
                       n - done, scale, loss);
  PGV_CHECK_LAUNCH("sqerr_fwd");
  return PGV_OK;
}

int pgv_sqerr_bwd(const float* xhat, const float* x, const float* g_loss, int64_t n, float scale, int hardtanh,
                  float* g, void* stream) {
  PGV_CHECK_ARG(xhat && x && g && n >= 0, "pgv_sqerr_bwd: bad argument");
  if (n == 0) return PGV_OK;
  int64_t done = 0;
  if (aligned16(xhat, x, g) && n >= 4) {
    done = n & ~(int64_t)3;
    hipLaunchKernelGGL(sqerr_bwd4_kernel, dim3(grid_for(done, 8)), dim3(kBlock), 0, pgv_stream(stream),
                       (const float4*)xhat, (const float4*)x, g_loss, done / 4, scale, hardtanh, (float4*)g);
  }
  if (done < n)
    hipLaunchKernelGGL(sqerr_bwd_kernel, dim3(grid_for(n - done, 4)), dim3(kBlock), 0, pgv_stream(stream), xhat + done,
                       x + done, g_loss, n - done, scale, hardtanh, g + done);
  PGV_CHECK_LAUNCH("sqerr_bwd");
  return PGV_OK;
}

int pgv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                  float beta2, float eps, float weight_decay, void* stream) {
  PGV_CHECK_ARG(p && g && m && v && hyper && n >= 0, "pgv_adam_step: bad argument");
  if (n == 0) return PGV_OK;
  int64_t done = 0;
  if (aligned16(p, g, m, v) && n >= 4) {
    done = n & ~(int64_t)3;
    hipLaunchKernelGGL(adam4_kernel, dim3(grid_for(done, 8)), dim3(kBlock), 0, pgv_stream(stream), (float4*)p,
                       (const float4*)g, (float4*)m, (float4*)v, done / 4, hyper, beta1, beta2, eps, weight_decay);
  }
  if (done < n)
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n - done, 4)), dim3(kBlock), 0, pgv_stream(stream), p + done,
                       g + done, m + done, v + done, n - done, hyper, beta1, beta2, eps, weight_decay);
  PGV_CHECK_LAUNCH("adam_step");
  return PGV_OK;
}

int pgv_adam_tick(double* pows, float* hyper, float beta1, float beta2, void* stream) {
  PGV_CHECK_ARG(pows && hyper, "pgv_adam_tick: bad argument");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, pgv_stream(stream), pows, hyper, (double)beta1,
                     (double)beta2);
  PGV_CHECK_LAUNCH("adam_tick");
  return PGV_OK;
}

int pgv_step_tick(double* pows, float* hyper, float beta1, float beta2, uint64_t* rng_state, uint64_t rng_inc,
                  const float* loss_a, const float* loss_b, const float* weight_b, const float* loss_c, float* total,
                  float* finite, void* stream) {
  PGV_CHECK_ARG(pows && hyper && (!total || (loss_a && loss_b && weight_b)), "pgv_step_tick: bad argument");
  hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, pgv_stream(stream), pows, hyper, (double)beta1,
                     (double)beta2, rng_state, rng_inc, loss_a, loss_b, weight_b, loss_c, total, finite);
  PGV_CHECK_LAUNCH("step_tick");
  return PGV_OK;
}

int pgv_fill(float* p, int64_t n, float v, void* stream) {
  PGV_CHECK_ARG(p && n >= 0, "pgv_fill: bad argument");
  if (n == 0) return PGV_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(kBlock), 0, pgv_stream(stream), p, n, v);
  PGV_CHECK_LAUNCH("fill");
  return PGV_OK;
}

int pgv_axpy(int64_t n, float a, const float* x, float* y, void* stream) {
  PGV_CHECK_ARG(x && y && n >= 0, "pgv_axpy: bad argument");
  if (n == 0) return PGV_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(kBlock), 0, pgv_stream(stream), n, a, x, y);
  PGV_CHECK_LAUNCH("axpy");
  return PGV_OK;
}

int pgv_copy(const float* src, float* dst, int64_t n, void* stream) {
  PGV_CHECK_ARG(src && dst && n >= 0 && (n % 4) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                "pgv_copy: needs n%%4==0 and 16-byte aligned pointers");
  if (n == 0) return PGV_OK;
  hipLaunchKernelGGL(copy4_kernel, dim3(2048), dim3(kBlock), 0, pgv_stream(stream), (const float4*)src, (float4*)dst,
                     n / 4);
  PGV_CHECK_LAUNCH("copy");
  return PGV_OK;
}

}  // extern "C"
