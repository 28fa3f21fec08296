// Gather-GEMM convolutions for the DEEP layers of speccnn8l1_bn (enc5..enc8, dec1..dec4; model/encoder.py:249-255,
// :56-69, model/decoder.py:72-75,205-210): planes are tiny (9x12 ... 3x4 pixels) and channel counts large (128..2048),
// so the work is a genuine dense contraction whose cost is streaming the weights (up to 8 MB per layer).  Unlike the
// band kernels of conv_mfma.hip, the pixel dimension here is flattened over the whole minibatch (n = (b, oh, ow)), so
// every weight element fetched into LDS is reused by 64 pixels of many samples.
//
// 64x64 output tile per 256-thread workgroup (4 waves x one v_mfma_f32_32x32x2_f32 accumulator), K in slabs of 16:
//   DOWN  k4s2p2: slab = the 16 taps of one input channel;   1x1: slab = 16 input channels
//   UP    k4s2p2: per sub-pixel phase (grid.z), slab = 4 input channels x 4 taps;  1x1: 16 channels
//   WGRAD: M = cs, N = (cb, tap), K = (b, oh, ow) pixels in slabs of 16, split-K with float atomics.
// Operands are gathered element-wise (index decomposition hoisted out of the K loop), the producer's BatchNorm affine
// is applied in the gather (zero padding stays zero), epilogues match conv_mfma.hip (bias, activation, BN statistics).
#include "conv_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 64, BN = 64, BK = 16, LDP = 65;

struct Tiles {
  float As[BK][LDP];
  float Bs[BK][LDP];
};

__device__ __forceinline__ void mma_slab(const Tiles& t, f32x16& acc, int lane, int wm, int wn) {
#pragma unroll
  for (int kk = 0; kk < BK; kk += 2) {
    const float a = t.As[kk + (lane >> 5)][wm + (lane & 31)];
    const float b = t.Bs[kk + (lane >> 5)][wn + (lane & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
}

__device__ __forceinline__ float half32_sum(float v) {
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Epilogue shared by DOWN/UP: acc rows = output channel, cols = pixel.  `addr(n)` -> element offset of channel 0 of
// pixel n or -1; chan_stride = elements between channels.
template <typename AddrFn>
__device__ __forceinline__ void epilogue_store(const f32x16& acc, int lane, int c0, int C, int n, AddrFn addr,
                                               int64_t chan_stride, const float* __restrict__ bias, int act,
                                               float slope, float* __restrict__ out, double* __restrict__ stats) {
  const int64_t base = addr(n);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float v = 0.f;
    const bool ok = base >= 0 && c < C;
    if (ok) {
      v = pgv_act(acc[r] + (bias ? bias[c] : 0.f), act, slope);
      out[base + (int64_t)c * chan_stride] = v;
    }
    if (stats) {
      const float s = half32_sum(ok ? v : 0.f), q = half32_sum(ok ? v * v : 0.f);
      if ((lane & 31) == 0 && c < C) {
        atomicAdd(&stats[c], (double)s);
        atomicAdd(&stats[C + c], (double)q);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void conv_down_gemm_kernel(pgv_conv_desc d, const float* __restrict__ big,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int act, float slope,
                                                             float* __restrict__ out, double* __restrict__ stats) {
  __shared__ Tiles t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = d.Hs * d.Ws, N = d.B * P, HWb = d.Hb * d.Wb;
  const int n0 = blockIdx.x * BN, cs0 = blockIdx.y * BM;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  // B loader: pixel n_local, tap column / channel lane i0
  const int n_local = tid & 63, i0 = tid >> 6;
  const int n = n0 + n_local;
  const bool n_ok = n < N;
  const int nb = n_ok ? n / P : 0, pix = n_ok ? n - nb * P : 0;
  const int oh = pix / d.Ws, ow = pix - oh * d.Ws;
  const float* xb = big + (int64_t)nb * d.Cb * HWb;
  int offs[4];
  bool okv[4];
  if (KS == 4) {
    const int iw = ow * 2 - d.pad + i0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ih = oh * 2 - d.pad + j;
      okv[j] = n_ok && iw >= 0 && iw < d.Wb && ih >= 0 && ih < d.Hb;
      offs[j] = ih * d.Wb + iw;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      okv[j] = n_ok;
      offs[j] = pix;
    }
  }
  // A loader: k_local, rows q0 + 16 j
  const int k_local = tid & 15, q0 = tid >> 4;
  const int Kw = (KS == 4) ? d.Cb * 16 : d.Cb;  // weight row length
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nslab = (KS == 4) ? d.Cb : (d.Cb + 15) / 16;
  float ra[4], rb[4];
  auto load = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = cs0 + q0 + 16 * j;
      const int k = s * 16 + k_local;
      ra[j] = (m < d.Cs && k < Kw) ? w[(int64_t)m * Kw + k] : 0.f;
      const int cb = (KS == 4) ? s : s * 16 + i0 + 4 * j;
      float v = 0.f;
      if (okv[j] && cb < d.Cb) {
        v = xb[(int64_t)cb * HWb + offs[j]];
        if (in_scale) v = fmaf(v, in_scale[cb], in_shift[cb]);
      }
      rb[j] = v;
    }
  };
  load(0);
  for (int s = 0; s < nslab; ++s) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t.As[k_local][q0 + 16 * j] = pgv_opnd(ra[j], (d.flags & PGV_COMPUTE_BF16) != 0);
      t.Bs[(KS == 4) ? 4 * j + i0 : i0 + 4 * j][n_local] = pgv_opnd(rb[j], (d.flags & PGV_COMPUTE_BF16) != 0);
    }
    __syncthreads();
    if (s + 1 < nslab) load(s + 1);
    mma_slab(t, acc, lane, wm, wn);
  }
  const int ne = n0 + wn + (lane & 31);
  auto addr = [&](int nn) -> int64_t {
    if (nn >= N) return -1;
    const int bb = nn / P;
    return (int64_t)bb * d.Cs * P + (nn - bb * P);
  };
  epilogue_store(acc, lane, cs0 + wm, d.Cs, ne, addr, (int64_t)P, bias, act, slope, out, stats);
}

// ------------------------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void conv_up_gemm_kernel(pgv_conv_desc d, const float* __restrict__ small_in,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int act, float slope, float* __restrict__ out,
                                                           double* __restrict__ stats, int Hg, int Wg) {
  __shared__ Tiles t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ph = (KS == 4) ? (int)(blockIdx.z >> 1) : 0, pw = (KS == 4) ? (int)(blockIdx.z & 1) : 0;
  const int Pg = Hg * Wg, N = d.B * Pg, HWs = d.Hs * d.Ws;
  const int n0 = blockIdx.x * BN, cb0 = blockIdx.y * BM;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int n_local = tid & 63, i0 = tid >> 6;
  const int n = n0 + n_local;
  const bool n_ok = n < N;
  const int nb = n_ok ? n / Pg : 0, pix = n_ok ? n - nb * Pg : 0;
  const int u = pix / Wg, v = pix - u * Wg;
  const float* xb = small_in + (int64_t)nb * d.Cs * HWs;
  int offB = 0;
  bool okB = n_ok;
  if (KS == 4) {
    const int ih = u + 1 - (i0 >> 1), iw = v + 1 - (i0 & 1);
    okB = n_ok && ih >= 0 && ih < d.Hs && iw >= 0 && iw < d.Ws;
    offB = ih * d.Ws + iw;
  } else {
    offB = pix;
  }
  // A loader: m_local = cb, k lanes i0 (+4j)
  const int m_local = tid & 63;
  const int tapw = (KS == 4) ? (ph + 2 * (i0 >> 1)) * 4 + pw + 2 * (i0 & 1) : 0;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int per = (KS == 4) ? 4 : 16;  // input channels per slab
  const int nslab = (d.Cs + per - 1) / per;
  float ra[4], rb[4];
  auto load = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cs = (KS == 4) ? s * 4 + j : s * 16 + i0 + 4 * j;
      const int cb = cb0 + m_local;
      float a = 0.f, bv = 0.f;
      if (cs < d.Cs) {
        if (cb < d.Cb) a = (KS == 4) ? w[((int64_t)cs * d.Cb + cb) * 16 + tapw] : w[(int64_t)cs * d.Cb + cb];
        if (okB) {
          bv = xb[(int64_t)cs * HWs + offB];
          if (in_scale) bv = fmaf(bv, in_scale[cs], in_shift[cs]);
        }
      }
      ra[j] = a;
      rb[j] = bv;
    }
  };
  load(0);
  for (int s = 0; s < nslab; ++s) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int krow = (KS == 4) ? 4 * j + i0 : i0 + 4 * j;
      t.As[krow][m_local] = pgv_opnd(ra[j], (d.flags & PGV_COMPUTE_BF16) != 0);
      t.Bs[krow][n_local] = pgv_opnd(rb[j], (d.flags & PGV_COMPUTE_BF16) != 0);
    }
    __syncthreads();
    if (s + 1 < nslab) load(s + 1);
    mma_slab(t, acc, lane, wm, wn);
  }
  const int ne = n0 + wn + (lane & 31);
  const int HWb = d.Hb * d.Wb;
  auto addr = [&](int nn) -> int64_t {
    if (nn >= N) return -1;
    const int bb = nn / Pg, pp = nn - bb * Pg;
    const int uu = pp / Wg, vv = pp - uu * Wg;
    const int ih = (KS == 4) ? 2 * uu + ph : uu, iw = (KS == 4) ? 2 * vv + pw : vv;
    if (ih >= d.Hb || iw >= d.Wb) return -1;
    return (int64_t)bb * d.Cb * HWb + (int64_t)ih * d.Wb + iw;
  };
  epilogue_store(acc, lane, cb0 + wm, d.Cb, ne, addr, (int64_t)HWb, bias, act, slope, out, stats);
}

// ------------------------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void conv_wgrad_gemm_kernel(pgv_conv_desc d, const float* __restrict__ big,
                                                              const float* __restrict__ big_scale,
                                                              const float* __restrict__ big_shift,
                                                              const float* __restrict__ small_in,
                                                              const float* __restrict__ small_scale,
                                                              const float* __restrict__ small_shift,
                                                              float* __restrict__ gw, int k_per_split) {
  __shared__ Tiles t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = d.Hs * d.Ws, NP = d.B * P, HWb = d.Hb * d.Wb;
  constexpr int KK = KS * KS;
  const int Nw = d.Cb * KK;  // gw row length
  const int nn0 = blockIdx.x * BN, cs0 = blockIdx.y * BM;
  const int kbeg = blockIdx.z * k_per_split, kend = min(NP, kbeg + k_per_split);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int k_local = tid & 15, q0 = tid >> 4;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float ra[4], rb[4];
  auto load = [&](int k0) {
    const int p = k0 + k_local;
    const bool p_ok = p < kend;
    const int b = p_ok ? p / P : 0, pix = p_ok ? p - b * P : 0;
    const int oh = pix / d.Ws, ow = pix - oh * d.Ws;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cs = cs0 + q0 + 16 * j;
      float a = 0.f;
      if (p_ok && cs < d.Cs) {
        a = small_in[((int64_t)b * d.Cs + cs) * P + pix];
        if (small_scale) a = fmaf(a, small_scale[cs], small_shift[cs]);
      }
      ra[j] = a;
      const int nn = nn0 + q0 + 16 * j;  // (cb, tap)
      float bv = 0.f;
      if (p_ok && nn < Nw) {
        const int cb = (KS == 4) ? nn >> 4 : nn;
        const int tap = (KS == 4) ? nn & 15 : 0;
        const int ih = (KS == 4) ? oh * 2 - d.pad + (tap >> 2) : oh, iw = (KS == 4) ? ow * 2 - d.pad + (tap & 3) : ow;
        if (ih >= 0 && ih < d.Hb && iw >= 0 && iw < d.Wb) {
          bv = big[((int64_t)b * d.Cb + cb) * HWb + ih * d.Wb + iw];
          if (big_scale) bv = fmaf(bv, big_scale[cb], big_shift[cb]);
        }
      }
      rb[j] = bv;
    }
  };
  if (kbeg < kend) load(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t.As[k_local][q0 + 16 * j] = pgv_opnd(ra[j], (d.flags & PGV_COMPUTE_BF16) != 0);
      t.Bs[k_local][q0 + 16 * j] = pgv_opnd(rb[j], (d.flags & PGV_COMPUTE_BF16) != 0);
    }
    __syncthreads();
    if (k0 + BK < kend) load(k0 + BK);
    mma_slab(t, acc, lane, wm, wn);
  }
  const int nn = nn0 + wn + (lane & 31);
  if (nn < Nw) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cs = cs0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (cs < d.Cs) atomicAdd(&gw[(int64_t)cs * Nw + nn], acc[r]);
    }
  }
}

bool shape_k4(const pgv_conv_desc* d) { return d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 2; }
bool shape_k1(const pgv_conv_desc* d) { return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0; }

}  // namespace

int pgv_conv_down_gemm(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                       const float* w, const float* bias, int act, float slope, float* out, double* stats,
                       hipStream_t st) {
  const bool k4 = shape_k4(d), k1 = shape_k1(d);
  if (!k4 && !k1) return 0;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_gemm: memset failed");
    return PGV_E_LAUNCH;
  }
  const int64_t N = (int64_t)d->B * d->Hs * d->Ws;
  dim3 grid((unsigned)pgv_cdiv(N, BN), (unsigned)pgv_cdiv(d->Cs, BM));
  if (k4)
    hipLaunchKernelGGL(conv_down_gemm_kernel<4>, grid, dim3(256), 0, st, *d, big, in_scale, in_shift, w, bias, act,
                       slope, out, stats);
  else
    hipLaunchKernelGGL(conv_down_gemm_kernel<1>, grid, dim3(256), 0, st, *d, big, in_scale, in_shift, w, bias, act,
                       slope, out, stats);
  PGV_CHECK_LAUNCH("conv_down_gemm");
  return 1;
}

int pgv_conv_up_gemm(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* out, double* stats,
                     hipStream_t st) {
  const bool k4 = shape_k4(d), k1 = shape_k1(d);
  if (!k4 && !k1) return 0;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_gemm: memset failed");
    return PGV_E_LAUNCH;
  }
  const int Hg = k4 ? (d->Hb + 1) / 2 : d->Hb, Wg = k4 ? (d->Wb + 1) / 2 : d->Wb;
  const int64_t N = (int64_t)d->B * Hg * Wg;
  dim3 grid((unsigned)pgv_cdiv(N, BN), (unsigned)pgv_cdiv(d->Cb, BM), k4 ? 4 : 1);
  if (k4)
    hipLaunchKernelGGL(conv_up_gemm_kernel<4>, grid, dim3(256), 0, st, *d, small_in, in_scale, in_shift, w, bias, act,
                       slope, out, stats, Hg, Wg);
  else
    hipLaunchKernelGGL(conv_up_gemm_kernel<1>, grid, dim3(256), 0, st, *d, small_in, in_scale, in_shift, w, bias, act,
                       slope, out, stats, Hg, Wg);
  PGV_CHECK_LAUNCH("conv_up_gemm");
  return 1;
}

int pgv_conv_wgrad_gemm(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                        const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                        hipStream_t st) {
  const bool k4 = shape_k4(d), k1 = shape_k1(d);
  if (!k4 && !k1) return 0;
  const int KK = d->kh * d->kw;
  const int64_t Nw = (int64_t)d->Cb * KK;
  if (!(d->flags & PGV_PREZEROED) && hipMemsetAsync(gw, 0, sizeof(float) * (size_t)d->Cs * Nw, st) != hipSuccess) {
    pgv_set_error("conv_wgrad_gemm: memset failed");
    return PGV_E_LAUNCH;
  }
  const int64_t NP = (int64_t)d->B * d->Hs * d->Ws;
  if (NP == 0) return 1;
  const int64_t tiles = pgv_cdiv(Nw, BN) * pgv_cdiv(d->Cs, BM);
  int splits = (int)max((int64_t)1, min(pgv_cdiv(1024, tiles), pgv_cdiv(NP, BK * 8)));
  const int k_per_split = (int)(pgv_cdiv(pgv_cdiv(NP, splits), BK) * BK);
  splits = (int)pgv_cdiv(NP, k_per_split);
  dim3 grid((unsigned)pgv_cdiv(Nw, BN), (unsigned)pgv_cdiv(d->Cs, BM), (unsigned)splits);
  if (k4)
    hipLaunchKernelGGL(conv_wgrad_gemm_kernel<4>, grid, dim3(256), 0, st, *d, big, big_scale, big_shift, small_in,
                       small_scale, small_shift, gw, k_per_split);
  else
    hipLaunchKernelGGL(conv_wgrad_gemm_kernel<1>, grid, dim3(256), 0, st, *d, big, big_scale, big_shift, small_in,
                       small_scale, small_shift, gw, k_per_split);
  PGV_CHECK_LAUNCH("conv_wgrad_gemm");
  return 1;
}
