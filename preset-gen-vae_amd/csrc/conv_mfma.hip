// Implicit-GEMM convolutions on the gfx950 f32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fma chain, same
// peak as the fp32 VALU but one instruction retires 1024 MACs from two VGPR operands).
//
// Stride-2 / pad-2 layers of the reference tables (model/encoder.py:241-255, model/decoder.py:205-218; k = 4 or 5).
//
//  * DOWN  (Conv2d forward, ConvTranspose2d input-gradient):  D[cs][pixel] = sum_k W[cs][k] * X[k][pixel],
//      k = (cb, kh, kw).  One MFMA consumes the 4 kw taps of one (cb, kh): the B operand of lane (pixel j, kw) is read
//      straight out of the RAW input tile in LDS at  2*row*Wt + 2*col + kw  (no im2col is ever materialised), the A
//      operand is the weight tile transposed to [k][cs].
//  * UP    (ConvTranspose2d forward, Conv2d input-gradient) by sub-pixel phases: output pixel (2u+ph, 2v+pw) only
//      sees taps kh = ph+2*th, kw = pw+2*tw, reading input (u+1-th, v+1-tw) — the SAME input gather for all four
//      phases.  So  D[(cb,ph,pw)][(u,v)] = sum_k W'[(cb,ph,pw)][k] * X[k][(u,v)],  k = (cs, th, tw): M = 4*Cb rows, one
//      MFMA per (cs, th-group); each lane ends up with the 2x2 output block of its (u,v) for one channel and stores
//      it as two float2 rows.
//
// A workgroup (4 waves) owns a full-width band of output rows of one sample, so every global row it touches is a
// contiguous NCHW segment; the band's input rows (+halo, zero padding, and the producer's BatchNorm affine folded in)
// are staged once into LDS, weights are staged per channel chunk, each wave owns NT pixel tiles x all MT channel
// tiles of accumulators.  Epilogue: bias + LeakyReLU/Hardtanh, store, and per-channel sum / sum-of-squares partials
// (wave shuffle -> LDS -> one float64 atomic per channel per workgroup) for the BatchNorm that follows.
#include "conv_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxLds = 160 * 1024;

__device__ __forceinline__ float group16_sum(float v) {
  // sum over the 16 lanes that share lane>>4 (xor butterflies stay inside the group)
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// DOWN
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int MT, int NT, int CK>
__global__ __launch_bounds__(256) void conv_down_mfma_kernel(pgv_conv_desc d, const float* __restrict__ big,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int act, float slope,
                                                             float* __restrict__ out, double* __restrict__ stats,
                                                             int R, int Wt) {
  constexpr int KWS = (KS + 3) / 4;  // MFMA k-groups along kw
  constexpr int KWP = KWS * 4;       // kw padded to a multiple of 4 (zero weights beyond KS)
  constexpr int CSP = MT * 16 + 1;   // padded [k][cs] weight row: odd stride => conflict-free transposing writes
  constexpr int KC = CK * KS * KWP;  // k rows per channel chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rows_in = 2 * (R - 1) + KS;
  const int plane = rows_in * Wt;
  float* in_tile = lds;
  float* w_tile = in_tile + CK * plane;
  float* st_tile = w_tile + KC * CSP;  // [4 waves][MT*16][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int oh0 = blockIdx.x * R;
  const int rows_out = min(R, d.Hs - oh0);
  const int Pb = rows_out * d.Ws;
  const int ih0 = oh0 * 2 - d.pad;

  int offB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wave * NT + t) * 16 + (lane & 15);
    const int pv = p < Pb ? p : 0;
    const int r = pv / d.Ws, c = pv - r * d.Ws;
    offB[t] = 2 * r * Wt + 2 * c + (lane >> 4);
  }
  const int offA = (lane >> 4) * CSP + (lane & 15);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int cb0 = 0; cb0 < d.Cb; cb0 += CK) {
    __syncthreads();
    for (int row_id = wave; row_id < CK * rows_in; row_id += 4) {
      const int c = row_id / rows_in, rr = row_id - c * rows_in;
      const int cb = cb0 + c, ih = ih0 + rr;
      const bool row_ok = cb < d.Cb && ih >= 0 && ih < d.Hb;
      float sc = 1.f, sh = 0.f;
      if (in_scale && cb < d.Cb) {
        sc = in_scale[cb];
        sh = in_shift[cb];
      }
      const float* src = big + (((int64_t)b * d.Cb + (row_ok ? cb : 0)) * d.Hb + (row_ok ? ih : 0)) * d.Wb;
      float* dst = in_tile + c * plane + rr * Wt;
      for (int cc = lane; cc < Wt; cc += 64) {
        const int iw = cc - d.pad;
        float v = 0.f;
        if (row_ok && iw >= 0 && iw < d.Wb) v = fmaf(src[iw], sc, sh);
        dst[cc] = v;
      }
    }
    for (int idx = tid; idx < KC * MT * 16; idx += 256) {
      const int cs = idx / KC, k = idx - cs * KC;
      const int c = k / (KS * KWP), rem = k - c * (KS * KWP);
      const int kh = rem / KWP, kw = rem - kh * KWP;
      float v = 0.f;
      if (cs < d.Cs && cb0 + c < d.Cb && kw < KS) v = w[(((int64_t)cs * d.Cb + cb0 + c) * KS + kh) * KS + kw];
      w_tile[k * CSP + cs] = v;
    }
    __syncthreads();
    for (int c = 0; c < CK; ++c) {
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
        for (int kws = 0; kws < KWS; ++kws) {
          const int kidx = ((c * KS + kh) * KWS + kws) * 4;
          float a[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = w_tile[kidx * CSP + offA + m * 16];
          const float* bp = in_tile + c * plane + kh * Wt + kws * 4;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float bv = bp[offB[t]];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv, acc[m][t], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue.  D layout: col = lane&15 (pixel), row = (lane>>4)*4 + reg (channel inside the M tile)
  const int64_t out_base = (int64_t)b * d.Cs * d.Hs * d.Ws + (int64_t)oh0 * d.Ws;
  const int64_t cstride = (int64_t)d.Hs * d.Ws;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int cs = m * 16 + (lane >> 4) * 4 + reg;
      if (cs < d.Cs) {
        const float bv = bias ? bias[cs] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int p = (wave * NT + t) * 16 + (lane & 15);
          if (p < Pb) {
            const float v = pgv_act(acc[m][t][reg] + bv, act, slope);
            out[out_base + cs * cstride + p] = v;
            s[reg] += v;
            q[reg] = fmaf(v, v, q[reg]);
          }
        }
      }
    }
    if (stats) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float ss = group16_sum(s[reg]), qq = group16_sum(q[reg]);
        if ((lane & 15) == 0) {
          const int cl = m * 16 + (lane >> 4) * 4 + reg;
          st_tile[(wave * MT * 16 + cl) * 2 + 0] = ss;
          st_tile[(wave * MT * 16 + cl) * 2 + 1] = qq;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    if (tid < MT * 16 && tid < d.Cs) {
      double ss = 0.0, qq = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        ss += (double)st_tile[(wv * MT * 16 + tid) * 2 + 0];
        qq += (double)st_tile[(wv * MT * 16 + tid) * 2 + 1];
      }
      atomicAdd(&stats[tid], ss);
      atomicAdd(&stats[d.Cs + tid], qq);
    }
  }
}

struct DownPlan {
  int R, Wt;
  size_t lds_bytes;
};

template <int KS, int MT, int NT, int CK>
bool plan_down(const pgv_conv_desc* d, DownPlan* pl) {
  constexpr int KWS = (KS + 3) / 4, KWP = KWS * 4, CSP = MT * 16 + 1, KC = CK * KS * KWP;
  const int Wt = max(d->Wb + 2 * d->pad, 2 * (d->Ws - 1) + KWP);
  int R = min(d->Hs, (64 * NT) / d->Ws);
  while (R >= 1) {
    const size_t bytes = sizeof(float) * ((size_t)CK * (2 * (R - 1) + KS) * Wt + (size_t)KC * CSP + 4 * MT * 16 * 2);
    if (bytes <= 72 * 1024 || (R == 1 && bytes <= (size_t)kMaxLds)) {
      pl->R = R;
      pl->Wt = Wt;
      pl->lds_bytes = bytes;
      return true;
    }
    --R;
  }
  return false;
}

template <int KS, int MT, int NT, int CK>
int launch_down(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift, const float* w,
                const float* bias, int act, float slope, float* out, double* stats, hipStream_t st) {
  DownPlan pl;
  if (!plan_down<KS, MT, NT, CK>(d, &pl)) return 0;
  auto kern = conv_down_mfma_kernel<KS, MT, NT, CK>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
      pgv_set_error("conv_down_mfma: cannot raise the dynamic LDS limit");
      return PGV_E_LAUNCH;
    }
    attr_done = true;
  }
  if (stats) {
    if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
      pgv_set_error("conv_down_mfma: memset failed");
      return PGV_E_LAUNCH;
    }
  }
  dim3 grid((unsigned)pgv_cdiv(d->Hs, pl.R), (unsigned)d->B);
  hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds_bytes, st, *d, big, in_scale, in_shift, w, bias, act, slope, out,
                     stats, pl.R, pl.Wt);
  PGV_CHECK_LAUNCH("conv_down_mfma");
  return 1;
}


// ---------------------------------------------------------------------------------------------------------------
// UP  (sub-pixel phases; see the header comment)
// ---------------------------------------------------------------------------------------------------------------
// KS = 4: T = 2 taps per axis, one MFMA k-group per input channel with k = th*2 + tw.
// KS = 5: T = 3 taps per axis, three k-groups per input channel (group = th) with k = tw padded to 4 (zero weight).
template <int KS, int MT, int NT, int CK>
__global__ __launch_bounds__(256) void conv_up_mfma_kernel(pgv_conv_desc d, const float* __restrict__ small_in,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int act, float slope, float* __restrict__ out,
                                                           double* __restrict__ stats, int R, int Wg, int Hg) {
  constexpr int T = (KS + 1) / 2;         // taps per axis
  constexpr int KG = (KS == 4) ? 1 : T;   // MFMA k-groups per input channel
  constexpr int TWR = (KS == 4) ? 2 : 4;  // tw values a k-group reads along the row
  constexpr int MSP = MT * 16 + 1;        // padded [k][m] weight row
  constexpr int KC = CK * KG * 4;         // k rows per channel chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rows_in = R + T - 1;
  const int Wt = Wg + TWR - 1;
  const int plane = rows_in * Wt;
  float* in_tile = lds;
  float* w_tile = in_tile + CK * plane;
  float* st_tile = w_tile + KC * MSP;  // [4 waves][MT*4][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int u0 = blockIdx.x * R;
  const int rows_g = min(R, Hg - u0);
  const int Pb = rows_g * Wg;
  const int ih0 = u0 + 2 - T;    // input row of local row 0
  const int iw0 = 2 - TWR;       // input col of local col 0

  int offB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wave * NT + t) * 16 + (lane & 15);
    const int pv = p < Pb ? p : 0;
    const int ur = pv / Wg, v = pv - ur * Wg;
    const int k = lane >> 4;
    if (KS == 4)
      offB[t] = (ur + 1 - (k >> 1)) * Wt + v + 1 - (k & 1);
    else
      offB[t] = (ur + T - 1) * Wt + v + (TWR - 1) - k;
  }
  const int offA = (lane >> 4) * MSP + (lane & 15);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int M = 4 * d.Cb;
  for (int cs0 = 0; cs0 < d.Cs; cs0 += CK) {
    __syncthreads();
    for (int row_id = wave; row_id < CK * rows_in; row_id += 4) {
      const int c = row_id / rows_in, rr = row_id - c * rows_in;
      const int cs = cs0 + c, ih = ih0 + rr;
      const bool row_ok = cs < d.Cs && ih >= 0 && ih < d.Hs;
      float sc = 1.f, sh = 0.f;
      if (in_scale && cs < d.Cs) {
        sc = in_scale[cs];
        sh = in_shift[cs];
      }
      const float* src = small_in + (((int64_t)b * d.Cs + (row_ok ? cs : 0)) * d.Hs + (row_ok ? ih : 0)) * d.Ws;
      float* dst = in_tile + c * plane + rr * Wt;
      for (int cc = lane; cc < Wt; cc += 64) {
        const int iw = iw0 + cc;
        float v = 0.f;
        if (row_ok && iw >= 0 && iw < d.Ws) v = fmaf(src[iw], sc, sh);
        dst[cc] = v;
      }
    }
    // weights: w_tile[(c*KG + g)*4 + kk][m], m = cb*4 + ph*2 + pw, value w[cs][cb][ph+2th][pw+2tw]
    for (int idx = tid; idx < KC * MT * 16; idx += 256) {
      const int krow = idx / (MT * 16), m = idx - krow * (MT * 16);
      const int c = krow / (KG * 4), rem = krow - c * (KG * 4);
      const int g = rem >> 2, kk = rem & 3;
      const int th = (KS == 4) ? (kk >> 1) : g, tw = (KS == 4) ? (kk & 1) : kk;
      const int cb = m >> 2, ph = (m >> 1) & 1, pw = m & 1;
      const int kh = ph + 2 * th, kw = pw + 2 * tw;
      float v = 0.f;
      if (m < M && cs0 + c < d.Cs && kh < KS && kw < KS)
        v = w[(((int64_t)(cs0 + c) * d.Cb + cb) * KS + kh) * KS + kw];
      w_tile[krow * MSP + m] = v;
    }
    __syncthreads();
    for (int c = 0; c < CK; ++c) {
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const int kidx = (c * KG + g) * 4;
        float a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = w_tile[kidx * MSP + offA + m * 16];
        const float* bp = in_tile + c * plane - ((KS == 4) ? 0 : g * Wt);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float bv = bp[offB[t]];
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv, acc[m][t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: lane owns channel cb = mt*4 + (lane>>4) at grid pixel (u,v); regs = (ph,pw)
  const bool vec2 = (d.Wb & 1) == 0;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int cb = m * 4 + (lane >> 4);
    float s = 0.f, q = 0.f;
    if (cb < d.Cb) {
      const float bv = bias ? bias[cb] : 0.f;
      float* obase = out + ((int64_t)b * d.Cb + cb) * d.Hb * d.Wb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p = (wave * NT + t) * 16 + (lane & 15);
        if (p < Pb) {
          const int ur = p / Wg, v = p - ur * Wg;
          const int ow = 2 * v;
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
            const int oh = 2 * (u0 + ur) + ph;
            if (oh < d.Hb) {
              const float v0 = pgv_act(acc[m][t][ph * 2 + 0] + bv, act, slope);
              const float v1 = pgv_act(acc[m][t][ph * 2 + 1] + bv, act, slope);
              float* o = obase + (int64_t)oh * d.Wb + ow;
              if (ow + 1 < d.Wb) {
                if (vec2)
                  *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                else {
                  o[0] = v0;
                  o[1] = v1;
                }
                s += v0 + v1;
                q = fmaf(v0, v0, fmaf(v1, v1, q));
              } else if (ow < d.Wb) {
                o[0] = v0;
                s += v0;
                q = fmaf(v0, v0, q);
              }
            }
          }
        }
      }
    }
    if (stats) {
      const float ss = group16_sum(s), qq = group16_sum(q);
      if ((lane & 15) == 0) {
        st_tile[(wave * MT * 4 + m * 4 + (lane >> 4)) * 2 + 0] = ss;
        st_tile[(wave * MT * 4 + m * 4 + (lane >> 4)) * 2 + 1] = qq;
      }
    }
  }
  if (stats) {
    __syncthreads();
    if (tid < MT * 4 && tid < d.Cb) {
      double ss = 0.0, qq = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        ss += (double)st_tile[(wv * MT * 4 + tid) * 2 + 0];
        qq += (double)st_tile[(wv * MT * 4 + tid) * 2 + 1];
      }
      atomicAdd(&stats[tid], ss);
      atomicAdd(&stats[d.Cb + tid], qq);
    }
  }
}

template <int KS, int MT, int NT, int CK>
int launch_up(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
              const float* w, const float* bias, int act, float slope, float* out, double* stats, hipStream_t st) {
  constexpr int T = (KS + 1) / 2, KG = (KS == 4) ? 1 : T, TWR = (KS == 4) ? 2 : 4, MSP = MT * 16 + 1,
                KC = CK * KG * 4;
  const int Hg = (d->Hb + 1) / 2, Wg = (d->Wb + 1) / 2;
  int R = min(Hg, (64 * NT) / Wg);
  if (R < 1) return 0;
  size_t bytes = 0;
  for (; R >= 1; --R) {
    bytes = sizeof(float) * ((size_t)CK * (R + T - 1) * (Wg + TWR - 1) + (size_t)KC * MSP + 4 * MT * 4 * 2);
    if (bytes <= 72 * 1024 || R == 1) break;
  }
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = conv_up_mfma_kernel<KS, MT, NT, CK>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
      pgv_set_error("conv_up_mfma: cannot raise the dynamic LDS limit");
      return PGV_E_LAUNCH;
    }
    attr_done = true;
  }
  if (stats) {
    if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
      pgv_set_error("conv_up_mfma: memset failed");
      return PGV_E_LAUNCH;
    }
  }
  dim3 grid((unsigned)pgv_cdiv(Hg, R), (unsigned)d->B);
  hipLaunchKernelGGL(kern, grid, dim3(256), bytes, st, *d, small_in, in_scale, in_shift, w, bias, act, slope, out,
                     stats, R, Wg, Hg);
  PGV_CHECK_LAUNCH("conv_up_mfma");
  return 1;
}


// ---------------------------------------------------------------------------------------------------------------
// WGRAD:  gw[cs][cb][kh][kw] = sum_{b,oh,ow} small[b,cs,oh,ow] * big[b,cb,2oh-2+kh,2ow-2+kw]
// GEMM with M = cs, N = (cb, tap) — one 16-wide N tile per (cb, 16 taps) — and K = output pixels, 4 consecutive ow
// per MFMA.  A[cs][pixel] comes from the small tile, B[pixel][tap] again straight from the raw big tile:
// lane (tap j, pixel k) reads  cb*plane + (2r+kh)*Wt + 2*(ow0+k) + kw.
// Workgroups are persistent over (sample, band) units: accumulators stay in registers across units and are flushed
// once at the end with float atomics (Cs*Cb*k*k values per wave), so atomic traffic is grid-size x weight-size, not
// unit-count x weight-size.  Waves split the N tiles WN ways and the pixel steps 4/WN ways.
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int MT, int NB, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(pgv_conv_desc d, const float* __restrict__ big,
                                                              const float* __restrict__ big_scale,
                                                              const float* __restrict__ big_shift,
                                                              const float* __restrict__ small_in,
                                                              const float* __restrict__ small_scale,
                                                              const float* __restrict__ small_shift,
                                                              float* __restrict__ gw, int R, int Wt, int WsP, int SP,
                                                              int bands, int units) {
  constexpr int KK = KS * KS;
  constexpr int NTAP_T = (KK + 15) / 16;  // N tiles per big channel
  constexpr int WK = 4 / WN;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rows_in = 2 * (R - 1) + KS;
  const int plane = rows_in * Wt;
  float* big_tile = lds;                      // [Cb][rows_in][Wt]
  float* small_tile = lds + d.Cb * plane;     // [Cs][SP]  (SP >= R*WsP, SP % 32 == 2)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gn = wave % WN, wk = wave / WN;
  const int n_tiles = d.Cb * NTAP_T;

  int offB[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    int nt = gn * NB + n;
    if (nt >= n_tiles) nt = 0;
    const int cb = nt / NTAP_T;
    const int tau = (nt - cb * NTAP_T) * 16 + (lane & 15);
    const int kh = tau < KK ? tau / KS : 0, kw = tau < KK ? tau - (tau / KS) * KS : 0;
    offB[n] = cb * plane + kh * Wt + kw + 2 * (lane >> 4);
  }
  int offA[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) offA[m] = min(m * 16 + (lane & 15), d.Cs - 1) * SP + (lane >> 4);

  f32x4 acc[MT][NB];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int steps_per_row = WsP / 4;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int b = u / bands, band = u - b * bands;
    const int oh0 = band * R;
    const int rows_out = min(R, d.Hs - oh0);
    const int ih0 = oh0 * 2 - d.pad;
    __syncthreads();
    for (int row_id = wave; row_id < d.Cb * rows_in; row_id += 4) {
      const int c = row_id / rows_in, rr = row_id - c * rows_in;
      const int ih = ih0 + rr;
      const bool row_ok = ih >= 0 && ih < d.Hb;
      float sc = 1.f, sh = 0.f;
      if (big_scale) {
        sc = big_scale[c];
        sh = big_shift[c];
      }
      const float* src = big + (((int64_t)b * d.Cb + c) * d.Hb + (row_ok ? ih : 0)) * d.Wb;
      float* dst = big_tile + c * plane + rr * Wt;
      for (int cc = lane; cc < Wt; cc += 64) {
        const int iw = cc - d.pad;
        float v = 0.f;
        if (row_ok && iw >= 0 && iw < d.Wb) v = fmaf(src[iw], sc, sh);
        dst[cc] = v;
      }
    }
    for (int row_id = wave; row_id < d.Cs * R; row_id += 4) {
      const int cs = row_id / R, r = row_id - cs * R;
      const bool row_ok = r < rows_out;
      float sc = 1.f, sh = 0.f;
      if (small_scale) {
        sc = small_scale[cs];
        sh = small_shift[cs];
      }
      const float* src = small_in + (((int64_t)b * d.Cs + cs) * d.Hs + (row_ok ? oh0 + r : 0)) * d.Ws;
      float* dst = small_tile + cs * SP + r * WsP;
      for (int cc = lane; cc < WsP; cc += 64) {
        float v = 0.f;
        if (row_ok && cc < d.Ws) v = fmaf(src[cc], sc, sh);
        dst[cc] = v;
      }
    }
    __syncthreads();
    const int S = rows_out * steps_per_row;
    for (int s = wk; s < S; s += WK) {
      const int r = s / steps_per_row, ow0 = (s - r * steps_per_row) * 4;
      const float* ap = small_tile + r * WsP + ow0;
      const float* bp = big_tile + 2 * r * Wt + 2 * ow0;
      float a[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = ap[offA[m]];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const float bv = bp[offB[n]];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv, acc[m][n], 0, 0, 0);
      }
    }
  }
  // ---- flush: D col = lane&15 = tap within the N tile, row = (lane>>4)*4 + reg = cs within the M tile
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int nt = gn * NB + n;
    if (nt < n_tiles) {
      const int cb = nt / NTAP_T;
      const int tau = (nt - cb * NTAP_T) * 16 + (lane & 15);
      if (tau < KK) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int cs = m * 16 + (lane >> 4) * 4 + reg;
            if (cs < d.Cs) atomicAdd(&gw[((int64_t)cs * d.Cb + cb) * KK + tau], acc[m][n][reg]);
          }
        }
      }
    }
  }
}

template <int KS, int MT, int NB, int WN>
int launch_wgrad(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                 const float* small_in, const float* small_scale, const float* small_shift, float* gw, hipStream_t st) {
  constexpr int KK = KS * KS;
  const int WsP = (d->Ws + 3) / 4 * 4;
  const int Wt = max(d->Wb + 2 * d->pad, 2 * (WsP - 1) + KS + 1);
  int R = min(d->Hs, 4);
  size_t bytes = 0;
  int SP = 0;
  for (; R >= 1; --R) {
    SP = R * WsP;
    SP += (34 - (SP % 32)) % 32;  // SP % 32 == 2: conflict-free A-fragment reads
    bytes = sizeof(float) * ((size_t)d->Cb * (2 * (R - 1) + KS) * Wt + (size_t)d->Cs * SP);
    if (bytes <= 72 * 1024 || R == 1) break;
  }
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = conv_wgrad_mfma_kernel<KS, MT, NB, WN>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
      pgv_set_error("conv_wgrad_mfma: cannot raise the dynamic LDS limit");
      return PGV_E_LAUNCH;
    }
    attr_done = true;
  }
  if (hipMemsetAsync(gw, 0, sizeof(float) * (size_t)d->Cs * d->Cb * KK, st) != hipSuccess) {
    pgv_set_error("conv_wgrad_mfma: memset failed");
    return PGV_E_LAUNCH;
  }
  const int bands = (int)pgv_cdiv(d->Hs, R);
  const int units = bands * d->B;
  if (units == 0) return 1;
  const int per_cu = (int)max((size_t)1, min((size_t)2, (size_t)kMaxLds / bytes));
  const int grid = min(units, 256 * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, st, *d, big, big_scale, big_shift, small_in, small_scale,
                     small_shift, gw, R, Wt, WsP, SP, bands, units);
  PGV_CHECK_LAUNCH("conv_wgrad_mfma");
  return 1;
}

}  // namespace

int pgv_conv_down_tuned(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                        const float* w, const float* bias, int act, float slope, float* small_out, double* stats,
                        hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != d->kw) return 0;
  if (d->kh == 4) {
    if (d->Cs <= 16)
      return launch_down<4, 1, 6, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, st);
    if (d->Cs <= 32)
      return launch_down<4, 2, 3, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, st);
    if (d->Cs <= 64)
      return launch_down<4, 4, 3, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, st);
    return 0;
  }
  if (d->kh == 5) {
    if (d->Cs <= 16)
      return launch_down<5, 1, 6, 1>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, st);
    return 0;
  }
  return 0;
}

int pgv_conv_up_tuned(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                      const float* w, const float* bias, int act, float slope, float* big_out, double* stats,
                      hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != d->kw) return 0;
  if (d->kh == 4) {
    if (d->Cb <= 8)
      return launch_up<4, 2, 6, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, st);
    if (d->Cb <= 16)
      return launch_up<4, 4, 3, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, st);
    if (d->Cb <= 32)
      return launch_up<4, 8, 3, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, st);
    return 0;
  }
  if (d->kh == 5) {
    if (d->Cb <= 4)
      return launch_up<5, 1, 6, 8>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, st);
    return 0;
  }
  return 0;
}

int64_t pgv_conv_wgrad_tuned_workspace(const pgv_conv_desc*) { return 0; }

int pgv_conv_wgrad_tuned(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                         const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                         void* /*workspace*/, int64_t /*workspace_bytes*/, hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != d->kw) return 0;
#define WG(KS, MT, NB, WN) \
  return launch_wgrad<KS, MT, NB, WN>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, st)
  if (d->kh == 4) {
    // N tiles = Cb; a wave group covers NB of them, WN groups cover NB*WN >= Cb
    if (d->Cs <= 16 && d->Cb <= 8) WG(4, 1, 8, 1);
    if (d->Cs <= 32 && d->Cb <= 16) WG(4, 2, 8, 2);
    if (d->Cs <= 64 && d->Cb <= 32) WG(4, 4, 8, 4);
    return 0;
  }
  if (d->kh == 5) {
    if (d->Cs <= 16 && d->Cb <= 2) WG(5, 1, 4, 1);  // 2 tap tiles per big channel
    return 0;
  }
#undef WG
  return 0;
}
