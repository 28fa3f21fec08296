// Direct (vector-ALU) kernels for the 1 <-> 8 channel 5x5 stride-2 layers at the spectrogram end of the network:
// enc1 = Conv2d(1,8,5,2,2) (model/encoder.py:241) and the output layer ConvTranspose2d(8,1,5,2,2)
// (model/decoder.py:218), i.e. the [B,1,257,347] <-> [B,8,129,174] pair, forward / input-gradient / weight-gradient.
//
// These layers move the largest tensors of the step (91 MB + 368 MB per op at B=256) for only 4.5 MMAC per sample:
// 50 FLOP per output byte — HBM-bound, and with 1 or 8 channels a 16x16 MFMA tile would be mostly padding.  So:
// one lane per output pixel (DOWN, WGRAD) or per 2x2 output block (UP), lanes on consecutive pixels of a row so every
// global access is a contiguous 256 B per wave; the band's input rows are staged once in LDS (conv_tile.h, 16-byte
// copies, producer's BatchNorm affine folded in); weights are wave-uniform and come through the scalar cache.
#include "conv_tile.h"

namespace {

constexpr int KS = 5, KK = 25, PAD = 2;

// ---- DOWN: out[b,cs,oh,ow] = act(bias[cs] + sum_{kh,kw} x[b,0,2oh-2+kh,2ow-2+kw] * w[cs,0,kh,kw]) ------------------
// BF16 (PGV_COMPUTE_BF16): the 25 taps as 13 bf16 pairs through v_dot2c_f32_bf16 (both operands rounded to bfloat16,
// fp32 accumulation): the weights are rounded and packed once per workgroup (LDS), then live in registers.
constexpr int KP = (KK + 1) / 2;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  const bf16x2_t v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

template <int CS, bool BF16>
__global__ __launch_bounds__(256) void down_c1_kernel(pgv_conv_desc d, const float* __restrict__ big,
                                                      const float* __restrict__ in_scale,
                                                      const float* __restrict__ in_shift,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      int act, float slope, float* __restrict__ out, int R, int plane) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* aff = lds;  // [2] (+2 pad)
  const int tid = threadIdx.x;
  const int b = blockIdx.y, oh0 = blockIdx.x * R;
  const int rows_out = min(R, d.Hs - oh0), Pb = rows_out * d.Ws;
  const int Wb = d.Wb, rows_in = 2 * (R - 1) + KS, ih0 = oh0 * 2 - PAD;
  const int lead = (max(ih0, 0) - ih0) * Wb;
  float* tile = lds + 8 + ((4 - (lead & 3)) & 3);
  stage_affine(aff, in_scale, in_shift, 1, tid);
  if (in_scale) __syncthreads();
  stage_rows_contig<4>(tile, plane, big + (int64_t)b * d.Hb * Wb, 1, d.Hb, Wb, 0, 1, rows_in, ih0,
                       in_scale ? aff : nullptr, aff + 1, tid);
  unsigned* wl = reinterpret_cast<unsigned*>(lds + 16 + plane);  // [CS][KP] packed bf16 weight pairs (BF16 only)
  if constexpr (BF16) {
    for (int i = tid; i < CS * KP; i += 256) {
      const int cs = i / KP, k = 2 * (i - cs * KP);
      wl[i] = pack_bf16x2(cs < d.Cs ? w[cs * KK + k] : 0.f, (cs < d.Cs && k + 1 < KK) ? w[cs * KK + k + 1] : 0.f);
    }
  }
  __syncthreads();
  unsigned wr[BF16 ? CS * KP : 1];
  if constexpr (BF16) {
#pragma unroll
    for (int i = 0; i < CS * KP; ++i) wr[i] = wl[i];
  }
  const float inv_ws = 1.0f / (float)d.Ws;
  const int64_t cstride = (int64_t)d.Hs * d.Ws;
  float* ob = out + (int64_t)b * d.Cs * cstride + (int64_t)oh0 * d.Ws;
  for (int p = tid; p < Pb; p += 256) {
    const int r = fast_div(p, inv_ws), c = p - r * d.Ws;
    const float* tp = tile + 2 * r * Wb + 2 * c - PAD;
    float x[KK];
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) {
      const bool ok = (unsigned)(2 * c - PAD + kw) < (unsigned)Wb;
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        const float raw = tp[kh * Wb + kw];
        x[kh * KS + kw] = ok ? raw : 0.f;
      }
    }
    if constexpr (BF16) {
      unsigned xp[KP];
#pragma unroll
      for (int j = 0; j < KP; ++j) xp[j] = pack_bf16x2(x[2 * j], 2 * j + 1 < KK ? x[2 * j + 1] : 0.f);
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) {
        if (cs < d.Cs) {
          float a = bias ? bias[cs] : 0.f;
#pragma unroll
          for (int j = 0; j < KP; ++j) a = dot2_bf16(xp[j], wr[cs * KP + j], a);
          ob[cs * cstride + p] = pgv_act(a, act, slope);
        }
      }
    } else {
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) {
        if (cs < d.Cs) {
          float a = bias ? bias[cs] : 0.f;
#pragma unroll
          for (int k = 0; k < KK; ++k) a = fmaf(x[k], w[cs * KK + k], a);
          ob[cs * cstride + p] = pgv_act(a, act, slope);
        }
      }
    }
  }
}

// ---- UP: out[b,0,2u+ph,2v+pw] = act(bias + sum_{cs,th,tw} x'[b,cs,u+1-th,v+1-tw] * w[cs,0,ph+2th,pw+2tw]) ----------
// BF16: channel pairs (2c, 2c+1) per v_dot2c_f32_bf16 - 4 x 25 packed weight pairs in registers.
template <int CS, bool BF16>
__global__ __launch_bounds__(256) void up_c1_kernel(pgv_conv_desc d, const float* __restrict__ small_in,
                                                    const float* __restrict__ in_scale,
                                                    const float* __restrict__ in_shift, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int act, float slope,
                                                    float* __restrict__ out, int R, int plane, int Hg, int Wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* aff = lds;  // [2*CS]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, u0 = blockIdx.x * R;
  const int rows_g = min(R, Hg - u0), Pb = rows_g * Wg;
  const int Ws = d.Ws, rows_in = R + 2, ih0 = u0 - 1;
  const int lead = (max(ih0, 0) - ih0) * Ws;
  float* tile = lds + 2 * CS + 8 + ((4 - (lead & 3)) & 3);
  stage_affine(aff, in_scale, in_shift, d.Cs, tid);
  if (in_scale) __syncthreads();
  stage_rows_contig<4>(tile, plane, small_in + (int64_t)b * d.Cs * d.Hs * Ws, d.Cs, d.Hs, Ws, 0, CS, rows_in, ih0,
                       in_scale ? aff : nullptr, aff + d.Cs, tid);
  static_assert(CS % 2 == 0, "channel pairs");
  constexpr int CP = CS / 2;
  unsigned* wl = reinterpret_cast<unsigned*>(lds + 2 * CS + 16 + CS * plane);  // [CP][KK] packed pairs (BF16 only)
  if constexpr (BF16) {
    for (int i = tid; i < CP * KK; i += 256) {
      const int c = i / KK, k = i - c * KK;
      wl[i] = pack_bf16x2(2 * c < d.Cs ? w[(2 * c) * KK + k] : 0.f, 2 * c + 1 < d.Cs ? w[(2 * c + 1) * KK + k] : 0.f);
    }
  }
  __syncthreads();
  unsigned wr[BF16 ? CP * KK : 1];
  if constexpr (BF16) {
#pragma unroll
    for (int i = 0; i < CP * KK; ++i) wr[i] = wl[i];
  }
  const float inv_wg = 1.0f / (float)Wg;
  const float bv = bias ? bias[0] : 0.f;
  float* ob = out + (int64_t)b * d.Hb * d.Wb;
  for (int p = tid; p < Pb; p += 256) {
    const int ur = fast_div(p, inv_wg), v = p - ur * Wg;
    // local rows ur+2-th (th = 0..2) <-> input rows u+1-th ; cols v+1-tw
    const float* tp = tile + (ur + 2) * Ws + v + 1;
    bool okc[3];
#pragma unroll
    for (int tw = 0; tw < 3; ++tw) okc[tw] = (unsigned)(v + 1 - tw) < (unsigned)Ws;
    float a00 = bv, a01 = bv, a10 = bv, a11 = bv;
    if constexpr (BF16) {
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        if (2 * c < d.Cs) {
          const bool hi = 2 * c + 1 < d.Cs;
          unsigned x[3][3];
#pragma unroll
          for (int th = 0; th < 3; ++th)
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
              const float r0 = tp[(2 * c) * plane - th * Ws - tw], r1 = tp[(2 * c + 1) * plane - th * Ws - tw];
              x[th][tw] = pack_bf16x2(okc[tw] ? r0 : 0.f, (okc[tw] && hi) ? r1 : 0.f);
            }
          const unsigned* wc = wr + c * KK;  // [kh][kw], kh = ph + 2 th, kw = pw + 2 tw
#pragma unroll
          for (int th = 0; th < 3; ++th)
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
              a00 = dot2_bf16(x[th][tw], wc[(2 * th) * KS + 2 * tw], a00);
              if (tw < 2) a01 = dot2_bf16(x[th][tw], wc[(2 * th) * KS + 2 * tw + 1], a01);
              if (th < 2) a10 = dot2_bf16(x[th][tw], wc[(2 * th + 1) * KS + 2 * tw], a10);
              if (th < 2 && tw < 2) a11 = dot2_bf16(x[th][tw], wc[(2 * th + 1) * KS + 2 * tw + 1], a11);
            }
        }
      }
    } else {
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) {
        if (cs < d.Cs) {
          float x[3][3];
#pragma unroll
          for (int th = 0; th < 3; ++th)
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
              const float raw = tp[cs * plane - th * Ws - tw];
              x[th][tw] = okc[tw] ? raw : 0.f;
            }
          const float* wc = w + cs * KK;  // [kh][kw], kh = ph + 2 th, kw = pw + 2 tw
#pragma unroll
          for (int th = 0; th < 3; ++th)
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
              a00 = fmaf(x[th][tw], wc[(2 * th) * KS + 2 * tw], a00);
              if (tw < 2) a01 = fmaf(x[th][tw], wc[(2 * th) * KS + 2 * tw + 1], a01);
              if (th < 2) a10 = fmaf(x[th][tw], wc[(2 * th + 1) * KS + 2 * tw], a10);
              if (th < 2 && tw < 2) a11 = fmaf(x[th][tw], wc[(2 * th + 1) * KS + 2 * tw + 1], a11);
            }
        }
      }
    }
    const int oh = 2 * (u0 + ur), ow = 2 * v;
    float* o = ob + (int64_t)oh * d.Wb + ow;
    const bool c1 = ow + 1 < d.Wb;
    o[0] = pgv_act(a00, act, slope);
    if (c1) o[1] = pgv_act(a01, act, slope);
    if (oh + 1 < d.Hb) {
      o[d.Wb] = pgv_act(a10, act, slope);
      if (c1) o[d.Wb + 1] = pgv_act(a11, act, slope);
    }
  }
}

// ---- WGRAD: gw[cs,0,kh,kw] = sum_{b,oh,ow} small'[b,cs,oh,ow] * big'[b,0,2oh-2+kh,2ow-2+kw] --------------------------
// Persistent workgroups over (sample, band) units; each lane keeps CG*25 partial sums for CG channels per pass.
template <int CS, int CG>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(pgv_conv_desc d, const float* __restrict__ big,
                                                       const float* __restrict__ big_scale,
                                                       const float* __restrict__ big_shift,
                                                       const float* __restrict__ small_in,
                                                       const float* __restrict__ small_scale,
                                                       const float* __restrict__ small_shift, float* __restrict__ gw,
                                                       int R, int plane, int SP, int bands, int units) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* aff_b = lds;             // [2]
  float* aff_s = lds + 4;         // [2*CS]
  float* red = lds + 4 + 2 * CS;  // [4 waves][CG*KK]
  float* small_tile = red + 4 * CG * KK + ((4 - ((4 * CG * KK) & 3)) & 3);  // [CS][SP]
  float* big_base = small_tile + CS * SP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Wb = d.Wb, Ws = d.Ws, rows_in = 2 * (R - 1) + KS;
  stage_affine(aff_b, big_scale, big_shift, 1, tid);
  stage_affine(aff_s, small_scale, small_shift, d.Cs, tid);
  __syncthreads();
  const float inv_ws = 1.0f / (float)Ws;
  // waves {0,1} accumulate channels [0,CG), waves {2,3} channels [CG,2CG): CG*25 partial sums per lane, every pair of
  // waves sweeps all pixels of the band
  static_assert(CS == 2 * CG, "two wave pairs");
  const int g = wave >> 1, t128 = tid & 127;
  float acc[CG * KK];
#pragma unroll
  for (int i = 0; i < CG * KK; ++i) acc[i] = 0.f;

  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int b = u / bands, oh0 = (u - b * bands) * R;
    const int rows_out = min(R, d.Hs - oh0), Pb = rows_out * Ws;
    const int ih0 = oh0 * 2 - PAD;
    const int lead = (max(ih0, 0) - ih0) * Wb;
    float* tile = big_base + 8 + ((4 - (lead & 3)) & 3);
    if (u != (int)blockIdx.x) __syncthreads();
    stage_rows_contig<4>(tile, plane, big + (int64_t)b * d.Hb * Wb, 1, d.Hb, Wb, 0, 1, rows_in, ih0,
                         big_scale ? aff_b : nullptr, aff_b + 1, tid);
    stage_rows_contig<4>(small_tile, SP, small_in + (int64_t)b * d.Cs * d.Hs * Ws, d.Cs, d.Hs, Ws, 0, CS, R, oh0,
                         small_scale ? aff_s : nullptr, aff_s + d.Cs, tid);
    __syncthreads();
    for (int p = t128; p < Pb; p += 128) {
      const int r = fast_div(p, inv_ws), c = p - r * Ws;
      const float* tp = tile + 2 * r * Wb + 2 * c - PAD;
      float x[KK];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const bool ok = (unsigned)(2 * c - PAD + kw) < (unsigned)Wb;
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
          const float raw = tp[kh * Wb + kw];
          x[kh * KS + kw] = ok ? pgv_opnd(raw, (d.flags & PGV_COMPUTE_BF16) != 0) : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        const float gsm = pgv_opnd(small_tile[(g * CG + j) * SP + p], (d.flags & PGV_COMPUTE_BF16) != 0);
#pragma unroll
        for (int k = 0; k < KK; ++k) acc[j * KK + k] = fmaf(gsm, x[k], acc[j * KK + k]);
      }
    }
  }
  // ---- reduce the per-lane partial sums: wave shuffle, then the two waves of each channel group through LDS
  __syncthreads();
#pragma unroll
  for (int i = 0; i < CG * KK; ++i) {
    const float sv = pgv_wave_sum(acc[i]);
    if (lane == 0) red[wave * CG * KK + i] = sv;
  }
  __syncthreads();
  for (int i = tid; i < CS * KK; i += 256) {
    const int cs = i / KK, gg = cs / CG, ii = i - gg * CG * KK;
    if (cs < d.Cs) atomicAdd(&gw[i], red[(2 * gg) * CG * KK + ii] + red[(2 * gg + 1) * CG * KK + ii]);
  }
}

bool shape_ok(const pgv_conv_desc* d) {
  return d->kh == 5 && d->kw == 5 && d->stride == 2 && d->pad == 2 && d->Cb == 1 && d->Cs <= 8 && d->B > 0;
}

}  // namespace

int pgv_conv_down_direct(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                         const float* w, const float* bias, int act, float slope, float* out, double* stats,
                         hipStream_t st) {
  if (!shape_ok(d) || stats) return 0;
  int R = min(d->Hs, 8);
  size_t bytes = 0;
  int plane = 0;
  for (; R >= 1; --R) {
    plane = ((2 * (R - 1) + KS) * d->Wb + 16 + 3) / 4 * 4;
    bytes = sizeof(float) * (16 + (size_t)plane + 8 * KP);
    if (bytes <= 40 * 1024 || R == 1) break;
  }
  if (bytes > (size_t)kMaxLds) return 0;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  auto kern = bf16 ? down_c1_kernel<8, true> : down_c1_kernel<8, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[bf16], "conv_down_direct");
  if (rc) return rc;
  dim3 grid((unsigned)pgv_cdiv(d->Hs, R), (unsigned)d->B);
  hipLaunchKernelGGL(kern, grid, dim3(256), bytes, st, *d, big, in_scale, in_shift, w, bias, act, slope, out, R, plane);
  PGV_CHECK_LAUNCH("conv_down_direct");
  return 1;
}

int pgv_conv_up_direct(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                       const float* w, const float* bias, int act, float slope, float* out, double* stats,
                       hipStream_t st) {
  if (!shape_ok(d) || stats) return 0;
  const int Hg = (d->Hb + 1) / 2, Wg = (d->Wb + 1) / 2;
  int R = min(Hg, 8);
  size_t bytes = 0;
  int plane = 0;
  for (; R >= 1; --R) {
    plane = ((R + 2) * d->Ws + 16 + 3) / 4 * 4;
    bytes = sizeof(float) * (16 + 16 + (size_t)8 * plane + 4 * KK);
    if (bytes <= 40 * 1024 || R == 1) break;
  }
  if (bytes > (size_t)kMaxLds) return 0;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  auto kern = bf16 ? up_c1_kernel<8, true> : up_c1_kernel<8, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[bf16], "conv_up_direct");
  if (rc) return rc;
  dim3 grid((unsigned)pgv_cdiv(Hg, R), (unsigned)d->B);
  hipLaunchKernelGGL(kern, grid, dim3(256), bytes, st, *d, small_in, in_scale, in_shift, w, bias, act, slope, out, R,
                     plane, Hg, Wg);
  PGV_CHECK_LAUNCH("conv_up_direct");
  return 1;
}

int pgv_conv_wgrad_direct(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                          const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                          hipStream_t st) {
  if (!shape_ok(d)) return 0;
  constexpr int CG = 4;
  int R = min(d->Hs, 4);
  size_t bytes = 0;
  int plane = 0, SP = 0;
  for (; R >= 1; --R) {
    plane = ((2 * (R - 1) + KS) * d->Wb + 16 + 3) / 4 * 4;
    SP = (R * d->Ws + 8 + 3) / 4 * 4;
    bytes = sizeof(float) * (4 + 16 + 4 * CG * KK + 4 + (size_t)8 * SP + 16 + (size_t)plane);
    if (bytes <= 48 * 1024 || R == 1) break;
  }
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = wgrad_c1_kernel<8, CG>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_wgrad_direct");
  if (rc) return rc;
  if (!(d->flags & PGV_PREZEROED) && hipMemsetAsync(gw, 0, sizeof(float) * (size_t)d->Cs * KK, st) != hipSuccess) {
    pgv_set_error("conv_wgrad_direct: memset failed");
    return PGV_E_LAUNCH;
  }
  const int bands = (int)pgv_cdiv(d->Hs, R), units = bands * d->B;
  const int grid = min(units, 256 * 3);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, st, *d, big, big_scale, big_shift, small_in, small_scale,
                     small_shift, gw, R, plane, SP, bands, units);
  PGV_CHECK_LAUNCH("conv_wgrad_direct");
  return 1;
}
