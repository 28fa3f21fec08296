// LDS tile staging helpers shared by the band convolution kernels (conv_mfma.hip, conv_pipe.hip).
#pragma once
#include "conv_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// bf16 operands (PGV_COMPUTE_BF16) as the kernels build them: four consecutive k values per lane and 16-deep step, packed from
// fp32 with round-to-nearest-even (v_cvt_pk_bf16_f32); products of bf16 values are exact in fp32, accumulation is fp32
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x4 pack_bf16x4(float a, float b, float c, float d) {
  const bf16x2_t lo = {(__bf16)a, (__bf16)b}, hi = {(__bf16)c, (__bf16)d};
  const unsigned u[2] = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
  return __builtin_bit_cast(s16x4, u);
}
// gfx950's K = 32 form: one v_mfma_f32_16x16x32_bf16 consumes the operands of TWO consecutive 16-deep steps - a lane's
// eight values are its four of step s followed by its four of step s + 1 (the contraction only needs A and B to agree on
// the position of every k, and both are built this way), in the 16 cycles the legacy 16x16x16 form spends on one step.
// An odd step count pads the last pair with zeros.
// The eight values live in ONE 4-register operand (u32x4) whose halves are written in place (set_half): built from two
// separate s16x4 values the register allocator has to copy them next to each other (spills in the large-tile kernels).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int HALF>
__device__ __forceinline__ void set_half(u32x4& dst, s16x4 v) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 u = __builtin_bit_cast(u32x2, v);
  dst[2 * HALF] = u[0];
  dst[2 * HALF + 1] = u[1];
}
__device__ __forceinline__ f32x4 mfma_bf16_k32(u32x4 a, u32x4 b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16_k32(s16x4 a_lo, s16x4 a_hi, s16x4 b_lo, s16x4 b_hi, f32x4 acc) {
  u32x4 a, b;
  set_half<0>(a, a_lo), set_half<1>(a, a_hi), set_half<0>(b, b_lo), set_half<1>(b, b_hi);
  return mfma_bf16_k32(a, b, acc);
}
__device__ __forceinline__ s16x4 zero_bf16x4() { return s16x4{0, 0, 0, 0}; }
// fp32 value rounded to bf16 precision (kernels without a bf16 MFMA loop emulate the operand precision this way)
__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }

constexpr int kMaxLds = 160 * 1024;
constexpr int kLdsTarget = 76 * 1024;  // aim at two workgroups per CU

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float group16_sum(float v) {
  // sum over the 16 lanes that share lane>>4, result in all 16: DPP only (quad xor 1, quad xor 2, half-row mirror,
  // row mirror) - __shfl_xor would go through ds_bpermute, ~100 cycles of LDS latency per step
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

// Exact n / d for 0 <= n < 2^20, 1 <= d < 2^12 (tile sizes here), via one float multiply.
// Persistent kernels walk the units  bid, bid + grid, bid + 2 grid, ...  Blocks b and b + 8 share an XCD and its L2
// (round-robin placement, MI355X_MICROARCH.md; a speed matter only), so the blocks of one XCD are given CONSECUTIVE
// units - neighbouring bands of a sample, processed at about the same time: the halo rows that two neighbouring bands
// both stage then come out of that L2 instead of being fetched from HBM once per band.
__device__ __forceinline__ int pgv_xcd_block() {
  const int g = (int)gridDim.x, b = (int)blockIdx.x;
  return (g & 7) ? b : (b & 7) * (g >> 3) + (b >> 3);
}

__device__ __forceinline__ int fast_div(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }

// Copy a [nch][rows][Wt] window of one sample's [C][H][W] tensor into LDS: element (c, rr, cc) <-> global
// (c0+c, ih0+rr, iw0+cc), zero outside the image / beyond C, optional per-channel affine (aff_sc/aff_sh indexed by the
// global channel) applied inside the image only — i.e. the zero padding stays zero, as nn.Conv2d pads the *BatchNorm
// output*.  U independent loads are issued per lane before the first one is consumed.
template <int U>
__device__ __forceinline__ void stage_window(float* __restrict__ tile, int plane_stride,
                                             const float* __restrict__ src, int C, int H, int W, int c0, int nch,
                                             int rows, int Wt, int ih0, int iw0, const float* __restrict__ aff_sc,
                                             const float* __restrict__ aff_sh, int tid) {
  const int total = nch * rows * Wt;
  const float inv_wt = 1.0f / (float)Wt, inv_rows = 1.0f / (float)rows;
  for (int e0 = tid; e0 < total; e0 += 256 * U) {
    float v[U];
    int li[U], ch[U];
    bool inb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = min(e0 + u * 256, total - 1);
      const int row = fast_div(e, inv_wt), cc = e - row * Wt;
      const int c = fast_div(row, inv_rows), rr = row - c * rows;
      const int cg = c0 + c, ih = ih0 + rr, iw = iw0 + cc;
      inb[u] = cg < C && ih >= 0 && ih < H && iw >= 0 && iw < W;
      v[u] = inb[u] ? src[((int64_t)cg * H + ih) * W + iw] : 0.f;
      li[u] = c * plane_stride + rr * Wt + cc;
      ch[u] = cg;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e0 + u * 256 < total) {
        float x = v[u];
        if (aff_sc && inb[u]) x = fmaf(x, aff_sc[ch[u]], aff_sh[ch[u]]);
        tile[li[u]] = x;
      }
    }
  }
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load from a 4-byte aligned address

// Row-band copy for tiles whose LDS row stride equals the image width W: rows [ih_lo, ih_hi) of a channel are ONE
// contiguous NCHW segment, copied 16 bytes per lane per load with U loads in flight (global_load_dwordx4 ->
// ds_write_b128) to  tile + c*plane_stride + (ih_lo-ih0)*W ; window rows outside the image are zero-filled, channels
// beyond C are zero planes; the producer's BatchNorm affine is applied to the copied data (the horizontal zero padding
// is not stored at all: consumers mask out-of-range columns).  Caller guarantees 16-byte alignment of
// tile + (ih_lo-ih0)*W and plane_stride % 4 == 0.
template <int U>
__device__ __forceinline__ void stage_rows_contig(float* __restrict__ tile, int plane_stride,
                                                  const float* __restrict__ src, int C, int H, int W, int c0, int nch,
                                                  int rows, int ih0, const float* __restrict__ aff_sc,
                                                  const float* __restrict__ aff_sh, int tid) {
  const int ih_lo = max(ih0, 0), ih_hi = min(ih0 + rows, H);
  const int L = max(ih_hi - ih_lo, 0) * W;  // floats per channel
  const int Q = (L + 3) >> 2;
  const int lead = (ih_lo - ih0) * W;
  const float inv_q = 1.0f / (float)max(Q, 1);
  const int items = nch * Q;
  for (int e0 = tid; e0 < items; e0 += 256 * U) {
    f32x4 v[U];
    int cc[U], qq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = min(e0 + u * 256, items - 1);
      const int c = fast_div(e, inv_q), q = e - c * Q;
      const int cg = c0 + c;
      cc[u] = c;
      qq[u] = q;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (cg < C) {
        const float* g = src + ((int64_t)cg * H + ih_lo) * W + 4 * q;
        if (4 * q + 4 <= L) {
          const f4u t = *reinterpret_cast<const f4u*>(g);
          v[u] = f32x4{t.x, t.y, t.z, t.w};
        } else {
          v[u].x = g[0];
          if (4 * q + 1 < L) v[u].y = g[1];
          if (4 * q + 2 < L) v[u].z = g[2];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e0 + u * 256 < items) {
        f32x4 x = v[u];
        const int cg = c0 + cc[u];
        if (aff_sc && cg < C) {
          const float sc = aff_sc[cg], sh = aff_sh[cg];
          const int rem = L - 4 * qq[u];  // elements of this quad inside the segment
          x.x = fmaf(x.x, sc, sh);
          x.y = rem > 1 ? fmaf(x.y, sc, sh) : 0.f;
          x.z = rem > 2 ? fmaf(x.z, sc, sh) : 0.f;
          x.w = rem > 3 ? fmaf(x.w, sc, sh) : 0.f;
        }
        *reinterpret_cast<f32x4*>(tile + cc[u] * plane_stride + lead + 4 * qq[u]) = x;
      }
    }
  }
  // zero rows above / below the image (first and last bands only)
  const int tail0 = lead + 4 * Q, tail_n = rows * W - tail0;
  if (lead > 0 || tail_n > 0) {
    for (int c = 0; c < nch; ++c) {
      float* p = tile + c * plane_stride;
      for (int i = tid; i < lead; i += 256) p[i] = 0.f;
      for (int i = tid; i < tail_n; i += 256) p[tail0 + i] = 0.f;
    }
  }
}

// Copy an LDS tile [nch][row_stride] (first len floats of each row) to nch contiguous global segments dst + c*cstride,
// 16 bytes per lane per store (global_store_dwordx4 to 4-byte aligned addresses): dword stores from the MFMA
// accumulator layout are store-issue-bound (~6x the time per byte of 16-byte stores on gfx950).
// row_stride % 4 == 0 and tile 16-byte aligned.
__device__ __forceinline__ void store_rows_contig(const float* __restrict__ tile, int row_stride,
                                                  float* __restrict__ dst, int64_t cstride, int nch, int len,
                                                  int tid) {
  const int Q = len >> 2, rem = len & 3;
  const float inv_q = 1.0f / (float)max(Q, 1);
  const int items = nch * Q;
  for (int e = tid; e < items; e += 256) {
    const int c = fast_div(e, inv_q), q = e - c * Q;
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + c * row_stride + 4 * q);
    f4u o;
    o.x = v.x, o.y = v.y, o.z = v.z, o.w = v.w;
    *reinterpret_cast<f4u*>(dst + c * cstride + 4 * q) = o;
  }
  if (rem) {
    for (int e = tid; e < nch * 4; e += 256) {
      const int c = e >> 2, i = e & 3;
      if (i < rem) dst[c * cstride + 4 * Q + i] = tile[c * row_stride + 4 * Q + i];
    }
  }
}

// Derivative of the block activation recovered from the saved ACTIVATED tensor (pgv_bwd_fuse): LeakyReLU keeps the sign
// (slope >= 0), Hardtanh passes no gradient at / beyond the bounds (torch semantics).
struct pgv_actd_params {
  float ns, lo, hi;
};
__device__ __forceinline__ pgv_actd_params pgv_actd_setup(int act, float slope) {
  pgv_actd_params p;
  p.ns = act == PGV_ACT_LEAKY_RELU ? slope : 1.0f;
  p.lo = act == PGV_ACT_HARDTANH ? -1.0f : -__builtin_inff();
  p.hi = act == PGV_ACT_HARDTANH ? 1.0f : __builtin_inff();
  return p;
}
// g_y = act'(a) * (ka*g + kb*a + kc)
__device__ __forceinline__ float pgv_bwd_apply(float g, float a, float ka, float kb, float kc, const pgv_actd_params& p) {
  const float t = fmaf(g, ka, fmaf(a, kb, kc));
  const float d = a > 0.f ? t : p.ns * t;
  return (a > p.lo && a < p.hi) ? d : 0.f;
}

// store_rows_contig with the BatchNorm + activation backward of the next-lower block fused in (pgv_bwd_fuse): TPC =
// 256/NCHP lanes own one channel row each, read the saved activation `a` at the offsets they store to (all loads issued
// before the copy loop), store g_y = act'(a) * (ka*g + kb*a + kc) instead of g and add the row's sum of g_y (the bias
// gradient) into acc[c] (LDS, one owner lane per channel: accumulated over all units of a persistent workgroup,
// flushed once with float atomics).
// The loads of the saved activation, split off so that the caller can issue them BEFORE the band is transposed through
// LDS (bnred_fetch ... LDS writes ... barrier ... store_rows_bwd): issued inside the store pass every unit exposed one
// memory latency.
template <int NCHP, int MAXIT>
__device__ __forceinline__ void bnred_fetch(const float* __restrict__ a, int64_t cstride, int nch, int len, int tid,
                                            f4u (&av)[MAXIT]) {
  constexpr int TPC = 256 / NCHP;
  const int c = tid / TPC, j = tid - c * TPC;
  const int Q = len >> 2;
  const bool cok = c < nch;
  const float* ap = a + c * cstride;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i) {
    const int q = j + i * TPC;
    av[i] = (cok && q < Q) ? *reinterpret_cast<const f4u*>(ap + 4 * q) : f4u{0.f, 0.f, 0.f, 0.f};
  }
}

// coef: [3][Ctot] table of the output tensor's channels, already offset to the first channel of this group.
// CLSW != 0: rows of the band are CLSW (a multiple of 4) floats wide and start at image row `row0`; the sums are kept by
// (row parity, column parity) class in acc[4*c + 2*rp + cp] (pgv_bwd_fuse.cls), else one sum per channel in acc[c].
template <int NCHP, int MAXIT, int CLSW = 0>
__device__ __forceinline__ void store_rows_bwd(const float* __restrict__ tile, int row_stride, float* __restrict__ dst,
                                               const float* __restrict__ a, int64_t cstride, int nch, int len,
                                               int tid, const float* __restrict__ coef, int Ctot,
                                               const pgv_actd_params& actd, float* __restrict__ acc,
                                               const f4u (&av)[MAXIT], int row0 = 0) {
  constexpr int TPC = 256 / NCHP;
  static_assert(TPC == 8 || TPC == 16 || TPC == 32, "lanes per channel");
  static_assert(CLSW % 4 == 0, "class sums need rows of whole 16-byte pieces");
  const int c = tid / TPC, j = tid - c * TPC;
  const int Q = len >> 2, rem = len & 3;
  const bool cok = c < nch;
  const float ka = cok ? coef[c] : 0.f, kb = cok ? coef[Ctot + c] : 0.f, kc = cok ? coef[2 * Ctot + c] : 0.f;
  const float* ap = a + c * cstride;
  float* dp = dst + c * cstride;
  const float* tp = tile + c * row_stride;
  constexpr int NS = CLSW ? 4 : 1;
  float s1[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) s1[k] = 0.f;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i) {
    const int q = j + i * TPC;
    if (cok && q < Q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tp + 4 * q);
      f4u o;
      o.x = pgv_bwd_apply(v.x, av[i].x, ka, kb, kc, actd);
      o.y = pgv_bwd_apply(v.y, av[i].y, ka, kb, kc, actd);
      o.z = pgv_bwd_apply(v.z, av[i].z, ka, kb, kc, actd);
      o.w = pgv_bwd_apply(v.w, av[i].w, ka, kb, kc, actd);
      *reinterpret_cast<f4u*>(dp + 4 * q) = o;
      if (CLSW) {
        const bool rodd = (row0 + (4 * q) / (CLSW ? CLSW : 1)) & 1;
        const float ev = o.x + o.z, od = o.y + o.w;   // the piece starts at an even column
        s1[0] += rodd ? 0.f : ev, s1[1 % NS] += rodd ? 0.f : od, s1[2 % NS] += rodd ? ev : 0.f, s1[3 % NS] += rodd ? od : 0.f;
      } else {
        s1[0] += (o.x + o.y) + (o.z + o.w);
      }
    }
  }
  if (rem && cok && j == 0) {   // (CLSW: len is a multiple of 4, no remainder)
    for (int i = 0; i < rem; ++i) {
      const float o = pgv_bwd_apply(tp[4 * Q + i], ap[4 * Q + i], ka, kb, kc, actd);
      dp[4 * Q + i] = o;
      s1[0] += o;
    }
  }
  // sum over the TPC consecutive lanes of the channel
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float t = s1[k];
    t += dpp_mov<0xB1>(t);
    t += dpp_mov<0x4E>(t);
    t += dpp_mov<0x141>(t);  // 8 lanes
    if (TPC >= 16) t += dpp_mov<0x140>(t);
    if (TPC >= 32) t += __shfl_xor(t, 16, 64);
    if (cok && j == 0) acc[NS * c + k] += t;
  }
}

__device__ __forceinline__ void stage_affine(float* __restrict__ aff, const float* __restrict__ scale,
                                             const float* __restrict__ shift, int C, int tid) {
  if (scale)
    for (int i = tid; i < C; i += 256) {
      aff[i] = scale[i];
      aff[C + i] = shift[i];
    }
}

template <typename K>
int raise_lds_limit(K kern, bool* done, const char* who) {
  if (!*done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) {
      pgv_set_error("%s: cannot raise the dynamic LDS limit: %s", who, hipGetErrorString(e));
      return PGV_E_LAUNCH;
    }
    *done = true;
  }
  return PGV_OK;
}


}  // namespace
