// Register prefetch of LDS band tiles (global -> registers -> LDS with the producer's BatchNorm affine), shared by the
// band kernels (conv_band.hip) and the second-generation kernels (conv_v2.hip).
#pragma once
#include <type_traits>
#include "conv_tile.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Register prefetch of a band tile: CK channels x ROWS rows, LDS row stride WP (image columns 0..W-1, then >= 2 zero
// pad columns which double as the left padding of the next row), channel stride ROWS*WP.  A channel is PC = ROWS*WP/4
// 16-byte chunks; lane tid owns chunks i = tid + 256*k (k < SPC) of EVERY channel, so the chunk decode (row, column,
// partial-chunk flag) is done once per kernel for SPC slots, the channel of a register is a compile-time constant,
// global addresses are (uniform channel base) + (per-slot offset) and LDS addresses are (per-lane base) + immediate.
//   issue():  one 16-byte load per chunk, no branches, no waits; chunks that hold no image data (rows outside the
//             image, pad chunks, channels beyond C) load offset 0 of the sample and are zeroed at commit.  The partial
//             chunk at the end of a row (W % 4 != 0) reads the LAST four floats of the row and is rotated into place
//             at commit, so nothing outside the tensor is ever read.
//   commit(): producer's BatchNorm affine (uniform per channel: scalar loads) on image data only, then ds_write_b128.
// Between the two the workgroup runs the MFMA loop and the epilogue of the previous tile: HBM latency is hidden.
// ---------------------------------------------------------------------------------------------------------------
template <int CK, int ROWS, int W, int WP, int H, int MINPAD = 2, int CSTRIDE = ROWS * WP>
struct BandPrefetch {
  static constexpr int QR = WP / 4;
  static constexpr int PC = ROWS * QR;            // chunks per channel
  static_assert(CSTRIDE >= ROWS * WP && CSTRIDE % 4 == 0, "channel stride");
  static constexpr int SPC = (PC + 255) / 256;    // slots per channel and lane
  static constexpr int NPF = CK * SPC;
  static constexpr int NP = W % 4;
  static_assert(WP % 4 == 0 && WP >= W + MINPAD, "row stride");
  f32x4 v[NPF];
  int rr[SPC];         // tile row of slot k
  int col[SPC];        // first image column loaded by slot k (shifted back for the partial chunk)
  int ncol[SPC];       // image floats in the chunk: 4, W%4 (partial) or 0 (pad chunk / idle lane)
  unsigned offb[SPC];  // byte offset inside the channel plane for the current item (0 when not image data)
  unsigned live;       // bit k: slot k holds image data for the current item

  __device__ __forceinline__ void init(int tid) {
#pragma unroll
    for (int k = 0; k < SPC; ++k) {
      const int i = tid + 256 * k;
      const int r = i / QR, q = i - r * QR;
      const int nc = i < PC ? min(max(W - 4 * q, 0), 4) : 0;
      rr[k] = r;
      ncol[k] = nc;
      col[k] = 4 * q - ((NP != 0 && nc > 0 && nc < 4) ? 4 - NP : 0);
    }
  }
  // plane0 = first element of the first channel of this chunk in the current sample (always a valid address)
  __device__ __forceinline__ void issue(const float* __restrict__ plane0, int ih0, int nch) {
    live = 0;
#pragma unroll
    for (int k = 0; k < SPC; ++k) {
      const int ih = ih0 + rr[k];
      const bool ok = (unsigned)ih < (unsigned)H && ncol[k] > 0;
      offb[k] = ok ? (unsigned)(ih * W + col[k]) * 4u : 0u;
      live |= ok ? (1u << k) : 0u;
    }
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const char* pb = reinterpret_cast<const char*>(plane0 + (size_t)(c < nch ? c : 0) * (H * W));
#pragma unroll
      for (int k = 0; k < SPC; ++k) {
        const f4u t = *reinterpret_cast<const f4u*>(pb + offb[k]);
        v[c * SPC + k] = f32x4{t.x, t.y, t.z, t.w};
      }
    }
  }
  // aff = LDS copy of the producer's per-channel affine ([C] scales then [C] shifts, stage_affine) or null;
  // c0 = first channel of this chunk.  Chunks that hold no image data were loaded from a valid dummy address (finite
  // activations) and are multiplied by 0: they come out as exact zeros.
  // bf16: round the committed operand to bfloat16 (PGV_COMPUTE_BF16 in kernels that keep multiplying on the fp32 pipe)
  __device__ __forceinline__ void commit(float* __restrict__ tile, const float* __restrict__ aff, int C, int c0,
                                         int nch, int tid, bool bf16 = false) {
    float* lane_tile = tile + 4 * tid;
    float sc[CK], sh[CK];
#pragma unroll
    for (int c = 0; c < CK; ++c) {  // all LDS reads first: one latency for the whole chunk
      const int cg = min(c0 + c, C - 1);
      sc[c] = aff ? aff[cg] : 1.0f;
      sh[c] = aff ? aff[C + cg] : 0.0f;
    }
    float lk[SPC];
#pragma unroll
    for (int k = 0; k < SPC; ++k) lk[k] = ((live >> k) & 1u) ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const float cm = c < nch ? 1.0f : 0.0f;
#pragma unroll
      for (int k = 0; k < SPC; ++k) {
        if (256 * (k + 1) <= PC || tid + 256 * k < PC) {
          const f32x4 t = v[c * SPC + k];
          const float m = sc[c] * (lk[k] * cm), a = sh[c] * (lk[k] * cm);
          f32x4 x;
          if (NP == 0) {
            x.x = fmaf(t.x, m, a);
            x.y = fmaf(t.y, m, a);
            x.z = fmaf(t.z, m, a);
            x.w = fmaf(t.w, m, a);
          } else {
            // partial chunk (ncol == NP): the window was shifted back by 4-NP floats; floats beyond NP are pad zeros
            const bool part = ncol[k] < 4;
            const float e0 = part ? t[(4 - NP) & 3] : t.x;
            const float e1 = part ? t[(5 - NP) & 3] : t.y;
            const float e2 = part ? t[(6 - NP) & 3] : t.z;
            const float m1 = (part && NP < 2) ? 0.f : m, a1 = (part && NP < 2) ? 0.f : a;
            const float m2 = (part && NP < 3) ? 0.f : m, a2 = (part && NP < 3) ? 0.f : a;
            const float m3 = part ? 0.f : m, a3 = part ? 0.f : a;
            x.x = fmaf(e0, m, a);
            x.y = fmaf(e1, m1, a1);
            x.z = fmaf(e2, m2, a2);
            x.w = fmaf(t.w, m3, a3);
          }
          if (bf16) x = f32x4{round_bf16(x.x), round_bf16(x.y), round_bf16(x.z), round_bf16(x.w)};
          *reinterpret_cast<f32x4*>(lane_tile + c * CSTRIDE + 1024 * k) = x;
        }
      }
    }
  }
};

// Same interface, flat chunk list (chunk e = tid + 256*j over all CK*PC chunks of the tile, channel-major): used when a
// channel has far fewer than 256 chunks and the per-channel slots of BandPrefetch would leave most lanes idle.  One
// packed descriptor register per slot; the affine parameters are per-lane LDS reads.
template <int CK, int ROWS, int W, int WP, int H, int MINPAD = 2, int CSTRIDE = ROWS * WP>
struct FlatPrefetch {
  static constexpr int QR = WP / 4;
  static constexpr int PC = ROWS * QR;
  static_assert(CSTRIDE >= ROWS * WP && CSTRIDE % 4 == 0, "channel stride");
  static constexpr int ITEMS = CK * PC;
  static constexpr int NPF = (ITEMS + 255) / 256;
  static constexpr int NP = W % 4;
  static_assert(WP % 4 == 0 && WP >= W + MINPAD, "row stride");
  static_assert(ROWS < 256 && CK <= 256 && QR < 4096, "descriptor fields");
  f32x4 v[NPF];
  unsigned meta[NPF];  // rr | ncol << 8 | c << 12 | q << 20   (ncol = 0: pad chunk or idle lane)
  unsigned live;

  __device__ __forceinline__ void init(int tid) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int e = tid + 256 * j;
      const int ee = min(e, ITEMS - 1);
      const int rowi = ee / QR, q = ee - rowi * QR;
      const int c = rowi / ROWS, rr = rowi - c * ROWS;
      const int nc = e < ITEMS ? min(max(W - 4 * q, 0), 4) : 0;
      meta[j] = (unsigned)rr | ((unsigned)nc << 8) | ((unsigned)c << 12) | ((unsigned)q << 20);
    }
  }
  __device__ __forceinline__ void issue(const float* __restrict__ plane0, int ih0, int nch) {
    live = 0;
    const char* pb = reinterpret_cast<const char*>(plane0);
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int rr = meta[j] & 255, nc = (meta[j] >> 8) & 15, c = (meta[j] >> 12) & 255, q = meta[j] >> 20;
      const int ih = ih0 + rr;
      const bool ok = (unsigned)ih < (unsigned)H && nc > 0 && c < nch;
      const int col = 4 * q - ((NP != 0 && nc < 4) ? 4 - NP : 0);
      const unsigned off = ok ? (unsigned)((c * H + ih) * W + col) * 4u : 0u;
      live |= ok ? (1u << j) : 0u;
      const f4u t = *reinterpret_cast<const f4u*>(pb + off);
      v[j] = f32x4{t.x, t.y, t.z, t.w};
    }
  }
  // f(channel, tile row, chunk of the row, x): the finished values of every chunk this lane holds - affine applied, exact
  // zeros outside the image and behind the end of a row - for kernels that commit them in another format than the fp32
  // tile (conv_wgrad_bf16.hip: bfloat16 planes)
  template <class F>
  __device__ __forceinline__ void each(const float* __restrict__ aff, int C, int c0, int tid, F&& f) const {
    float sc[NPF], sh[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int cg = min(c0 + (int)((meta[j] >> 12) & 255), C - 1);
      sc[j] = aff ? aff[cg] : 1.0f;
      sh[j] = aff ? aff[C + cg] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      if (256 * (j + 1) <= ITEMS || tid + 256 * j < ITEMS) {
        const f32x4 t = v[j];
        const bool on = (live >> j) & 1u;
        const float m = on ? sc[j] : 0.f, a = on ? sh[j] : 0.f;
        f32x4 x;
        if (NP == 0) {
          x.x = fmaf(t.x, m, a);
          x.y = fmaf(t.y, m, a);
          x.z = fmaf(t.z, m, a);
          x.w = fmaf(t.w, m, a);
        } else {
          const bool part = ((meta[j] >> 8) & 15) < 4;
          const float e0 = part ? t[(4 - NP) & 3] : t.x;
          const float e1 = part ? t[(5 - NP) & 3] : t.y;
          const float e2 = part ? t[(6 - NP) & 3] : t.z;
          const float m1 = (part && NP < 2) ? 0.f : m, a1 = (part && NP < 2) ? 0.f : a;
          const float m2 = (part && NP < 3) ? 0.f : m, a2 = (part && NP < 3) ? 0.f : a;
          const float m3 = part ? 0.f : m, a3 = part ? 0.f : a;
          x.x = fmaf(e0, m, a);
          x.y = fmaf(e1, m1, a1);
          x.z = fmaf(e2, m2, a2);
          x.w = fmaf(t.w, m3, a3);
        }
        f((int)((meta[j] >> 12) & 255), (int)(meta[j] & 255), (int)(meta[j] >> 20), x);
      }
    }
  }
  __device__ __forceinline__ void commit(float* __restrict__ tile, const float* __restrict__ aff, int C, int c0,
                                         int /*nch*/, int tid, bool bf16 = false) {
    float* lane_tile = tile + 4 * tid;
    float sc[NPF], sh[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int cg = min(c0 + (int)((meta[j] >> 12) & 255), C - 1);
      sc[j] = aff ? aff[cg] : 1.0f;
      sh[j] = aff ? aff[C + cg] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      if (256 * (j + 1) <= ITEMS || tid + 256 * j < ITEMS) {
        const f32x4 t = v[j];
        const bool on = (live >> j) & 1u;
        const float m = on ? sc[j] : 0.f, a = on ? sh[j] : 0.f;
        f32x4 x;
        if (NP == 0) {
          x.x = fmaf(t.x, m, a);
          x.y = fmaf(t.y, m, a);
          x.z = fmaf(t.z, m, a);
          x.w = fmaf(t.w, m, a);
        } else {
          const bool part = ((meta[j] >> 8) & 15) < 4;
          const float e0 = part ? t[(4 - NP) & 3] : t.x;
          const float e1 = part ? t[(5 - NP) & 3] : t.y;
          const float e2 = part ? t[(6 - NP) & 3] : t.z;
          const float m1 = (part && NP < 2) ? 0.f : m, a1 = (part && NP < 2) ? 0.f : a;
          const float m2 = (part && NP < 3) ? 0.f : m, a2 = (part && NP < 3) ? 0.f : a;
          const float m3 = part ? 0.f : m, a3 = part ? 0.f : a;
          x.x = fmaf(e0, m, a);
          x.y = fmaf(e1, m1, a1);
          x.z = fmaf(e2, m2, a2);
          x.w = fmaf(t.w, m3, a3);
        }
        if (bf16) x = f32x4{round_bf16(x.x), round_bf16(x.y), round_bf16(x.z), round_bf16(x.w)};
        *reinterpret_cast<f32x4*>(lane_tile + 1024 * j + (int)((meta[j] >> 12) & 255) * (CSTRIDE - PC * 4)) = x;
      }
    }
  }
};

// channel slots when they are at least 80 % occupied, the flat list otherwise
template <int CK, int ROWS, int W, int WP, int H, int MINPAD = 2, int CSTRIDE = ROWS * WP>
struct PickPrefetch {
  static constexpr int PC = ROWS * (WP / 4);
  static constexpr int SPC = (PC + 255) / 256;
  static constexpr bool kChannelSlots = PC * 10 >= SPC * 256 * 8;
  using type = typename std::conditional<kChannelSlots, BandPrefetch<CK, ROWS, W, WP, H, MINPAD, CSTRIDE>,
                                         FlatPrefetch<CK, ROWS, W, WP, H, MINPAD, CSTRIDE>>::type;
};

}  // namespace
