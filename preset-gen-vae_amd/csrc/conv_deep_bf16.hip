// bf16-NATIVE kernels of the deep k4 s2 p2 layers (17x23, 9x12 and 5x7 planes, 64..512 channels; model/encoder.py:249-255,
// model/decoder.py:205-210) for PGV_COMPUTE_BF16.  The fp32-image kernels of conv_deep.hip run this mode at 0.5-0.6 of
// their fp32 time: they are bound by the instructions AROUND the matrix instruction (fp32 fragment reads, packing, one
// half-empty K = 32 instruction per channel and slab), not by it.  Here both operands are rounded ONCE, on their way into
// LDS, and live there in the layout v_mfma_f32_16x16x32_bf16 reads with one 16-byte load per fragment:
//   * channel-innermost images: a plane pixel is 16 channels = two 16-byte halves (8 channels each), so the B fragment of
//     output pixel n, kernel row kh, for lane group kq = kernel column kw is the half-pixel at (2oh+kh, 2ow+kq): the K = 32
//     of one instruction is 8 channels x the 4 kernel columns of one kernel row.  The half of channel group g is stored at
//     g ^ bit3(pixel index), which makes the 16-lane groups of ds_read_b128 conflict free without padding the pixel;
//   * weights come from a bf16 SHADOW of the layer's weight, [cs][cb/8][16 taps][8 channels] (pgv_conv_weight_shadow, one
//     launch per layer and step): a slab of a weight row is 512 contiguous bytes, copied to LDS as it is.  The shadow
//     halves the weight stream, the bound of these layers (every sample group streams the whole weight through L2).
// One workgroup = 64 output channels x NS samples; 8 waves = 4 M tiles x 2 halves of the pixel tiles, every wave over the
// whole K: no reduction between waves (a first version split K over the waves, 0.42 KB of fragments per instruction
// instead of 1.3, and paid 3.6 us of LDS reduction rounds per workgroup for it).
// (round 6: this file keeps the forward / input-gradient kernels; the weight gradients are in conv_deep_wgrad_bf16.hip, the
// weight shadows in conv_shadow.hip)
#include "conv_tile.h"
#include "conv_deep_common.h"

static unsigned long long* g_deep_bf16_stamps = nullptr;
extern "C" void pgv_dbg_set_deep_bf16_stamps(void* p) { g_deep_bf16_stamps = (unsigned long long*)p; }
unsigned long long* pgv_deep_bf16_stamps() { return g_deep_bf16_stamps; }
// A/B knobs of the timing scripts (pgv_dbg_set_deep_bf16_variant; no effect on results beyond the summation order):
// 8 = deep / 1x1 weight gradients stay on the fp32-image kernels, 64 = 9x12 weight gradient on 16-sample blocks,
// 128 = 17x23 weight gradient on 8-sample blocks
static int g_deep_bf16_dbg = 0;
extern "C" int pgv_dbg_set_deep_bf16_variant(int v) {
  const int old = g_deep_bf16_dbg;
  g_deep_bf16_dbg = v;
  return old;
}
int pgv_deep_bf16_dbg() { return g_deep_bf16_dbg; }

namespace {

typedef unsigned short u16;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// ---------------------------------------------------------------------------------------------------------------
// DOWN, K split over the waves (the 5x7 and 17x23 layers): 8 waves = 8 K groups (2 channel groups x 4 kernel rows of a
// 16-channel slab), each with the full 64 x N register tile - (4 + NT) fragment reads per 4 NT instructions - and LDS
// reduction rounds before the epilogue.  Measured against the N-split form below: 27 / 27 us against 54 / 46 us on 5x7 /
// 17x23, 34 against 28 us on 9x12 (whose 9-tile accumulator spills here).
template <int H_, int W_, int NS_>
struct DownBK {
  static constexpr int H = H_, W = W_, NS = NS_;
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, HW = H * W, HP = 2 * Hs + 2;
  // image pixels (rows / columns -2 .. 2Hs-1 / 2Ws-1) and plane strides in pixels: scratch/deep_bf16_strides.py - no bank
  // conflicts on the 5x7 planes, 0.11 / 0.14 extra LDS cycles per fragment read on 9x12 / 17x23
  static constexpr int WP = (H == 5 && W == 7) ? 12 : (H == 9 && W == 12) ? 23 : (H == 17 && W == 23) ? 28 : 2 * Ws + 2;
  static constexpr int PLANE = (H == 5 && W == 7) ? 104 : (H == 9 && W == 12) ? 278 : HP * WP;
  static_assert(WP >= 2 * Ws + 2 && PLANE >= HP * WP, "padded plane");
  static constexpr int N = NS * P, NT = (N + 15) / 16;
  static constexpr int CK = 16;                            // channels per slab = two groups of 8
  static constexpr int A_ROW = 2 * 256 + 32;               // bytes per weight row of a slab: conflict-free fragment reads
  static constexpr int A_BYTES = 64 * A_ROW;
  static constexpr int B_BYTES = NS * PLANE * 32;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int QA = 64 * 32 / 512;                 // 16-byte weight pieces per thread and slab
  static constexpr int QUADS = (HW + 3) / 4;               // pixel quads of a plane (the last one shifted back)
  static constexpr int ITEMS = NS * 8 * QUADS;             // (sample, channel pair, quad)
  static constexpr int QB = (ITEMS + 511) / 512;
  static constexpr int RED_BYTES = 8 * NT * 1024, OUT_BYTES = NS * 64 * P * 4;
  static_assert(STAGE % 16 == 0 && HW >= 4, "alignment");
  static_assert(RED_BYTES + OUT_BYTES <= 2 * STAGE, "reduction and output tile fit the stages");
};

template <class G>
__global__ __launch_bounds__(512) void deep_down_bf16_ksplit_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift,
                                                             const u16* __restrict__ wsh, const float* __restrict__ bias,
                                                             int act, float slope, float* __restrict__ out,
                                                             double* __restrict__ stats, int groups, int stat_stride,
                                                             pgv_bn_src in_bn,
                                                             unsigned long long* __restrict__ stamps) {
  constexpr int NT = G::NT, HW = G::HW, P = G::P, NS = G::NS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  float* aff = reinterpret_cast<float*>(ldsb + 2 * G::STAGE);   // [2*CB]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cbg = wave >> 2, kh = wave & 3;
  int mb, grp;
  deep_block(CS / 64, groups, mb, grp);
  const int cs0 = mb * 64, b0 = grp * NS;
  BSTAMP(0);

  // zero both stages' images once (the data pixels are rewritten every slab, the padding never)
  for (int i = tid; i < G::B_BYTES / 16; i += 512) {
    reinterpret_cast<u32x4*>(ldsb + G::A_BYTES)[i] = u32x4{0, 0, 0, 0};
    reinterpret_cast<u32x4*>(ldsb + G::STAGE + G::A_BYTES)[i] = u32x4{0, 0, 0, 0};
  }
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
  const int cbgs = CB / 8;
  int a_src[G::QA], a_dst[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = tid + 512 * i, row = q >> 5, f = q & 31;
    a_src[i] = ((cs0 + row) * cbgs) * 256 + f * 16;   // bytes into the shadow (+ 512 per slab)
    a_dst[i] = row * G::A_ROW + f * 16;
  }
  int b_src[G::QB], b_dst[G::QB][4], b_cp[G::QB];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 512 * i, G::ITEMS - 1);
    b_ok[i] = tid + 512 * i < G::ITEMS;
    const int si = q / (8 * G::QUADS), rem = q - si * (8 * G::QUADS), cp = rem / G::QUADS, qi = rem - cp * G::QUADS;
    const int p0 = min(4 * qi, HW - 4);
    const int bs = min(b0 + si, B - 1);   // partial last group: duplicate the last sample (masked at the store)
    b_src[i] = (bs * CB + 2 * cp) * HW + p0;
    b_cp[i] = cp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pe = p0 + e, r = pe / G::W, c = pe - r * G::W;
      const int px = si * G::PLANE + (r + 2) * G::WP + c + 2;
      b_dst[i][e] = px * 32 + (((cp >> 2) ^ ((px >> 3) & 1)) * 16) + (cp & 3) * 4;
    }
  }
  // ---- fragment coordinates (bytes)
  const int a_frag = m * G::A_ROW + cbg * 256 + kh * 64 + kq * 16;   // M tile t: + t * 16 rows
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = min(t * 16 + m, G::N - 1);
    const int si = n / P, pix = n - si * P, oh = pix / G::Ws, ow = pix - oh * G::Ws;
    const int px = si * G::PLANE + (2 * oh + kh) * G::WP + 2 * ow + kq;
    boff[t] = G::A_BYTES + px * 32 + ((cbg ^ ((px >> 3) & 1)) * 16);
  }
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[G::QA];
  f4u rb[G::QB][2];
  auto issue = [&](int slab) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i)
      ra[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wsh) + a_src[i] + slab * 512);
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const float* p = big + b_src[i] + slab * (16 * HW);
      rb[i][0] = *reinterpret_cast<const f4u*>(p);
      rb[i][1] = *reinterpret_cast<const f4u*>(p + HW);
    }
  };
  auto commit = [&](unsigned char* st, int slab) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) *reinterpret_cast<u32x4*>(st + a_dst[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int c = slab * 16 + 2 * b_cp[i];
      const float s0 = aff[c], s1 = aff[c + 1], h0 = aff[CB + c], h1 = aff[CB + c + 1];
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<unsigned*>(st + G::A_BYTES + b_dst[i][e]) =
              pack_bf16x2(fmaf(rb[i][0][e], s0, h0), fmaf(rb[i][1][e], s1, h1));
      }
    }
  };

  const int nslab = CB / 16;
  BSTAMP(1);
  issue(0);
  __syncthreads();   // images zeroed, affine staged
  BSTAMP(2);
  commit(ldsb, 0);
  if (nslab > 1) issue(1);
  __syncthreads();
  BSTAMP(3);
#pragma unroll 1
  for (int s = 0; s < nslab; ++s) {
    const unsigned char* st = ldsb + (s & 1) * G::STAGE;
    u32x4 af[4], bf[NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(st + a_frag + mt * 16 * G::A_ROW);
#pragma unroll
    for (int t = 0; t < NT; ++t) bf[t] = *reinterpret_cast<const u32x4*>(st + boff[t]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] = mfma_bf16_k32(af[mt], bf[t], acc[mt][t]);
    if (s + 1 < nslab) {   // the next slab goes to the other stage under the running matrix pipe
      commit(ldsb + ((s + 1) & 1) * G::STAGE, s + 1);
      if (s + 2 < nslab) issue(s + 2);
    }
#pragma unroll
    for (int mt = 2; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[mt][t] = mfma_bf16_k32(af[mt], bf[t], acc[mt][t]);
    __syncthreads();
  }

  // ---- the 8 K groups' partial tiles are added up M tile by M tile (a round: every wave stores its 16 x N partial tile,
  // wave w sums the N tiles t = w, w + 8 in the fixed order of the waves - deterministic - and finishes them: bias,
  // activation, into the [sample][channel][P] output tile); the stages are free after the last slab's barrier
  BSTAMP(4);
  const pgv_act_params ap = pgv_act_setup(act, slope);
  f32x4* red = reinterpret_cast<f32x4*>(ldsb);
  float* otile = reinterpret_cast<float*>(ldsb + G::RED_BYTES);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int t = 0; t < NT; ++t) red[(wave * NT + t) * 64 + lane] = acc[mt][t];
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < (NT + 7) / 8; ++tt) {
      const int t = wave + 8 * tt;
      if (t < NT) {
        f32x4 v = red[t * 64 + lane];
#pragma unroll
        for (int u = 1; u < 8; ++u) v += red[(u * NT + t) * 64 + lane];
        const int n = t * 16 + m, si = n / P, pix = n - si * P, cl = mt * 16 + 4 * kq;
        if (n < G::N) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            otile[(si * 64 + cl + i) * P + pix] = pgv_act_apply(v[i] + (bias ? bias[cs0 + cl + i] : 0.f), ap);
        }
      }
    }
    __syncthreads();
  }
  BSTAMP(5);
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
  BSTAMP(6);
#pragma unroll
  for (int si = 0; si < NS; ++si) {
    if (b0 + si < B) {
      float* dst = out + ((int64_t)(b0 + si) * CS + cs0) * P;
      const float* src = otile + si * 64 * P;
      for (int i = tid; i < 64 * P; i += 512) dst[i] = src[i];
    }
  }
  BSTAMP(7);
}

// ---------------------------------------------------------------------------------------------------------------
// DOWN: out[b,cs,oh,ow] = act(bias[cs] + sum_{cb,kh,kw} w[cs,cb,kh,kw] * x'[b,cb,2oh-2+kh,2ow-2+kw])
template <int H_, int W_, int NS_>
struct DownB {
  static constexpr int H = H_, W = W_, NS = NS_;
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, HW = H * W, HP = 2 * Hs + 2;
  // image pixels (rows / columns -2 .. 2Hs-1 / 2Ws-1) and plane strides in pixels: scratch/deep_bf16_strides.py - no bank
  // conflicts on the 5x7 planes, 0.11 / 0.14 extra LDS cycles per fragment read on 9x12 / 17x23
  static constexpr int WP = (H == 5 && W == 7) ? 12 : (H == 9 && W == 12) ? 23 : (H == 17 && W == 23) ? 28 : 2 * Ws + 2;
  static constexpr int PLANE = (H == 5 && W == 7) ? 104 : (H == 9 && W == 12) ? 278 : HP * WP;
  static_assert(WP >= 2 * Ws + 2 && PLANE >= HP * WP, "padded plane");
  static constexpr int N = NS * P, NT = (N + 15) / 16;
  static constexpr int CK = 16;                            // channels per slab = two groups of 8
  static constexpr int A_ROW = 2 * 256 + 32;               // bytes per weight row of a slab: conflict-free fragment reads
  static constexpr int A_BYTES = 64 * A_ROW;
  static constexpr int B_BYTES = NS * PLANE * 32;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int QA = 64 * 32 / 512;                 // 16-byte weight pieces per thread and slab
  static constexpr int QUADS = (HW + 3) / 4;               // pixel quads of a plane (the last one shifted back)
  static constexpr int ITEMS = NS * 8 * QUADS;             // (sample, channel pair, quad)
  static constexpr int QB = (ITEMS + 511) / 512;
  static constexpr int TH = (NT + 1) / 2;                  // pixel tiles of a wave
  static constexpr int OUT_BYTES = NS * 64 * P * 4;
  static_assert(STAGE % 16 == 0 && HW >= 4, "alignment");
  static_assert(OUT_BYTES <= 2 * STAGE, "the output tile fits the stages");
};

template <class G>
__global__ __launch_bounds__(512) void deep_down_bf16_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift,
                                                             const u16* __restrict__ wsh, const float* __restrict__ bias,
                                                             int act, float slope, float* __restrict__ out,
                                                             double* __restrict__ stats, int groups, int stat_stride,
                                                             pgv_bn_src in_bn,
                                                             unsigned long long* __restrict__ stamps) {
  constexpr int NT = G::NT, HW = G::HW, P = G::P, NS = G::NS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  float* aff = reinterpret_cast<float*>(ldsb + 2 * G::STAGE);   // [2*CB]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  // 8 waves = 4 M tiles x 2 halves of the pixel tiles; every wave runs the whole K of its tiles: no reduction between waves
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), mt = wave & 3, nh = wave >> 2;
  int mb, grp;
  deep_block(CS / 64, groups, mb, grp);
  const int cs0 = mb * 64, b0 = grp * NS;
  BSTAMP(0);

  // zero both stages' images once (the data pixels are rewritten every slab, the padding never)
  for (int i = tid; i < G::B_BYTES / 16; i += 512) {
    reinterpret_cast<u32x4*>(ldsb + G::A_BYTES)[i] = u32x4{0, 0, 0, 0};
    reinterpret_cast<u32x4*>(ldsb + G::STAGE + G::A_BYTES)[i] = u32x4{0, 0, 0, 0};
  }
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
  const int cbgs = CB / 8;
  int a_src[G::QA], a_dst[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = tid + 512 * i, row = q >> 5, f = q & 31;
    a_src[i] = ((cs0 + row) * cbgs) * 256 + f * 16;   // bytes into the shadow (+ 512 per slab)
    a_dst[i] = row * G::A_ROW + f * 16;
  }
  int b_src[G::QB], b_dst[G::QB][4], b_cp[G::QB];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 512 * i, G::ITEMS - 1);
    b_ok[i] = tid + 512 * i < G::ITEMS;
    const int si = q / (8 * G::QUADS), rem = q - si * (8 * G::QUADS), cp = rem / G::QUADS, qi = rem - cp * G::QUADS;
    const int p0 = min(4 * qi, HW - 4);
    const int bs = min(b0 + si, B - 1);   // partial last group: duplicate the last sample (masked at the store)
    b_src[i] = (bs * CB + 2 * cp) * HW + p0;
    b_cp[i] = cp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pe = p0 + e, r = pe / G::W, c = pe - r * G::W;
      const int px = si * G::PLANE + (r + 2) * G::WP + c + 2;
      b_dst[i][e] = px * 32 + (((cp >> 2) ^ ((px >> 3) & 1)) * 16) + (cp & 3) * 4;
    }
  }
  // ---- fragment coordinates (bytes): this wave's tiles nh * TH + tt; per kernel row kh and channel group g
  constexpr int TH = G::TH;
  const int a_frag = (mt * 16 + m) * G::A_ROW + kq * 16;   // + 256 per channel group, + 64 per kernel row
  int boff[TH][4][2];
#pragma unroll
  for (int tt = 0; tt < TH; ++tt) {
    const int n = min((nh * TH + tt) * 16 + m, G::N - 1);
    const int si = n / P, pix = n - si * P, oh = pix / G::Ws, ow = pix - oh * G::Ws;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const int px = si * G::PLANE + (2 * oh + kh) * G::WP + 2 * ow + kq;
#pragma unroll
      for (int g = 0; g < 2; ++g) boff[tt][kh][g] = G::A_BYTES + px * 32 + ((g ^ ((px >> 3) & 1)) * 16);
    }
  }
  const int ntl = min(TH, NT - nh * TH);   // tiles of this wave
  f32x4 acc[TH];
#pragma unroll
  for (int tt = 0; tt < TH; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[cs0 + mt * 16 + 4 * kq + i] : 0.f;

  u32x4 ra[G::QA];
  f4u rb[G::QB][2];
  auto issue = [&](int slab) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i)
      ra[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wsh) + a_src[i] + slab * 512);
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const float* p = big + b_src[i] + slab * (16 * HW);
      rb[i][0] = *reinterpret_cast<const f4u*>(p);
      rb[i][1] = *reinterpret_cast<const f4u*>(p + HW);
    }
  };
  auto commit = [&](unsigned char* st, int slab) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) *reinterpret_cast<u32x4*>(st + a_dst[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int c = slab * 16 + 2 * b_cp[i];
      const float s0 = aff[c], s1 = aff[c + 1], h0 = aff[CB + c], h1 = aff[CB + c + 1];
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<unsigned*>(st + G::A_BYTES + b_dst[i][e]) =
              pack_bf16x2(fmaf(rb[i][0][e], s0, h0), fmaf(rb[i][1][e], s1, h1));
      }
    }
  };

  const int nslab = CB / 16;
  BSTAMP(1);
  issue(0);
  __syncthreads();   // images zeroed, affine staged
  BSTAMP(2);
  commit(ldsb, 0);
  if (nslab > 1) issue(1);
  __syncthreads();
  BSTAMP(3);
#pragma unroll 1
  for (int s = 0; s < nslab; ++s) {
    const unsigned char* st = ldsb + (s & 1) * G::STAGE;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(st + a_frag + g * 256 + kh * 64);
#pragma unroll
        for (int tt = 0; tt < TH; ++tt) {
          if (tt < ntl) acc[tt] = mfma_bf16_k32(af, *reinterpret_cast<const u32x4*>(st + boff[tt][kh][g]), acc[tt]);
        }
      }
      if (g == 0 && s + 1 < nslab) {   // the next slab goes to the other stage under the running matrix pipe
        commit(ldsb + ((s + 1) & 1) * G::STAGE, s + 1);
        if (s + 2 < nslab) issue(s + 2);
      }
    }
    __syncthreads();
  }
  BSTAMP(4);
  // ---- epilogue: bias, activation, into the [sample][channel][P] output tile (the stages are free after the last barrier)
  const pgv_act_params ap = pgv_act_setup(act, slope);
  float* otile = reinterpret_cast<float*>(ldsb);
#pragma unroll
  for (int tt = 0; tt < TH; ++tt) {
    const int n = (nh * TH + tt) * 16 + m, si = n / P, pix = n - si * P, cl = mt * 16 + 4 * kq;
    if (tt < ntl && n < G::N) {
#pragma unroll
      for (int i = 0; i < 4; ++i) otile[(si * 64 + cl + i) * P + pix] = pgv_act_apply(acc[tt][i] + bv[i], ap);
    }
  }
  __syncthreads();
  BSTAMP(5);
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
  BSTAMP(6);
#pragma unroll
  for (int si = 0; si < NS; ++si) {
    if (b0 + si < B) {
      float* dst = out + ((int64_t)(b0 + si) * CS + cs0) * P;
      const float* src = otile + si * 64 * P;
      for (int i = tid; i < 64 * P; i += 512) dst[i] = src[i];
    }
  }
  BSTAMP(7);
}

template <int H, int W, int NS, bool KSPLIT>
int launch_deep_down_bf16(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                          const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                          const pgv_bn_src* bn) {
  using G = typename std::conditional<KSPLIT, DownBK<H, W, NS>, DownB<H, W, NS>>::type;
  if (d->Cs % 64 || d->Cb % 16 || !d->w_shadow) return 0;
  if ((int64_t)d->B * d->Cb * G::HW * 4 >= (int64_t)1 << 31 || (int64_t)d->Cs * d->Cb * 32 >= (int64_t)1 << 31) return 0;
  const size_t bytes = 2 * (size_t)G::STAGE + sizeof(float) * (2 * (size_t)d->Cb + 8);
  if (bytes > (size_t)kMaxLds) return 0;
  typedef void (*kern_t)(int, int, int, const float*, const float*, const float*, const u16*, const float*, int, float, float*,
                         double*, int, int, pgv_bn_src, unsigned long long*);
  kern_t kern;
  if constexpr (KSPLIT)
    kern = deep_down_bf16_ksplit_kernel<G>;
  else
    kern = deep_down_bf16_kernel<G>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_down_deep_bf16");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_deep_bf16: memset failed");
    return PGV_E_LAUNCH;
  }
  const int groups = (d->B + NS - 1) / NS;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * (d->Cs / 64))), dim3(512), bytes, st, d->B, d->Cb, d->Cs, big, in_scale,
                     in_shift, (const u16*)d->w_shadow, bias, act, slope, out, stats, groups,
                     (d->flags & PGV_STATS_COPIES) ? 2 * d->Cs : 0, bn ? *bn : pgv_no_bn(), g_deep_bf16_stamps);
  PGV_CHECK_LAUNCH("conv_down_deep_bf16");
  return 1;
}


// ---------------------------------------------------------------------------------------------------------------
// UP: out[b,cb,ih,iw] = act(bias[cb] + sum_{cs,kh,kw} w[cs,cb,kh,kw] * s'[b,cs,oh,ow]),  ih = 2oh-2+kh, iw = 2ow-2+kw.
// Output pixel (ih,iw) = (2u+ph, 2v+pw) only meets the taps kh = ph+2th, kw = pw+2tw (th,tw in {0,1}) at oh = u+1-th,
// ow = v+1-tw: four 2x2-tap convolutions, one per output phase.  GEMM per phase: M = cb (32 per workgroup: the layers have
// 64..256 big channels, and every sample group streams the whole weight), K = (cs, 4 taps), N = the phase's output pixels
// of NS samples.  The K = 32 of one instruction is 8 small channels x the 4 taps of the phase (lane group kq = 2th+tw), the
// B fragment the quarter-pixel (32 channels per pixel) at (u+1-th, v+1-tw) of the zero-padded small image.  The 8 waves
// split N - wave = (phase, half of the phase's tiles), every wave runs the whole K: no reduction between waves.
template <int H_, int W_, int NS_>
struct UpB {
  static constexpr int H = H_, W = W_, NS = NS_;
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, HW = H * W;
  // small image with a zero row below / column right; strides in pixels (scratch/deep_bf16_strides.py: about one extra
  // LDS cycle per fragment read remains on every layer - the phases' short pixel rows do not tile the 16 slots)
  static constexpr int SWP = (H == 5 && W == 7) ? 5 : (H == 9 && W == 12) ? 11 : (H == 17 && W == 23) ? 14 : Ws + 1;
  static constexpr int SPLANE = (H == 5 && W == 7) ? 23 : (H == 9 && W == 12) ? 70 : (H == 17 && W == 23) ? 141 : (Hs + 1) * SWP;
  static_assert(SWP >= Ws + 1 && SPLANE >= (Hs + 1) * SWP, "padded plane");
  static constexpr int hu(int p) { return (p >> 1) ? H / 2 : (H + 1) / 2; }
  static constexpr int wu(int p) { return (p & 1) ? W / 2 : (W + 1) / 2; }
  static constexpr int cnt(int p) { return NS * hu(p) * wu(p); }
  static constexpr int ntp(int p) { return (cnt(p) + 15) / 16; }
  static constexpr int TMAX = (ntp(0) + 1) / 2;            // tiles of one wave (phase 0 has the most pixels)
  static constexpr int CK = 32;                            // small channels per slab = four groups of 8
  static constexpr int MT = 32;                            // big channels per workgroup
  static constexpr int A_ROW = 4 * 256 + 32;               // bytes per weight row of a slab
  static constexpr int A_BYTES = MT * A_ROW;
  static constexpr int B_BYTES = NS * SPLANE * 64;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int QA = MT * 64 / 512;
  static constexpr int QUADS = (P + 3) / 4;
  static constexpr int ITEMS = NS * 16 * QUADS;            // (sample, channel pair, pixel quad)
  static constexpr int QB = (ITEMS + 511) / 512;
  static constexpr int OUT_BYTES = NS * MT * HW * 4;
  static_assert(STAGE % 16 == 0 && P >= 4, "alignment");
  static_assert(OUT_BYTES <= 2 * STAGE, "output tile fits the stages");
};

template <class G>
__global__ __launch_bounds__(512) void deep_up_bf16_kernel(int B, int CB, int CS, const float* __restrict__ small_in,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift,
                                                           const u16* __restrict__ wsh, const float* __restrict__ bias,
                                                           int act, float slope, float* __restrict__ out,
                                                           double* __restrict__ stats, int groups, int stat_stride,
                                                           pgv_bn_src in_bn) {
  constexpr int HW = G::HW, P = G::P, NS = G::NS, TMAX = G::TMAX, MT = G::MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  float* aff = reinterpret_cast<float*>(ldsb + 2 * G::STAGE);   // [2*CS]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // phases 0..3 have decreasing pixel counts: the two waves of a SIMD (w, w + 4) take phases p and 3 - p
  const int ph = wave < 4 ? wave : 7 - wave, half = wave >> 2;
  int mb, grp;
  deep_block(CB / MT, groups, mb, grp);
  const int cb0 = mb * MT, b0 = grp * NS;

  for (int i = tid; i < G::B_BYTES / 16; i += 512) {
    reinterpret_cast<u32x4*>(ldsb + G::A_BYTES)[i] = u32x4{0, 0, 0, 0};
    reinterpret_cast<u32x4*>(ldsb + G::STAGE + G::A_BYTES)[i] = u32x4{0, 0, 0, 0};
  }
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
  const int csgs = CS / 8;
  int a_src[G::QA], a_dst[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = tid + 512 * i, row = q >> 6, f = q & 63;
    a_src[i] = ((cb0 + row) * csgs) * 256 + f * 16;   // bytes into the up shadow (+ 1024 per slab)
    a_dst[i] = row * G::A_ROW + f * 16;
  }
  int b_src[G::QB], b_dst[G::QB][4], b_cp[G::QB];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 512 * i, G::ITEMS - 1);
    b_ok[i] = tid + 512 * i < G::ITEMS;
    const int si = q / (16 * G::QUADS), rem = q - si * (16 * G::QUADS), cp = rem / G::QUADS, qi = rem - cp * G::QUADS;
    const int p0 = min(4 * qi, P - 4);
    const int bs = min(b0 + si, B - 1);
    b_src[i] = (bs * CS + 2 * cp) * P + p0;
    b_cp[i] = cp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pe = p0 + e, oh = pe / G::Ws, ow = pe - oh * G::Ws;
      const int px = si * G::SPLANE + oh * G::SWP + ow;
      b_dst[i][e] = px * 64 + (((cp >> 2) ^ ((px >> 2) & 3)) * 16) + (cp & 3) * 4;
    }
  }
  // ---- this wave's tiles: pixels [16 (t0 + t), +16) of phase ph's list (sample, u, v)
  const int phh = ph >> 1, pww = ph & 1;
  const int hu = phh ? G::H / 2 : (G::H + 1) / 2, wu = pww ? G::W / 2 : (G::W + 1) / 2;
  const int cnt = NS * hu * wu, ntp = (cnt + 15) >> 4;
  const int t0 = half ? (ntp + 1) >> 1 : 0, ntl = half ? ntp >> 1 : (ntp + 1) >> 1;
  const int th = kq >> 1, tw = kq & 1;
  const int a_frag = m * G::A_ROW + ph * 64 + kq * 16;   // + 256 per channel group, + 16 rows for the second M tile
  int boff[TMAX][4], opix[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const int n = (t0 + t) * 16 + m, nn = min(n, cnt - 1);
    const int si = nn / (hu * wu), rem = nn - si * (hu * wu), u = rem / wu, v = rem - u * wu;
    const int px = si * G::SPLANE + (u + 1 - th) * G::SWP + (v + 1 - tw);
#pragma unroll
    for (int g = 0; g < 4; ++g) boff[t][g] = G::A_BYTES + px * 64 + ((g ^ ((px >> 2) & 3)) * 16);
    opix[t] = (t < ntl && n < cnt) ? si * MT * HW + (2 * u + phh) * G::W + 2 * v + pww : -1;
  }
  float bv[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[mt][i] = bias ? bias[cb0 + mt * 16 + 4 * kq + i] : 0.f;
  f32x4 acc[2][TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) acc[0][t] = acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[G::QA];
  f4u rb[G::QB][2];
  auto issue = [&](int slab) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i)
      ra[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wsh) + a_src[i] + slab * 1024);
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const float* p = small_in + b_src[i] + slab * (32 * P);
      rb[i][0] = *reinterpret_cast<const f4u*>(p);
      rb[i][1] = *reinterpret_cast<const f4u*>(p + P);
    }
  };
  auto commit = [&](unsigned char* st, int slab) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) *reinterpret_cast<u32x4*>(st + a_dst[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int c = slab * 32 + 2 * b_cp[i];
      const float s0 = aff[c], s1 = aff[c + 1], h0 = aff[CS + c], h1 = aff[CS + c + 1];
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<unsigned*>(st + G::A_BYTES + b_dst[i][e]) =
              pack_bf16x2(fmaf(rb[i][0][e], s0, h0), fmaf(rb[i][1][e], s1, h1));
      }
    }
  };

  const int nslab = CS / 32;
  issue(0);
  __syncthreads();
  commit(ldsb, 0);
  if (nslab > 1) issue(1);
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < nslab; ++s) {
    const unsigned char* st = ldsb + (s & 1) * G::STAGE;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(st + a_frag + g * 256);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(st + a_frag + g * 256 + 16 * G::A_ROW);
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        if (t < ntl) {
          const u32x4 b = *reinterpret_cast<const u32x4*>(st + boff[t][g]);
          acc[0][t] = mfma_bf16_k32(a0, b, acc[0][t]);
          acc[1][t] = mfma_bf16_k32(a1, b, acc[1][t]);
        }
      }
      if (g == 1 && s + 1 < nslab) {   // the next slab goes to the other stage under the running matrix pipe
        commit(ldsb + ((s + 1) & 1) * G::STAGE, s + 1);
        if (s + 2 < nslab) issue(s + 2);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias, activation, into the [sample][channel][H*W] output tile (the stages are free)
  const pgv_act_params ap = pgv_act_setup(act, slope);
  float* otile = reinterpret_cast<float*>(ldsb);
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (opix[t] >= 0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          otile[opix[t] + (mt * 16 + 4 * kq + i) * HW] = pgv_act_apply(acc[mt][t][i] + bv[mt][i], ap);
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
int launch_deep_up_bf16(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                        const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                        const pgv_bn_src* bn) {
  using G = UpB<H, W, NS>;
  if (d->Cb % G::MT || d->Cs % 32 || !d->w_shadow) return 0;
  if ((int64_t)d->B * d->Cs * G::P * 4 >= (int64_t)1 << 31 || (int64_t)d->Cs * d->Cb * 32 >= (int64_t)1 << 31) return 0;
  const size_t bytes = 2 * (size_t)G::STAGE + sizeof(float) * (2 * (size_t)d->Cs + 8);
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = deep_up_bf16_kernel<G>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_up_deep_bf16");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_deep_bf16: memset failed");
    return PGV_E_LAUNCH;
  }
  const int groups = (d->B + NS - 1) / NS;
  const u16* up = (const u16*)d->w_shadow + (size_t)d->Cs * d->Cb * 16;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * (d->Cb / G::MT))), dim3(512), bytes, st, d->B, d->Cb, d->Cs, small_in,
                     in_scale, in_shift, up, bias, act, slope, out, stats, groups,
                     (d->flags & PGV_STATS_COPIES) ? 2 * d->Cb : 0, bn ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("conv_up_deep_bf16");
  return 1;
}


struct K1B {
  static constexpr int P = 12, NS = 4, NPX = NS * P, NT = NPX / 16, MT = 128, CK = 64;
  static constexpr int A_ROW = 128 + 32, A_BYTES = MT * A_ROW, B_BYTES = NPX * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int OUT_BYTES = NS * MT * P * 4;
  static_assert(NPX % 16 == 0 && OUT_BYTES <= 2 * STAGE && STAGE % 16 == 0, "tile shapes");
};

__global__ __launch_bounds__(512) void k1_fwd_bf16_kernel(int B, int M, int K, const float* __restrict__ in,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, const u16* __restrict__ wsh,
                                                          const float* __restrict__ bias, int act, float slope,
                                                          float* __restrict__ out, double* __restrict__ stats, int groups,
                                                          int stat_stride, pgv_bn_src in_bn) {
  using G = K1B;
  constexpr int P = G::P, NS = G::NS, NT = G::NT, MT = G::MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  float* aff = reinterpret_cast<float*>(ldsb + 2 * G::STAGE);   // [2*K]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int mb, grp;
  deep_block(M / MT, groups, mb, grp);
  const int m0 = mb * MT, b0 = grp * NS;

  for (int i = tid; i < K; i += 512) {
    float sc = 1.f, sh = 0.f;
    if (in_bn.stats)   // (the producer's BatchNorm finalized here instead of by a launch of its own: pgv_conv_*_bn)
      pgv_bn_finalize_dev(in_bn, K, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[K + i] = sh;
  }
  // ---- loader coordinates: weights 2 x 16 bytes per thread and slab; image: (sample, channel pair, pixel quad)
  int a_src[2], a_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + 512 * i, row = q >> 3, f = q & 7;
    a_src[i] = ((m0 + row) * K) * 2 + f * 16;   // bytes into the shadow (+ 128 per slab)
    a_dst[i] = row * G::A_ROW + f * 16;
  }
  const bool b_ok = tid < NS * 32 * 3;
  const int q = min(tid, NS * 32 * 3 - 1), si = q / 96, rem = q - si * 96, cp = rem / 3, qi = rem - cp * 3;
  const int bs = min(b0 + si, B - 1);
  const int b_src = (bs * K + 2 * cp) * P + 4 * qi;
  int b_dst[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int px = si * P + 4 * qi + e;
    b_dst[e] = G::A_BYTES + px * 128 + (((cp >> 2) ^ (px & 7)) * 16) + (cp & 3) * 4;
  }
  // ---- fragment coordinates
  const int a_frag = (wave * 16 + m) * G::A_ROW + kq * 16;   // + 64 for the second step of a slab
  int b_frag[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int px = t * 16 + m;
      b_frag[t][st] = G::A_BYTES + px * 128 + (((st * 4 + kq) ^ (px & 7)) * 16);
    }
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[m0 + wave * 16 + 4 * kq + i] : 0.f;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[2];
  f4u rb[2];
  auto issue = [&](int slab) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      ra[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wsh) + a_src[i] + slab * 128);
    const float* p = in + b_src + slab * (64 * P);
    rb[0] = *reinterpret_cast<const f4u*>(p);
    rb[1] = *reinterpret_cast<const f4u*>(p + P);
  };
  auto commit = [&](unsigned char* st, int slab) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(st + a_dst[i]) = ra[i];
    const int c = slab * 64 + 2 * cp;
    const float s0 = aff[c], s1 = aff[c + 1], h0 = aff[K + c], h1 = aff[K + c + 1];
    if (b_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<unsigned*>(st + b_dst[e]) = pack_bf16x2(fmaf(rb[0][e], s0, h0), fmaf(rb[1][e], s1, h1));
    }
  };
  const int nslab = K / 64;
  issue(0);
  __syncthreads();   // affine staged
  commit(ldsb, 0);
  if (nslab > 1) issue(1);
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < nslab; ++s) {
    const unsigned char* st = ldsb + (s & 1) * G::STAGE;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(st + a_frag + k2 * 64);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16_k32(a, *reinterpret_cast<const u32x4*>(st + b_frag[t][k2]), acc[t]);
      if (k2 == 0 && s + 1 < nslab) {
        commit(ldsb + ((s + 1) & 1) * G::STAGE, s + 1);
        if (s + 2 < nslab) issue(s + 2);
      }
    }
    __syncthreads();
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
int launch_k1_fwd_bf16(const pgv_conv_desc* d, bool up, const float* in, const float* in_scale, const float* in_shift,
                       const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                       const pgv_bn_src* bn) {
  using G = K1B;
  const int M = up ? d->Cb : d->Cs, K = up ? d->Cs : d->Cb;
  if ((int64_t)d->B * K * G::P * 4 >= (int64_t)1 << 31 || (int64_t)M * K * 2 >= (int64_t)1 << 31) return 0;
  const size_t bytes = 2 * (size_t)G::STAGE + sizeof(float) * (2 * (size_t)K + 8);
  if (bytes > (size_t)kMaxLds) return 0;
  static bool attr_done = false;
  int rc = raise_lds_limit(k1_fwd_bf16_kernel, &attr_done, "conv_k1_bf16");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * M, st) != hipSuccess) {
    pgv_set_error("conv_k1_bf16: memset failed");
    return PGV_E_LAUNCH;
  }
  const int groups = (d->B + G::NS - 1) / G::NS;
  const u16* wsh = (const u16*)d->w_shadow + (up ? (size_t)d->Cs * d->Cb : 0);
  hipLaunchKernelGGL(k1_fwd_bf16_kernel, dim3((unsigned)(groups * (M / G::MT))), dim3(512), bytes, st, d->B, M, K, in, in_scale,
                     in_shift, wsh, bias, act, slope, out, stats, groups, (d->flags & PGV_STATS_COPIES) ? 2 * M : 0,
                     bn ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("conv_k1_bf16");
  return 1;
}

}  // namespace

bool pgv_k1_bf16_shape(const pgv_conv_desc* d) {
  return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->Hb == 3 && d->Wb == 4 && d->Cb % 128 == 0 &&
         d->Cs % 128 == 0;
}

bool pgv_deep_bf16_shape(const pgv_conv_desc* d) {
  return d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 2 && d->Cb >= 64 && d->Cb % 16 == 0 && d->Cs % 64 == 0 &&
         ((d->Hb == 17 && d->Wb == 23) || (d->Hb == 9 && d->Wb == 12) || (d->Hb == 5 && d->Wb == 7));
}

int pgv_conv_down_deep_bf16(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                            const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                            const pgv_bn_src* bn) {
  if ((d->flags & PGV_COMPUTE_BF16) && d->w_shadow && pgv_k1_bf16_shape(d))
    return launch_k1_fwd_bf16(d, false, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (!(d->flags & PGV_COMPUTE_BF16) || !d->w_shadow || !pgv_deep_bf16_shape(d)) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_down_bf16<17, 23, 1, true>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_down_bf16<9, 12, 4, false>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_down_bf16<5, 7, 8, true>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  return 0;
}

int pgv_conv_up_deep_bf16(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                          const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                          const pgv_bn_src* bn) {
  if ((d->flags & PGV_COMPUTE_BF16) && d->w_shadow && pgv_k1_bf16_shape(d))
    return launch_k1_fwd_bf16(d, true, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (!(d->flags & PGV_COMPUTE_BF16) || !d->w_shadow || !pgv_deep_bf16_shape(d)) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_up_bf16<17, 23, 2>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_up_bf16<9, 12, 4>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_up_bf16<5, 7, 8>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  return 0;
}

int pgv_conv_up_big_bf16(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                         const float* bias, int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse,
                         hipStream_t st, const pgv_bn_src* bn) {
  return pgv_conv_up_big_split(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
}
int pgv_conv_down_big_bf16(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                           const float* bias, int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse,
                           hipStream_t st, const pgv_bn_src* bn) {
  return pgv_conv_down_big_split(d, big, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
}

// the shadows of n <= 8 layers in one launch (every descriptor must have a shadow: pgv_conv_weight_shadow_bytes > 0)
