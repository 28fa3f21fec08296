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
#include "conv_tile.h"
#include "conv_deep_common.h"

static unsigned long long* g_deep_bf16_stamps = nullptr;   // device buffer of the timing scripts: clock64() at the phase marks
extern "C" void pgv_dbg_set_deep_bf16_stamps(void* p) { g_deep_bf16_stamps = (unsigned long long*)p; }
#define BSTAMP(k)                                                                                   \
  do {                                                                                              \
    if (stamps && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 77)) stamps[(blockIdx.x ? 16 : 0) + (k)] = clock64(); \
  } while (0)
// A/B knobs of the timing scripts (scratch/time_deep_bf16.py, pgv_dbg_set_deep_bf16_variant; no effect on results beyond the
// summation order): 8 = deep / 1x1 weight gradients stay on the fp32-image kernels, 16 / 32 = no up_big / down_big kernels,
// 64 = 9x12 weight gradient on 16-sample blocks, 128 = 17x23 weight gradient on 8-sample blocks, 256 = 129x174 transposed
// convolution on up_big
static int g_deep_bf16_dbg = 0;
extern "C" int pgv_dbg_set_deep_bf16_variant(int v) {
  const int old = g_deep_bf16_dbg;
  g_deep_bf16_dbg = v;
  return old;
}

namespace {

typedef unsigned short u16;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight shadows.  down: D[cs][cb/8][kh*4+kw][8] (M = cs);  up: U[cb][cs/8][phase][th*2+tw][8] (M = cb), phase = 2ph+pw,
// taps kh = ph + 2th, kw = pw + 2tw.  One thread = (cs, channel group of 8 cb, kernel row): 8 x 16-byte reads, 4 x 16-byte
// writes of the down shadow; the up shadow is written by the thread that owns (cb, group of 8 cs, kernel row).
__device__ __forceinline__ void shadow_k4_item(int it, const float* __restrict__ w, int CS, int CB, u16* __restrict__ down,
                                               u16* __restrict__ up) {
  {
    const int kh = it & 3, g = (it >> 2) % (CB / 8), cs = (it >> 2) / (CB / 8);
    f32x4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const f32x4*>(w + ((size_t)(cs * CB + g * 8 + c) * 16 + kh * 4));
    u32x4* dst = reinterpret_cast<u32x4*>(down + ((size_t)(cs * (CB / 8) + g) * 16 + kh * 4) * 8);
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
      dst[kw] = u32x4{pack_bf16x2(v[0][kw], v[1][kw]), pack_bf16x2(v[2][kw], v[3][kw]), pack_bf16x2(v[4][kw], v[5][kw]),
                      pack_bf16x2(v[6][kw], v[7][kw])};
  }
  if (!up) return;
  {
    const int kh = it & 3, cb = (it >> 2) % CB, g = (it >> 2) / CB;   // cb fastest: the reads of a wave are 64-byte pieces
    f32x4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const f32x4*>(w + ((size_t)((g * 8 + c) * CB + cb) * 16 + kh * 4));
    const int ph = kh & 1, th = kh >> 1;
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
      const int pw = kw & 1, tw = kw >> 1;
      u32x4* dst = reinterpret_cast<u32x4*>(up + ((size_t)((cb * (CS / 8) + g) * 4 + 2 * ph + pw) * 4 + 2 * th + tw) * 8);
      *dst = u32x4{pack_bf16x2(v[0][kw], v[1][kw]), pack_bf16x2(v[2][kw], v[3][kw]), pack_bf16x2(v[4][kw], v[5][kw]),
                   pack_bf16x2(v[6][kw], v[7][kw])};
    }
  }
}
__global__ __launch_bounds__(256) void deep_shadow_kernel(const float* __restrict__ w, int CS, int CB,
                                                        u16* __restrict__ down, u16* __restrict__ up) {
  const int items = CS * (CB / 8) * 4;   // (= CB * (CS / 8) * 4: the same item count serves both layouts)
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) shadow_k4_item(it, w, CS, CB, down, up);
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


// ---------------------------------------------------------------------------------------------------------------
// WGRAD: gw[cs,cb,kh,kw] = sum_{b,oh,ow} s'[b,cs,oh,ow] * x'[b,cb,2oh-2+kh,2ow-2+kw]
// GEMM: M = cs (64 per workgroup), N = (cb, 16 taps): one 16-column tile per big channel, 8 big channels per workgroup,
// K = (sample, output pixel).  The contraction index inside a fragment is the SAMPLE: both images sit in LDS
// sample-innermost - pixel -> channel -> 16 samples of a block = two 16-byte halves - so the K = 32 of one instruction is
// 16 samples x 2 consecutive output pixels (lane group kq = 2 * pixel + half), the A fragment the half-row
// S[pixel][cs][half] and the B fragment of column (cb, tap) the half-row X[input pixel of (pixel, tap)][cb][half]: the
// im2col gather is the fragment ADDRESS, every read 16 aligned bytes.  A unit of work = (block of 16 samples, band of R
// output rows); a workgroup sweeps a range of units (double-buffered stages, register prefetch) with its 64 x 128
// accumulator tile in registers and stores it once: to gw, or - when the units of a tile are split over several
// workgroups (the layers with few tiles) - to a partial gradient that deep_wgrad_reduce_kernel adds up.
template <int H_, int W_, int R_, int WP_>
struct WgradB {
  static constexpr int H = H_, W = W_, R = R_, WP = WP_;
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, BANDS = Hs / R;
  static_assert(Hs % R == 0, "bands of whole output rows");
  static constexpr int SPX = R * Ws, SPX2 = (SPX + 1) / 2 * 2, STEPS = SPX2 / 2;   // pixels of a band, padded to pairs
  static constexpr int XR = 2 * R + 2;                       // input rows of a band
  static constexpr int XROWS = XR + (SPX2 > SPX ? 2 : 0);    // + the rows the pad pixel's fragment addresses touch (zeros)
  static_assert(WP >= 2 * Ws + 2 && (WP % 16 == 4 || WP % 16 == 12), "row stride: the 16 taps of a pixel on 16 distinct slots");
  static constexpr int S_BYTES = SPX2 * 64 * 32;             // [pixel][64 cs][16 samples] bf16
  static constexpr int X_BYTES = XROWS * WP * 256;           // [input pixel][8 cb][16 samples] bf16
  static constexpr int STAGE = S_BYTES + X_BYTES;
  static constexpr int QS = (SPX + 3) / 4, S_HALF = 64 * QS;       // items of one half of the block: (cs, quad)
  static constexpr int QX = (W + 3) / 4, X_HALF = 8 * XR * QX;     // (cb, row, quad)
  static_assert(SPX >= 4 && W >= 4, "shifted last quads");
};

template <class G>
__global__ __launch_bounds__(512) void deep_wgrad_bf16_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                              const float* __restrict__ big_scale,
                                                              const float* __restrict__ big_shift,
                                                              const float* __restrict__ small_in,
                                                              const float* __restrict__ small_scale,
                                                              const float* __restrict__ small_shift,
                                                              float* __restrict__ outp, int nsplit, int add,
                                                              unsigned long long* __restrict__ stamps) {
  constexpr int H = G::H, W = G::W, Hs = G::Hs, Ws = G::Ws, WP = G::WP, STEPS = G::STEPS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), mh = wave & 1, nq = wave >> 1;
  // (cs block, split) share the small operand, the CB/8 workgroups of one such combination sit on one XCD
  const int NB = CB / 8, ncombo = (CS / 64) * nsplit;
  int combo, nb;
  if (ncombo % 8 == 0) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    combo = x + 8 * (q / NB);
    nb = q - (q / NB) * NB;
  } else {
    combo = blockIdx.x / NB;
    nb = blockIdx.x - combo * NB;
  }
  const int mb = combo / nsplit, ks = combo - mb * nsplit;
  const int cs0 = mb * 64, cb0 = nb * 8;
  const int units = ((B + 15) >> 4) * G::BANDS, per = (units + nsplit - 1) / nsplit;
  const int u0 = ks * per, u1 = min(units, u0 + per);
  BSTAMP(0);

  // ---- loader coordinates.  An item = the 8 samples of one HALF of the block x one channel x four pixels: 8 loads (one per
  // sample; across the lanes of an instruction the addresses run over channels and quads of ONE sample - coalesced) and
  // 4 ds_write_b128 (one per pixel: the 8 samples of a half are the 16 contiguous bytes a fragment reads).  The half is
  // wave-uniform (threads 0-255 / 256-511), so the sample bases are scalars and a load is base + per-item offset.
  static_assert(G::S_HALF <= 256 && G::X_HALF <= 256, "one item of each operand per thread");
  const int hf = __builtin_amdgcn_readfirstlane(tid >> 8), it = tid & 255;
  const bool s_ok = it < G::S_HALF, x_ok = it < G::X_HALF;
  int s_off, s_dst, x_off, x_row, x_dst[4];
  float s_sc, s_sh, x_sc, x_sh;
  {
    const int q = min(it, G::S_HALF - 1), cs = q / G::QS, qi = q - cs * G::QS, p0 = min(4 * qi, G::SPX - 4);
    s_off = ((cs0 + cs) * (Hs * Ws) + p0) * 4;   // bytes; + sample * CS * P + band * SPX
    s_sc = small_scale ? small_scale[cs0 + cs] : 1.f;
    s_sh = small_scale ? small_shift[cs0 + cs] : 0.f;
    s_dst = p0 * 2048 + cs * 32 + hf * 16;       // + 2048 per pixel
  }
  {
    const int q = min(it, G::X_HALF - 1), cb = q / (G::XR * G::QX), rem = q - cb * (G::XR * G::QX);
    const int r = rem / G::QX, qi = rem - r * G::QX, c0 = min(4 * qi, W - 4);
    x_row = r;
    x_off = ((cb0 + cb) * (H * W) + c0) * 4;     // bytes; + sample * CB * H * W + image row * W
    x_sc = big_scale ? big_scale[cb0 + cb] : 1.f;
    x_sh = big_scale ? big_shift[cb0 + cb] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int px = r * WP + c0 + e + 2;
      x_dst[e] = G::S_BYTES + px * 256 + ((((cb * 2 + hf) ^ px) & 15) * 16);
    }
  }
  const bool s_aff = small_scale != nullptr, x_aff = big_scale != nullptr;

  // Loads run TWO units ahead of the matrix loop in two register sets (the units are short - 4 to 6 instructions of K per
  // wave - against ~2 us of L2 latency under load): inline asm with manual s_waitcnt, because the compiler's counter model
  // merges the in-flight sets at the loop header and would wait for both at every commit (conv_deep.hip does the same).
  struct RegSet {
    f4u rs[8], rx[8];
    float x_m;        // 0: the X item's values are zeros this unit (row outside the image)
    unsigned live;    // bit j: sample j of the half exists - all ones except in a partial last block
  };
  RegSet r0, r1;
  auto issue = [&](int u, RegSet& r) {
    const int sb = u / G::BANDS, band = u - sb * G::BANDS;
    const int b = sb * 16 + 8 * hf;
    r.live = (1u << min(max(B - b, 0), 8)) - 1u;
    const int ih = 2 * band * G::R - 2 + x_row;
    const bool in = (unsigned)ih < (unsigned)H;
    r.x_m = in ? 1.f : 0.f;
    const int so = s_off + band * (G::SPX * 4), xo = x_off + (in ? ih : 0) * (W * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int bj = min(b + j, B - 1);   // (uniform: scalar bases)
      const unsigned char* ps = reinterpret_cast<const unsigned char*>(small_in) + (size_t)bj * CS * (Hs * Ws) * 4;
      const unsigned char* px = reinterpret_cast<const unsigned char*>(big) + (size_t)bj * CB * (H * W) * 4;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.rs[j]) : "v"(so), "s"(ps) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.rx[j]) : "v"(xo), "s"(px) : "memory");
    }
  };
  // `younger`: the other set has been requested after this one and may stay in flight
  auto wait_set = [&](RegSet& r, bool younger) {
    if (younger)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("" : "+v"(r.rs[j]));   // the values exist from here on
      asm volatile("" : "+v"(r.rx[j]));
    }
  };
  auto pack8 = [&](const f4u (&r)[8], int e, bool aff, float sc, float sh, float msk, unsigned live) -> u32x4 {
    float v[8];
    if (live != 0xFFu) {      // partial last block (uniform): per-sample masks
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mj = ((live >> j) & 1u) ? msk : 0.f;
        v[j] = fmaf(r[j][e], sc * mj, sh * mj);
      }
    } else if (aff) {         // (uniform)
      const float a = sc * msk, c = sh * msk;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(r[j][e], a, c);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = r[j][e] * msk;
    }
    return u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  };
  auto commit = [&](unsigned char* st, const RegSet& r) {
    if (s_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<u32x4*>(st + s_dst + e * 2048) = pack8(r.rs, e, s_aff, s_sc, s_sh, 1.f, r.live);
    }
    if (x_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x4*>(st + x_dst[e]) = pack8(r.rx, e, x_aff, x_sc, x_sh, r.x_m, r.live);
    }
  };
  if (u0 < u1) issue(u0, r0);   // in flight while the stages are cleared
  if (u0 + 1 < u1) issue(u0 + 1, r1);

  for (int i = tid; i < 2 * G::STAGE / 16; i += 512) reinterpret_cast<u32x4*>(ldsb)[i] = u32x4{0, 0, 0, 0};

  // ---- fragment coordinates (bytes into a stage, identical for every unit).  A: row cs = (2 mh + t) * 16 + m, half
  // sg = kq & 1, pixel 2 step + (kq >> 1); rows 32 bytes apart, no swizzle: in a 16-lane group of ds_read_b128 rows m and
  // m + 8 always come with opposite halves (lanes {0-3, 12-15} of one kq, lanes {4-11} of the next), so its 16 fragments
  // fall on 16 distinct slots.  B: column tap n = m = (kh, kw) of big channel 2 nq + t at input pixel
  // (2 ohl + kh) * WP + 2 ow + kw, its half of channel cb stored at slot (2 cb + sg) ^ (pixel & 15).
  const int sg = kq & 1, pp = kq >> 1;
  const int a_frag = pp * 2048 + ((2 * mh) * 16 + m) * 32 + sg * 16;   // second M tile: + 512; step: + 4096
  int b_frag[STEPS][2];
#pragma unroll
  for (int sp = 0; sp < STEPS; ++sp) {
    const int p = 2 * sp + pp;
    const int px = (m >> 2) * WP + (m & 3) + 2 * (p / Ws) * WP + 2 * (p % Ws);
#pragma unroll
    for (int t = 0; t < 2; ++t) b_frag[sp][t] = G::S_BYTES + px * 256 + (((((2 * nq + t) * 2 + sg) ^ px) & 15) * 16);
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  BSTAMP(1);
  __syncthreads();   // stages zeroed
  BSTAMP(2);
  if (u0 < u1) {
    wait_set(r0, u0 + 1 < u1);
    commit(ldsb, r0);
  }
  if (u0 + 2 < u1) issue(u0 + 2, r0);
  __syncthreads();
  BSTAMP(3);
  // one unit: matrix loop over stage (u - u0) & 1; half way, unit u + 1 (in `rn`) goes to the other stage and unit u + 3
  // is requested into the registers it frees
  auto unit_step = [&](int u, RegSet& rn) {
    const unsigned char* st = ldsb + ((u - u0) & 1) * G::STAGE;
#pragma unroll
    for (int sp = 0; sp < STEPS; ++sp) {
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(st + a_frag + sp * 4096);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(st + a_frag + sp * 4096 + 512);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u32x4 b = *reinterpret_cast<const u32x4*>(st + b_frag[sp][t]);
        acc[0][t] = mfma_bf16_k32(a0, b, acc[0][t]);
        acc[1][t] = mfma_bf16_k32(a1, b, acc[1][t]);
      }
      if (sp == STEPS / 2 && u + 1 < u1) {
        wait_set(rn, u + 2 < u1);
        commit(ldsb + ((u + 1 - u0) & 1) * G::STAGE, rn);
        if (u + 3 < u1) issue(u + 3, rn);
      }
    }
    __syncthreads();
  };
#pragma unroll 1
  for (int u = u0; u < u1; u += 2) {
    unit_step(u, r1);
    if (u + 1 < u1) unit_step(u + 1, r0);
  }
  BSTAMP(4);
  // ---- store: D row 4 kq + i of M tile (2 mh + t), column tap m of big channel cb0 + 2 nq + t2
  float* o = outp + (nsplit > 1 ? (size_t)ks * CS * CB * 16 : 0);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t idx = ((size_t)(cs0 + (2 * mh + t) * 16 + 4 * kq + i) * CB + cb0 + 2 * nq + t2) * 16 + m;
        o[idx] = acc[t][t2][i] + ((add && nsplit == 1) ? o[idx] : 0.f);
      }
  BSTAMP(5);
}

// The same product on blocks of 8 samples: images [pixel][channel][8 samples] (16 bytes), the K = 32 of one instruction =
// 8 samples x 4 consecutive output pixels (lane group kq = pixel).  Half the LDS per pixel lets a unit be a whole 9x12 plane
// (35 pixels) or three rows of a 17x23 plane: 9 instructions of K per wave and unit instead of 4 - 6, 30 % fewer bytes per
// workgroup (no rows fetched twice at 9x12) and fewer, longer pipeline steps.
// NP = 3 (PGV_COMPUTE_F32_SPLIT): three plane images of each operand (x = x1 + x2 + x3 exactly, split at the commit) and six
// instructions per fragment pair, smallest terms first - the fp32 product on the bf16 matrix pipe.
template <int H_, int W_, int R_, int WP_, int NP_ = 1>
struct Wgrad8 {
  static constexpr int H = H_, W = W_, R = R_, WP = WP_, NP = NP_;
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, BANDS = Hs / R;
  static_assert(Hs % R == 0, "bands of whole output rows");
  static constexpr int SPX = R * Ws, SPX4 = (SPX + 3) / 4 * 4, STEPS = SPX4 / 4;   // pixels of a band, padded to quads
  static constexpr int XR = 2 * R + 2;
  static constexpr int XROWS = XR + (SPX4 > SPX ? 2 : 0);    // + the rows the pad pixels' fragment addresses touch (zeros)
  static_assert(WP >= 2 * Ws + 2 && (WP % 16 == 4 || WP % 16 == 12), "row stride: the 16 taps of a pixel on 16 distinct slots");
  static constexpr int S_BYTES = SPX4 * 64 * 16;             // [pixel][64 cs][8 samples] bf16
  static constexpr int X_BYTES = XROWS * WP * 128;           // [input pixel][8 cb][8 samples] bf16
  static constexpr int PLANE_BYTES = S_BYTES + X_BYTES, STAGE = NP * PLANE_BYTES;
  static constexpr int QS = (SPX + 3) / 4, S_ITEMS = 64 * QS, QA = (S_ITEMS + 511) / 512;   // (cs, quad)
  static constexpr int QX = (W + 3) / 4, X_ITEMS = 8 * XR * QX;                             // (cb, row, quad)
  static constexpr int NLOADS = 8 * (QA + 1);                // per thread and unit
  // first thread of the X items: behind the S items' waves when both fit the workgroup (the commit's conversions - three
  // planes: ~250 VALU instructions per item - then spread over 5 - 6 waves instead of piling up on waves 0 - 2)
  static constexpr int XT0 = (NP == 3 && QA == 1 && (S_ITEMS + 63) / 64 * 64 + X_ITEMS <= 512) ? (S_ITEMS + 63) / 64 * 64 : 0;
  static_assert(SPX >= 4 && W >= 4 && X_ITEMS <= 512 && NLOADS <= 63 && 2 * STAGE <= 160 * 1024, "tile shapes");
};

template <class G>
__global__ __launch_bounds__(512) void deep_wgrad8_bf16_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                               const float* __restrict__ big_scale,
                                                               const float* __restrict__ big_shift,
                                                               const float* __restrict__ small_in,
                                                               const float* __restrict__ small_scale,
                                                               const float* __restrict__ small_shift,
                                                               float* __restrict__ outp, int nsplit, int add) {
  constexpr int H = G::H, W = G::W, Hs = G::Hs, Ws = G::Ws, WP = G::WP, STEPS = G::STEPS;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const pgv_split_sel sel = pgv_split_sel_make();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), mh = wave & 1, nq = wave >> 1;
  const int NB = CB / 8, ncombo = (CS / 64) * nsplit;
  int combo, nb;
  if (ncombo % 8 == 0) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    combo = x + 8 * (q / NB);
    nb = q - (q / NB) * NB;
  } else {
    combo = blockIdx.x / NB;
    nb = blockIdx.x - combo * NB;
  }
  const int mb = combo / nsplit, ks = combo - mb * nsplit;
  const int cs0 = mb * 64, cb0 = nb * 8;
  const int units = ((B + 7) >> 3) * G::BANDS, per = (units + nsplit - 1) / nsplit;
  const int u0 = ks * per, u1 = min(units, u0 + per);

  // ---- loader items: the 8 samples of the block x one channel x four pixels -> 8 loads, 4 ds_write_b128
  int s_off[G::QA], s_dst[G::QA];
  float s_sc[G::QA], s_sh[G::QA];
  bool s_ok[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = min(tid + 512 * i, G::S_ITEMS - 1), cs = q / G::QS, qi = q - cs * G::QS, p0 = min(4 * qi, G::SPX - 4);
    s_ok[i] = tid + 512 * i < G::S_ITEMS;
    s_off[i] = ((cs0 + cs) * (Hs * Ws) + p0) * 4;   // bytes; + sample * CS * P + band * SPX
    s_sc[i] = small_scale ? small_scale[cs0 + cs] : 1.f;
    s_sh[i] = small_scale ? small_shift[cs0 + cs] : 0.f;
    s_dst[i] = p0 * 1024 + cs * 16;                 // + 1024 per pixel
  }
  const bool x_ok = (unsigned)(tid - G::XT0) < (unsigned)G::X_ITEMS;
  int x_off, x_row, x_dst[4];
  float x_sc, x_sh;
  {
    const int q = min(max(tid - G::XT0, 0), G::X_ITEMS - 1), cb = q / (G::XR * G::QX), rem = q - cb * (G::XR * G::QX);
    const int r = rem / G::QX, qi = rem - r * G::QX, c0 = min(4 * qi, W - 4);
    x_row = r;
    x_off = ((cb0 + cb) * (H * W) + c0) * 4;
    x_sc = big_scale ? big_scale[cb0 + cb] : 1.f;
    x_sh = big_scale ? big_shift[cb0 + cb] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int px = r * WP + c0 + e + 2;
      x_dst[e] = G::S_BYTES + px * 128 + ((cb ^ ((px >> 1) & 7)) * 16);
    }
  }
  const bool s_aff = small_scale != nullptr, x_aff = big_scale != nullptr;

  struct RegSet {
    f4u rs[G::QA][8], rx[8];
    float x_m;
    unsigned live;
  };
  RegSet r0;   // (one set, one unit ahead: two sets of 24 loads do not fit the register file)
  auto issue = [&](int u, RegSet& r) {
    const int sb = u / G::BANDS, band = u - sb * G::BANDS;
    const int b = sb * 8;
    r.live = (1u << min(max(B - b, 0), 8)) - 1u;
    const int ih = 2 * band * G::R - 2 + x_row;
    const bool in = (unsigned)ih < (unsigned)H;
    r.x_m = in ? 1.f : 0.f;
    const int xo = x_off + (in ? ih : 0) * (W * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int bj = min(b + j, B - 1);   // (uniform: scalar bases)
      const unsigned char* ps = reinterpret_cast<const unsigned char*>(small_in) + (size_t)bj * CS * (Hs * Ws) * 4;
      const unsigned char* px = reinterpret_cast<const unsigned char*>(big) + (size_t)bj * CB * (H * W) * 4;
#pragma unroll
      for (int i = 0; i < G::QA; ++i) {
        const int so = s_off[i] + band * (G::SPX * 4);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.rs[i][j]) : "v"(so), "s"(ps) : "memory");
      }
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.rx[j]) : "v"(xo), "s"(px) : "memory");
    }
  };
  auto wait_set = [&](RegSet& r) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < G::QA; ++i) asm volatile("" : "+v"(r.rs[i][j]));
      asm volatile("" : "+v"(r.rx[j]));
    }
  };
  auto pack8 = [&](const f4u (&r)[8], int e, bool aff, float sc, float sh, float msk, unsigned live) -> u32x4 {
    float v[8];
    if (live != 0xFFu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mj = ((live >> j) & 1u) ? msk : 0.f;
        v[j] = fmaf(r[j][e], sc * mj, sh * mj);
      }
    } else if (aff) {
      const float a = sc * msk, c = sh * msk;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(r[j][e], a, c);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = r[j][e] * msk;
    }
    return u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  };
  // NP = 3: the 8 values as three planes (hi, mid, lo)
  auto store8 = [&](unsigned char* dst, const f4u (&r)[8], int e, bool aff, float sc, float sh, float msk, unsigned live) {
    if constexpr (G::NP == 1) {
      *reinterpret_cast<u32x4*>(dst) = pack8(r, e, aff, sc, sh, msk, live);
    } else {
      float v[8];
      if (live != 0xFFu) {      // partial last block (uniform): per-sample masks
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float mj = ((live >> j) & 1u) ? msk : 0.f;
          v[j] = fmaf(r[j][e], sc * mj, sh * mj);
        }
      } else if (aff) {         // (uniform)
        const float a = sc * msk, c = sh * msk;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(r[j][e], a, c);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = r[j][e] * msk;
      }
      u32x4 ph, pm, pl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned a, b, c;
        pgv_split3_pair(v[2 * j], v[2 * j + 1], a, b, c, sel);
        ph[j] = a, pm[j] = b, pl[j] = c;
      }
      *reinterpret_cast<u32x4*>(dst) = ph;
      *reinterpret_cast<u32x4*>(dst + G::PLANE_BYTES) = pm;
      *reinterpret_cast<u32x4*>(dst + 2 * G::PLANE_BYTES) = pl;
    }
  };
  auto commit = [&](unsigned char* st, const RegSet& r) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) {
      if (s_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) store8(st + s_dst[i] + e * 1024, r.rs[i], e, s_aff, s_sc[i], s_sh[i], 1.f, r.live);
      }
    }
    if (x_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) store8(st + x_dst[e], r.rx, e, x_aff, x_sc, x_sh, r.x_m, r.live);
    }
  };
  if (u0 < u1) issue(u0, r0);

  for (int i = tid; i < 2 * G::STAGE / 16; i += 512) reinterpret_cast<u32x4*>(ldsb)[i] = u32x4{0, 0, 0, 0};

  // ---- fragments: A rows cs = (2 mh + t) * 16 + m at pixel 4 step + kq; B column tap m = (kh, kw) of big channel 2 nq + t at
  // the input pixel of (pixel, tap); the 16-byte entry of channel cb sits at cb ^ ((input pixel >> 1) & 7)
  const int a_frag = kq * 1024 + ((2 * mh) * 16 + m) * 16;   // second M tile: + 256; step: + 4096
  int b_frag[STEPS][2];
#pragma unroll
  for (int sp = 0; sp < STEPS; ++sp) {
    const int p = 4 * sp + kq;
    const int px = (m >> 2) * WP + (m & 3) + 2 * (p / Ws) * WP + 2 * (p % Ws);
#pragma unroll
    for (int t = 0; t < 2; ++t) b_frag[sp][t] = G::S_BYTES + px * 128 + (((2 * nq + t) ^ ((px >> 1) & 7)) * 16);
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();   // stages zeroed
  if (u0 < u1) {
    wait_set(r0);
    commit(ldsb, r0);
  }
  if (u0 + 1 < u1) issue(u0 + 1, r0);
  __syncthreads();
  auto unit_step = [&](int u, RegSet& rn) {
    const unsigned char* st = ldsb + ((u - u0) & 1) * G::STAGE;
#pragma unroll
    for (int sp = 0; sp < STEPS; ++sp) {
      if constexpr (G::NP == 1) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(st + a_frag + sp * 4096);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(st + a_frag + sp * 4096 + 256);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const u32x4 b = *reinterpret_cast<const u32x4*>(st + b_frag[sp][t]);
          acc[0][t] = mfma_bf16_k32(a0, b, acc[0][t]);
          acc[1][t] = mfma_bf16_k32(a1, b, acc[1][t]);
        }
      } else {
        u32x4 a[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[0][p] = *reinterpret_cast<const u32x4*>(st + p * G::PLANE_BYTES + a_frag + sp * 4096);
          a[1][p] = *reinterpret_cast<const u32x4*>(st + p * G::PLANE_BYTES + a_frag + sp * 4096 + 256);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          u32x4 b[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const u32x4*>(st + p * G::PLANE_BYTES + b_frag[sp][t]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {   // the six products, smallest first
            f32x4 c = acc[mt][t];
            c = mfma_bf16_k32(a[mt][0], b[2], c);
            c = mfma_bf16_k32(a[mt][2], b[0], c);
            c = mfma_bf16_k32(a[mt][1], b[1], c);
            c = mfma_bf16_k32(a[mt][0], b[1], c);
            c = mfma_bf16_k32(a[mt][1], b[0], c);
            acc[mt][t] = mfma_bf16_k32(a[mt][0], b[0], c);
          }
        }
      }
      if (sp == STEPS / 2 && u + 1 < u1) {
        wait_set(rn);
        commit(ldsb + ((u + 1 - u0) & 1) * G::STAGE, rn);
        if (u + 2 < u1) issue(u + 2, rn);
      }
    }
    // (LDS traffic complete + barrier, WITHOUT the vector-memory wait of __syncthreads(): the loads of unit u + 2, issued
    // half a unit ago, stay in flight until wait_set() of the next unit asks for them - with the wait here every unit
    // exposed their latency)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
#pragma unroll 1
  for (int u = u0; u < u1; ++u) unit_step(u, r0);
  float* o = outp + (nsplit > 1 ? (size_t)ks * CS * CB * 16 : 0);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t idx = ((size_t)(cs0 + (2 * mh + t) * 16 + 4 * kq + i) * CB + cb0 + 2 * nq + t2) * 16 + m;
        o[idx] = acc[t][t2][i] + ((add && nsplit == 1) ? o[idx] : 0.f);
      }
}

// gw = (add ? gw : 0) + sum of the partial gradients (fixed order: deterministic)
__global__ __launch_bounds__(256) void deep_wgrad_reduce_kernel(const f32x4* __restrict__ partial, int nparts, int n4,
                                                                f32x4* __restrict__ gw, int add) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = add ? gw[i] : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nparts; ++s) v += partial[(size_t)s * n4 + i];
  gw[i] = v;
}

// 1x1 layers on 3x4 planes: gw[cs][cb] = sum_{b,p} s'[b,cs,p] * x'[b,cb,p].  M = cs, N = cb, 128 x 128 per workgroup, the
// contraction index inside a fragment is the sample again: images [pixel][channel][8 samples] (16 bytes), the K = 32 of one
// instruction = 8 samples x 4 pixels (lane group kq = pixel); a unit = a block of 8 samples.
struct K1W {
  static constexpr int P = 12, T = 128, IMG = P * T * 16, STAGE = 2 * IMG, ITEMS = T * 3;   // items: (channel, pixel quad)
};

__global__ __launch_bounds__(512) void k1_wgrad_bf16_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                            const float* __restrict__ big_scale,
                                                            const float* __restrict__ big_shift,
                                                            const float* __restrict__ small_in,
                                                            const float* __restrict__ small_scale,
                                                            const float* __restrict__ small_shift,
                                                            float* __restrict__ outp, int nsplit, int add) {
  using G = K1W;
  constexpr int P = G::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), mh = wave & 1, nq = wave >> 1;
  const int NBc = CB / G::T, tiles = (CS / G::T) * NBc;
  const int tile = blockIdx.x % tiles, ks = blockIdx.x / tiles;
  const int cs0 = (tile / NBc) * G::T, cb0 = (tile % NBc) * G::T;
  const int units = (B + 7) >> 3, per = (units + nsplit - 1) / nsplit;
  const int u0 = ks * per, u1 = min(units, u0 + per);

  const bool ok = tid < G::ITEMS;
  const int q = min(tid, G::ITEMS - 1), ch = q / 3, qi = q - ch * 3;
  const int s_off = ((cs0 + ch) * P + 4 * qi) * 4, x_off = ((cb0 + ch) * P + 4 * qi) * 4;   // bytes; + sample * C * P
  const int dst = (4 * qi) * (G::T * 16) + ch * 16;                                        // + T * 16 per pixel
  const float s_sc = small_scale ? small_scale[cs0 + ch] : 1.f, s_sh = small_scale ? small_shift[cs0 + ch] : 0.f;
  const float x_sc = big_scale ? big_scale[cb0 + ch] : 1.f, x_sh = big_scale ? big_shift[cb0 + ch] : 0.f;
  const bool s_aff = small_scale != nullptr, x_aff = big_scale != nullptr;

  f4u rs[8], rx[8];
  unsigned live;
  auto issue = [&](int u) {
    const int b = u * 8;
    live = (1u << min(max(B - b, 0), 8)) - 1u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int bj = min(b + j, B - 1);
      rs[j] = *reinterpret_cast<const f4u*>(reinterpret_cast<const unsigned char*>(small_in) + (size_t)bj * CS * P * 4 + s_off);
      rx[j] = *reinterpret_cast<const f4u*>(reinterpret_cast<const unsigned char*>(big) + (size_t)bj * CB * P * 4 + x_off);
    }
  };
  auto pack8 = [&](const f4u (&r)[8], int e, bool aff, float sc, float sh) -> u32x4 {
    float v[8];
    if (live != 0xFFu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mj = ((live >> j) & 1u) ? 1.f : 0.f;
        v[j] = fmaf(r[j][e], sc * mj, sh * mj);
      }
    } else if (aff) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(r[j][e], sc, sh);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = r[j][e];
    }
    return u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  };
  auto commit = [&](unsigned char* st) {
    if (ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<u32x4*>(st + dst + e * (G::T * 16)) = pack8(rs, e, s_aff, s_sc, s_sh);
        *reinterpret_cast<u32x4*>(st + G::IMG + dst + e * (G::T * 16)) = pack8(rx, e, x_aff, x_sc, x_sh);
      }
    }
  };
  // fragments: A rows cs = (4 mh + t) * 16 + m, B columns cb = (2 nq + t) * 16 + m; pixel 4 step + kq
  const int a_frag = kq * (G::T * 16) + (4 * mh * 16 + m) * 16;             // + 256 per M tile, + 4 pixels per step
  const int b_frag = G::IMG + kq * (G::T * 16) + (2 * nq * 16 + m) * 16;   // + 256 per N tile
  f32x4 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (u0 < u1) {
    issue(u0);
    commit(ldsb);
  }
  if (u0 + 1 < u1) issue(u0 + 1);
  __syncthreads();
#pragma unroll 1
  for (int u = u0; u < u1; ++u) {
    const unsigned char* st = ldsb + ((u - u0) & 1) * G::STAGE;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      u32x4 a[4], b[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const u32x4*>(st + a_frag + sp * 4 * (G::T * 16) + t * 256);
#pragma unroll
      for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const u32x4*>(st + b_frag + sp * 4 * (G::T * 16) + t * 256);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) acc[t][t2] = mfma_bf16_k32(a[t], b[t2], acc[t][t2]);
      if (sp == 1 && u + 1 < u1) {
        commit(ldsb + ((u + 1 - u0) & 1) * G::STAGE);
        if (u + 2 < u1) issue(u + 2);
      }
    }
    __syncthreads();
  }
  float* o = outp + (nsplit > 1 ? (size_t)ks * CS * CB : 0);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t idx = (size_t)(cs0 + (4 * mh + t) * 16 + 4 * kq + i) * CB + cb0 + (2 * nq + t2) * 16 + m;
        o[idx] = acc[t][t2][i] + ((add && nsplit == 1) ? o[idx] : 0.f);
      }
}

// The 1x1 weight gradient with fp32 products as SIX bf16 instructions (PGV_COMPUTE_F32_SPLIT): the 128 x 128 tile of
// k1_wgrad_bf16_kernel with THREE plane images of both operands (x = x1 + x2 + x3 exactly, split where a block of 8 samples
// is committed) - 144 KB, so ONE stage: a matrix phase (3 K steps of 8 samples x 4 pixels; a wave multiplies 4 x 2 tiles:
// 18 fragment reads per 48 instructions) and a vector phase (the next block is converted and committed; its loads were
// issued before the matrix phase) with a barrier each that does not wait for vector memory.  The S items sit on threads
// 0 .. 383, the X items on threads 128 .. 511: three items per SIMD.  (Tried: half-quad items with 8-byte loads, exactly
// three per thread - the commit fell from 3.7 k to 2.5 k clocks per block, but 24 load instructions per thread instead of 16
// cost more than that wherever they were issued: 61.8 / 65.1 us against 61.3.)
struct K1WS {
  // PS: pixel stride of an image, [pixel][128 channels][8 samples] + one 16-byte slot: the three items of a channel (pixel
  // quads 0, 4, 8) then start 64 bytes apart modulo the bank period instead of on the same banks (3-way conflicts on every store)
  static constexpr int P = 12, T = 128, PS = T * 16 + 16, IMG = P * PS, OPER = 3 * IMG, STAGE = 2 * OPER, ITEMS = T * 3;
};

__global__ __launch_bounds__(512) void k1_wgrad_split_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                             const float* __restrict__ big_scale,
                                                             const float* __restrict__ big_shift,
                                                             const float* __restrict__ small_in,
                                                             const float* __restrict__ small_scale,
                                                             const float* __restrict__ small_shift,
                                                             float* __restrict__ outp, int nsplit, int add,
                                                             unsigned long long* __restrict__ stamps) {
  using G = K1WS;
  constexpr int P = G::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const pgv_split_sel sel = pgv_split_sel_make();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), mh = wave & 1, nq = wave >> 1;
  const int NBc = CB / G::T, tiles = (CS / G::T) * NBc;
  const int tile = blockIdx.x % tiles, ks = blockIdx.x / tiles;
  const int cs0 = (tile / NBc) * G::T, cb0 = (tile % NBc) * G::T;
  const int units = (B + 7) >> 3, per = (units + nsplit - 1) / nsplit;
  const int u0 = ks * per, u1 = min(units, u0 + per);

  const bool s_ok = tid < G::ITEMS, x_ok = tid >= 512 - G::ITEMS;
  const int qs = min(tid, G::ITEMS - 1), sch = qs / 3, sqi = qs - sch * 3;
  const int qx = max(tid - (512 - G::ITEMS), 0), xch = qx / 3, xqi = qx - xch * 3;
  const int s_off = ((cs0 + sch) * P + 4 * sqi) * 4, x_off = ((cb0 + xch) * P + 4 * xqi) * 4;   // bytes; + sample * C * P
  const int s_dst = (4 * sqi) * G::PS + sch * 16;                                        // + T * 16 per pixel
  const int x_dst = G::OPER + (4 * xqi) * G::PS + xch * 16;
  const float s_sc = small_scale ? small_scale[cs0 + sch] : 1.f, s_sh = small_scale ? small_shift[cs0 + sch] : 0.f;
  const float x_sc = big_scale ? big_scale[cb0 + xch] : 1.f, x_sh = big_scale ? big_shift[cb0 + xch] : 0.f;
  const bool s_aff = small_scale != nullptr, x_aff = big_scale != nullptr;

  f4u rs[8], rx[8];
  unsigned live = 0xFFu;
  auto issue = [&](int u) {
    const int b = u * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int bj = min(b + j, B - 1);
      rs[j] = *reinterpret_cast<const f4u*>(reinterpret_cast<const unsigned char*>(small_in) + (size_t)bj * CS * P * 4 + s_off);
      rx[j] = *reinterpret_cast<const f4u*>(reinterpret_cast<const unsigned char*>(big) + (size_t)bj * CB * P * 4 + x_off);
    }
  };
  // the 8 samples of pixel e of an item -> three 16-byte entries (one per plane)
  auto store8 = [&](unsigned char* dst, const f4u (&r)[8], int e, bool aff, float sc, float sh) {
    float v[8];
    if (live != 0xFFu) {      // partial last block (uniform): per-sample masks
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mj = ((live >> j) & 1u) ? 1.f : 0.f;
        v[j] = fmaf(r[j][e], sc * mj, sh * mj);
      }
    } else if (aff) {         // (uniform)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(r[j][e], sc, sh);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = r[j][e];
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned a, b, c;
      pgv_split3_pair(v[2 * j], v[2 * j + 1], a, b, c, sel);
      ph[j] = a, pm[j] = b, pl[j] = c;
    }
    *reinterpret_cast<u32x4*>(dst) = ph;
    *reinterpret_cast<u32x4*>(dst + G::IMG) = pm;
    *reinterpret_cast<u32x4*>(dst + 2 * G::IMG) = pl;
  };
  auto commit = [&](int u) {
    live = (1u << min(max(B - u * 8, 0), 8)) - 1u;
    if (s_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) store8(ldsb + s_dst + e * G::PS, rs, e, s_aff, s_sc, s_sh);
    }
    if (x_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) store8(ldsb + x_dst + e * G::PS, rx, e, x_aff, x_sc, x_sh);
    }
  };
  auto sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // fragments: A rows cs = (4 mh + t) * 16 + m, B columns cb = (2 nq + t) * 16 + m; pixel 4 step + kq
  const int a_frag = kq * G::PS + (4 * mh * 16 + m) * 16;                // + 256 per M tile, + 4 pixels per step
  const int b_frag = G::OPER + kq * G::PS + (2 * nq * 16 + m) * 16;     // + 256 per N tile
  f32x4 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  BSTAMP(0);
  if (u0 < u1) {
    issue(u0);
    commit(u0);
  }
  if (u0 + 1 < u1) issue(u0 + 1);
  sync();
  BSTAMP(1);
#pragma unroll 1
  for (int u = u0; u < u1; ++u) {
    // ---- matrix phase
    if (u == u0 + 2) BSTAMP(2);
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      u32x4 a[4][3], b[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t][p] = *reinterpret_cast<const u32x4*>(ldsb + p * G::IMG + a_frag + sp * 4 * G::PS + t * 256);
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t][p] = *reinterpret_cast<const u32x4*>(ldsb + p * G::IMG + b_frag + sp * 4 * G::PS + t * 256);
      }
      // the six products, smallest first, over eight independent accumulators
      constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) acc[t][t2] = mfma_bf16_k32(a[t][TA[term]], b[t2][TB[term]], acc[t][t2]);
    }
    if (u == u0 + 2) BSTAMP(3);
    sync();
    if (u == u0 + 2) BSTAMP(4);
    // ---- vector phase: the next block of samples
    if (u + 1 < u1) {
      commit(u + 1);
      if (u + 2 < u1) issue(u + 2);
    }
    if (u == u0 + 2) BSTAMP(5);
    sync();
    if (u == u0 + 2) BSTAMP(6);
  }
  BSTAMP(7);
  float* o = outp + (nsplit > 1 ? (size_t)ks * CS * CB : 0);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t idx = (size_t)(cs0 + (4 * mh + t) * 16 + 4 * kq + i) * CB + cb0 + (2 * nq + t2) * 16 + m;
        o[idx] = acc[t][t2][i] + ((add && nsplit == 1) ? o[idx] : 0.f);
      }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BSTAMP(8);
}

template <int H, int W, int R, int WP>
int launch_deep_wgrad_bf16(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                           const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                           void* workspace, int64_t workspace_bytes, int nsplit, hipStream_t st) {
  using G = WgradB<H, W, R, WP>;
  if (d->Cs % 64 || d->Cb % 8 || d->B <= 0) return 0;
  const int64_t gw_bytes = (int64_t)d->Cs * d->Cb * 64;
  const int units = ((d->B + 15) / 16) * G::BANDS;
  nsplit = max(1, min(nsplit, units));
  if (nsplit > 1 && (!workspace || workspace_bytes < nsplit * gw_bytes || ((uintptr_t)workspace & 15) || ((uintptr_t)gw & 15)))
    nsplit = 1;   // no room for partial gradients: one workgroup per tile
  const size_t bytes = 2 * (size_t)G::STAGE;
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = deep_wgrad_bf16_kernel<G>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_wgrad_deep_bf16");
  if (rc) return rc;
  const int add = (d->flags & PGV_PREZEROED) ? 1 : 0;
  const int grid = (d->Cs / 64) * (d->Cb / 8) * nsplit;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), bytes, st, d->B, d->Cb, d->Cs, big, big_scale, big_shift, small_in,
                     small_scale, small_shift, nsplit > 1 ? (float*)workspace : gw, nsplit, add, g_deep_bf16_stamps);
  PGV_CHECK_LAUNCH("conv_wgrad_deep_bf16");
  if (nsplit > 1) {
    const int n4 = (int)(gw_bytes / 16);
    hipLaunchKernelGGL(deep_wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       (const f32x4*)workspace, nsplit, n4, (f32x4*)gw, add);
    PGV_CHECK_LAUNCH("conv_wgrad_deep_bf16 reduce");
  }
  return 1;
}

template <int H, int W, int R, int WP, int NP = 1>
int launch_deep_wgrad8_bf16(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                            const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                            void* workspace, int64_t workspace_bytes, int nsplit, hipStream_t st) {
  using G = Wgrad8<H, W, R, WP, NP>;
  if (d->Cs % 64 || d->Cb % 8 || d->B <= 0) return 0;
  const int64_t gw_bytes = (int64_t)d->Cs * d->Cb * 64;
  const int units = ((d->B + 7) / 8) * G::BANDS;
  nsplit = max(1, min(nsplit, units));
  if (nsplit > 1 && (!workspace || workspace_bytes < nsplit * gw_bytes || ((uintptr_t)workspace & 15) || ((uintptr_t)gw & 15)))
    nsplit = 1;
  auto kern = deep_wgrad8_bf16_kernel<G>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_wgrad_deep_bf16");
  if (rc) return rc;
  const int add = (d->flags & PGV_PREZEROED) ? 1 : 0;
  const int grid = (d->Cs / 64) * (d->Cb / 8) * nsplit;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 2 * (size_t)G::STAGE, st, d->B, d->Cb, d->Cs, big, big_scale,
                     big_shift, small_in, small_scale, small_shift, nsplit > 1 ? (float*)workspace : gw, nsplit, add);
  PGV_CHECK_LAUNCH("conv_wgrad_deep_bf16");
  if (nsplit > 1) {
    const int n4 = (int)(gw_bytes / 16);
    hipLaunchKernelGGL(deep_wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       (const f32x4*)workspace, nsplit, n4, (f32x4*)gw, add);
    PGV_CHECK_LAUNCH("conv_wgrad_deep_bf16 reduce");
  }
  return 1;
}

// tiles of 64 x 8 channels over 256 CUs: how many workgroups share the units of a tile
int deep_wgrad_bf16_split(const pgv_conv_desc* d) {
  const int tiles = (d->Cs / 64) * (d->Cb / 8);
  return tiles >= 192 ? 1 : max(1, 256 / max(1, tiles));
}


// ---------------------------------------------------------------------------------------------------------------
// 1x1 layers on 3x4 planes (enc8 / dec1: 512 <-> 2048 channels, model/encoder.py:64-69, model/decoder.py:72-75):
// out[b,m,p] = act(bias[m] + sum_k Wt[m][k] * in'[b,k,p]) - both directions are this one product (forward: m = cs, k = cb,
// Wt = the weight; transposed: m = cb, k = cs, Wt = its transpose), the shadow holds both as [m][k] bf16.  One workgroup =
// 128 output channels (16 per wave) x 4 samples (48 pixels = 3 tiles): every wave runs the whole K, no reduction.  Images
// are channel-innermost, 64 channels = 128 bytes per pixel, the 16-byte group g of a pixel stored at g ^ (pixel & 7).
__device__ __forceinline__ void shadow_k1_item(int it, const float* __restrict__ w, int CS, int CB, u16* __restrict__ down,
                                               u16* __restrict__ up) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)it * 8), b = *reinterpret_cast<const f32x4*>(w + (size_t)it * 8 + 4);
  *reinterpret_cast<u32x4*>(down + (size_t)it * 8) =
      u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  // transposed: up[cb][8 consecutive cs]; cb fastest across the lanes (coalesced reads of 8 weight rows)
  const int cb = it % CB, g = it / CB;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = w[(size_t)(g * 8 + c) * CB + cb];
  *reinterpret_cast<u32x4*>(up + (size_t)cb * CS + g * 8) =
      u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}
__global__ __launch_bounds__(256) void k1_shadow_kernel(const float* __restrict__ w, int CS, int CB, u16* __restrict__ down,
                                                        u16* __restrict__ up) {
  const int n8 = CS * CB / 8;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < n8; it += gridDim.x * 256) shadow_k1_item(it, w, CS, CB, down, up);
}

// the shadows of several layers in ONE launch (a conv stack's forward pass: 4 - 6 us of launch latency per layer otherwise)
struct ShadowTable {
  static constexpr int MAXN = 8;
  const float* w[MAXN];
  u16* down[MAXN];
  int CS[MAXN], CB[MAXN], k1[MAXN], items[MAXN], blk0[MAXN + 1];
  int n;
};
__global__ __launch_bounds__(256) void shadow_multi_kernel(ShadowTable t) {
  int e = 0;
  while (e + 1 < t.n && (int)blockIdx.x >= t.blk0[e + 1]) ++e;
  const int it = ((int)blockIdx.x - t.blk0[e]) * 256 + threadIdx.x;
  if (it >= t.items[e]) return;
  u16* up = t.down[e] + (size_t)t.CS[e] * t.CB[e] * (t.k1[e] ? 1 : 16);
  if (t.k1[e] == 3) {   // deep split: the down layout, then the up layout (both in fragment order, three planes each)
    const int nd = t.CS[e] * t.CB[e] * 2;
    if (it < nd)
      shadow_split_down_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e]);
    else
      shadow_split_up_item(it - nd, t.w[e], t.CS[e], t.CB[e], t.down[e] + (size_t)3 * t.CS[e] * t.CB[e] * 16);
  } else if (t.k1[e] == 5) {   // large-plane split (conv_big_split.hip): down fragments, then up fragments
    const int nd = t.CS[e] * t.CB[e] * 2;
    if (it < nd)
      shadow_bigq_down_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e]);
    else
      shadow_bigq_up_item(it - nd, t.w[e], t.CS[e], t.CB[e], t.down[e] + (size_t)3 * t.CS[e] * t.CB[e] * 16);
  } else if (t.k1[e] == 6) {   // bf16 operand mode on the large-plane kernels: the same fragment orders, one plane each
    const int nd = t.CS[e] * t.CB[e] * 2;
    if (it < nd)
      shadow_bigq_down_item<1>(it, t.w[e], t.CS[e], t.CB[e], t.down[e]);
    else
      shadow_bigq_up_item<1>(it - nd, t.w[e], t.CS[e], t.CB[e], t.down[e] + (size_t)t.CS[e] * t.CB[e] * 16);
  } else if (t.k1[e] == 4)
    shadow_split_k1_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e]);
  else if (t.k1[e])
    shadow_k1_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e], up);
  else
    shadow_k4_item(it, t.w[e], t.CS[e], t.CB[e], t.down[e], up);
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

bool k1_bf16_shape(const pgv_conv_desc* d) {
  return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->Hb == 3 && d->Wb == 4 && d->Cb % 128 == 0 &&
         d->Cs % 128 == 0;
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


bool deep_bf16_shape(const pgv_conv_desc* d) {
  return d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 2 && d->Cb >= 64 && d->Cb % 16 == 0 && d->Cs % 64 == 0 &&
         ((d->Hb == 17 && d->Wb == 23) || (d->Hb == 9 && d->Wb == 12) || (d->Hb == 5 && d->Wb == 7));
}

}  // namespace

// bytes of the bf16 weight shadow of a layer (down + up layouts), 0: the layer has no bf16-native kernels
int64_t pgv_conv_weight_shadow_bytes_impl(const pgv_conv_desc* d) {
  if (!(d->flags & PGV_COMPUTE_BF16)) {   // PGV_COMPUTE_F32_SPLIT: 3 bf16 planes; the deep layers hold a down and an up layout
    if (pgv_deep_split_shape(d)) return (int64_t)12 * d->Cs * d->Cb * 16;
    if (pgv_k1_split_shape(d)) return (int64_t)12 * d->Cs * d->Cb;
    return pgv_big_split_shape(d) ? (int64_t)12 * d->Cs * d->Cb * 16 : 0;
  }
  if (k1_bf16_shape(d)) return (int64_t)4 * d->Cs * d->Cb;
  return (deep_bf16_shape(d) || pgv_big_bf16q_shape(d)) ? (int64_t)4 * d->Cs * d->Cb * 16 : 0;
}

int pgv_conv_weight_shadow_impl(const pgv_conv_desc* d, const float* w, void* shadow, hipStream_t st) {
  if (!(d->flags & PGV_COMPUTE_BF16) || pgv_big_bf16q_shape(d)) {
    if (pgv_deep_split_shape(d) || pgv_k1_split_shape(d) || pgv_big_split_shape(d) || pgv_big_bf16q_shape(d)) {
      const pgv_conv_desc* one[1] = {d};
      const float* ws[1] = {w};
      void* sh[1] = {shadow};
      return pgv_conv_weight_shadows_impl(1, one, ws, sh, st);
    }
    return 0;
  }
  if (k1_bf16_shape(d)) {
    u16* down = (u16*)shadow;
    hipLaunchKernelGGL(k1_shadow_kernel, dim3((unsigned)min((d->Cs * d->Cb / 8 + 255) / 256, 2048)), dim3(256), 0, st, w, d->Cs,
                       d->Cb, down, down + (size_t)d->Cs * d->Cb);
    PGV_CHECK_LAUNCH("conv_weight_shadow");
    return 1;
  }
  if (!deep_bf16_shape(d)) return 0;
  u16* down = (u16*)shadow;
  u16* up = down + (size_t)d->Cs * d->Cb * 16;
  const int items = d->Cs * (d->Cb / 8) * 4;
  hipLaunchKernelGGL(deep_shadow_kernel, dim3((unsigned)min((items + 255) / 256, 2048)), dim3(256), 0, st, w, d->Cs, d->Cb,
                     down, up);
  PGV_CHECK_LAUNCH("conv_weight_shadow");
  return 1;
}

// 1 = launched, 0 = not this kernel family's case (no shadow in the descriptor, shape not covered)
int pgv_conv_down_deep_bf16(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                            const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                            const pgv_bn_src* bn) {
  if ((d->flags & PGV_COMPUTE_BF16) && d->w_shadow && k1_bf16_shape(d))
    return launch_k1_fwd_bf16(d, false, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (!(d->flags & PGV_COMPUTE_BF16) || !d->w_shadow || !deep_bf16_shape(d)) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_down_bf16<17, 23, 1, true>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_down_bf16<9, 12, 4, false>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_down_bf16<5, 7, 8, true>(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  return 0;
}

int pgv_conv_up_deep_bf16(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                          const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                          const pgv_bn_src* bn) {
  if ((d->flags & PGV_COMPUTE_BF16) && d->w_shadow && k1_bf16_shape(d))
    return launch_k1_fwd_bf16(d, true, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (!(d->flags & PGV_COMPUTE_BF16) || !d->w_shadow || !deep_bf16_shape(d)) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_up_bf16<17, 23, 2>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_up_bf16<9, 12, 4>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_up_bf16<5, 7, 8>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn);
  return 0;
}

static int k1_wgrad_split(const pgv_conv_desc* d) {
  const int tiles = (d->Cs / 128) * (d->Cb / 128);
  return tiles >= 192 ? 1 : max(1, min(8, 256 / max(1, tiles)));
}

int64_t pgv_conv_wgrad_deep_bf16_workspace(const pgv_conv_desc* d) {
  if (((d->flags & PGV_COMPUTE_BF16) && k1_bf16_shape(d)) || pgv_k1_split_shape(d)) {
    const int ns = k1_wgrad_split(d);
    return ns > 1 ? (int64_t)ns * d->Cs * d->Cb * 4 : 0;
  }
  if (!((d->flags & PGV_COMPUTE_BF16) ? deep_bf16_shape(d) : pgv_deep_split_shape(d)) || d->Cb % 8) return 0;
  const int ns = deep_wgrad_bf16_split(d);
  return ns > 1 ? (int64_t)ns * d->Cs * d->Cb * 64 : 0;
}

int pgv_conv_wgrad_deep_bf16(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                             const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                             void* workspace, int64_t workspace_bytes, hipStream_t st) {
  if ((d->flags & PGV_COMPUTE_BF16) && k1_bf16_shape(d) && !(g_deep_bf16_dbg & 8) && d->B > 0) {
    const int64_t gw_bytes = (int64_t)d->Cs * d->Cb * 4;
    int nsplit = min(k1_wgrad_split(d), (d->B + 7) / 8);
    if (nsplit > 1 && (!workspace || workspace_bytes < nsplit * gw_bytes || ((uintptr_t)workspace & 15) || ((uintptr_t)gw & 15)))
      nsplit = 1;
    static bool attr_done = false;
    int rc = raise_lds_limit(k1_wgrad_bf16_kernel, &attr_done, "conv_wgrad_k1_bf16");
    if (rc) return rc;
    const int add = (d->flags & PGV_PREZEROED) ? 1 : 0;
    const int grid = (d->Cs / 128) * (d->Cb / 128) * nsplit;
    hipLaunchKernelGGL(k1_wgrad_bf16_kernel, dim3((unsigned)grid), dim3(512), 2 * (size_t)K1W::STAGE, st, d->B, d->Cb, d->Cs, big,
                       big_scale, big_shift, small_in, small_scale, small_shift, nsplit > 1 ? (float*)workspace : gw, nsplit, add);
    PGV_CHECK_LAUNCH("conv_wgrad_k1_bf16");
    if (nsplit > 1) {
      const int n4 = (int)(gw_bytes / 16);
      hipLaunchKernelGGL(deep_wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                         (const f32x4*)workspace, nsplit, n4, (f32x4*)gw, add);
      PGV_CHECK_LAUNCH("conv_wgrad_k1_bf16 reduce");
    }
    return 1;
  }
  if (pgv_k1_split_shape(d) && d->B > 0) {   // the 1x1 weight gradient with six-instruction products
    const int64_t gw_bytes = (int64_t)d->Cs * d->Cb * 4;
    int nsplit = min(k1_wgrad_split(d), (d->B + 7) / 8);
    if (nsplit > 1 && (!workspace || workspace_bytes < nsplit * gw_bytes || ((uintptr_t)workspace & 15) || ((uintptr_t)gw & 15)))
      nsplit = 1;
    static bool attr_done = false;
    int rc = raise_lds_limit(k1_wgrad_split_kernel, &attr_done, "conv_wgrad_k1_split");
    if (rc) return rc;
    const int add = (d->flags & PGV_PREZEROED) ? 1 : 0;
    const int grid = (d->Cs / 128) * (d->Cb / 128) * nsplit;
    hipLaunchKernelGGL(k1_wgrad_split_kernel, dim3((unsigned)grid), dim3(512), (size_t)K1WS::STAGE, st, d->B, d->Cb, d->Cs, big,
                       big_scale, big_shift, small_in, small_scale, small_shift, nsplit > 1 ? (float*)workspace : gw, nsplit, add,
                       g_deep_bf16_stamps);
    PGV_CHECK_LAUNCH("conv_wgrad_k1_split");
    if (nsplit > 1) {
      const int n4 = (int)(gw_bytes / 16);
      hipLaunchKernelGGL(deep_wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                         (const f32x4*)workspace, nsplit, n4, (f32x4*)gw, add);
      PGV_CHECK_LAUNCH("conv_wgrad_k1_split reduce");
    }
    return 1;
  }
  if (pgv_deep_split_shape(d)) {   // fp32 products as six bf16 instructions: blocks of 8 samples, three plane images
    // (the operands are split in the loader.  A pre-pass that writes both operands as plane tensors in the image layout, so
    // that the loader is 16-byte copies without conversions, was measured at 92 / 106 / 100 us against 71 / 103 / 99 us: the
    // matrix loop between the unit barriers is the bound - 2 - 3 K steps of 24 instructions per wave - not the conversions)
    const int ns = deep_wgrad_bf16_split(d);
    if (d->Hb == 17 && d->Wb == 23)
      return launch_deep_wgrad8_bf16<17, 23, 1, 28, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
    if (d->Hb == 9 && d->Wb == 12)
      return launch_deep_wgrad8_bf16<9, 12, 1, 20, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
    return launch_deep_wgrad8_bf16<5, 7, 3, 12, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
  }
  if (!(d->flags & PGV_COMPUTE_BF16) || !deep_bf16_shape(d) || (g_deep_bf16_dbg & 8)) return 0;
  const int ns = deep_wgrad_bf16_split(d);
  if (d->Hb == 17 && d->Wb == 23)
    return launch_deep_wgrad_bf16<17, 23, 1, 28>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
  if (d->Hb == 9 && d->Wb == 12) {
    if (g_deep_bf16_dbg & 64)   // (A/B: blocks of 16 samples, one output row per unit)
      return launch_deep_wgrad_bf16<9, 12, 1, 20>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
    return launch_deep_wgrad8_bf16<9, 12, 5, 20>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
  }
  if (d->Hb == 17 && d->Wb == 23 && (g_deep_bf16_dbg & 128))   // (A/B: blocks of 8 samples, three output rows per unit)
    return launch_deep_wgrad8_bf16<17, 23, 3, 28>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
  if (d->Hb == 5 && d->Wb == 7)
    return launch_deep_wgrad_bf16<5, 7, 3, 12>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
  return 0;
}

// The large-plane k4 s2 p2 layers with a weight shadow in the descriptor (1 / 3 = launched, 3: with the class sums of the fused
// epilogue): conv_big_split.hip - fp32 products as six bf16 instructions (PGV_COMPUTE_F32_SPLIT, three operand planes) and,
// since round 6, bf16 operand mode on the same kernels with one plane (the round-4 bf16 kernels of these layers, up_big_bf16 /
// down_big_bf16, are gone: slower than the six-instruction kernels at a sixth of their matrix work, and their shadow layout
// with them).  0 = not this family's case: the callers (conv_v2_down / conv_v2_up) go on to kernels that read the weights.
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
int pgv_conv_weight_shadows_impl(int n, const pgv_conv_desc* const* descs, const float* const* ws, void* const* shadows,
                                 hipStream_t st) {
  ShadowTable t;
  t.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const pgv_conv_desc* d = descs[i];
    const bool bf = (d->flags & PGV_COMPUTE_BF16) != 0, split = pgv_big_split_shape(d), dsplit = pgv_deep_split_shape(d);
    const bool k1 = bf && k1_bf16_shape(d), k1split = pgv_k1_split_shape(d), bfq = pgv_big_bf16q_shape(d);
    if (!split && !dsplit && !k1split && !bfq && !(bf && (k1 || deep_bf16_shape(d)))) return 0;
    t.w[i] = ws[i];
    t.down[i] = (u16*)shadows[i];
    t.CS[i] = d->Cs, t.CB[i] = d->Cb;
    // kind: 0 k4 bf16, 1 1x1 bf16, 3 deep split down + up fragments, 4 split 1x1 fragments, 5 large-plane split fragments,
    // 6 large-plane fragments with one plane (bf16 operand mode)
    t.k1[i] = k1split ? 4 : dsplit ? 3 : split ? 5 : bfq ? 6 : (k1 ? 1 : 0);
    t.items[i] = k1split ? d->Cs * d->Cb / 4
                 : (dsplit || split || bfq) ? d->Cs * d->Cb * 4
                                     : (k1 ? d->Cs * d->Cb / 8 : d->Cs * (d->Cb / 8) * 4);
    t.blk0[i] = blocks;
    blocks += (t.items[i] + 255) / 256;
  }
  t.blk0[n] = blocks;
  hipLaunchKernelGGL(shadow_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t);
  PGV_CHECK_LAUNCH("conv_weight_shadows");
  return 1;
}
