// bf16-native WEIGHT GRADIENTS of the deep k4 s2 p2 layers and of the 1x1 layers on 3x4 planes (PGV_COMPUTE_BF16, and the
// three-plane forms of PGV_COMPUTE_F32_SPLIT that share their structure); split out of conv_deep_bf16.hip in round 6 - the
// layouts and the image conventions are described there.
#include "conv_tile.h"
#include "conv_deep_common.h"

namespace {

typedef unsigned short u16;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
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
                     small_scale, small_shift, nsplit > 1 ? (float*)workspace : gw, nsplit, add, pgv_deep_bf16_stamps());
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


}  // namespace

static int k1_wgrad_split(const pgv_conv_desc* d) {
  const int tiles = (d->Cs / 128) * (d->Cb / 128);
  return tiles >= 192 ? 1 : max(1, min(8, 256 / max(1, tiles)));
}

int64_t pgv_conv_wgrad_deep_bf16_workspace(const pgv_conv_desc* d) {
  if (((d->flags & PGV_COMPUTE_BF16) && pgv_k1_bf16_shape(d)) || pgv_k1_split_shape(d)) {
    const int ns = k1_wgrad_split(d);
    return ns > 1 ? (int64_t)ns * d->Cs * d->Cb * 4 : 0;
  }
  if (!((d->flags & PGV_COMPUTE_BF16) ? pgv_deep_bf16_shape(d) : pgv_deep_split_shape(d)) || d->Cb % 8) return 0;
  const int ns = deep_wgrad_bf16_split(d);
  return ns > 1 ? (int64_t)ns * d->Cs * d->Cb * 64 : 0;
}

int pgv_conv_wgrad_deep_bf16(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                             const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                             void* workspace, int64_t workspace_bytes, hipStream_t st) {
  if ((d->flags & PGV_COMPUTE_BF16) && pgv_k1_bf16_shape(d) && !(pgv_deep_bf16_dbg() & 8) && d->B > 0) {
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
                       pgv_deep_bf16_stamps());
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
  if (!(d->flags & PGV_COMPUTE_BF16) || !pgv_deep_bf16_shape(d) || (pgv_deep_bf16_dbg() & 8)) return 0;
  const int ns = deep_wgrad_bf16_split(d);
  if (d->Hb == 17 && d->Wb == 23)
    return launch_deep_wgrad_bf16<17, 23, 1, 28>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
  if (d->Hb == 9 && d->Wb == 12) {
    if (pgv_deep_bf16_dbg() & 64)   // (A/B: blocks of 16 samples, one output row per unit)
      return launch_deep_wgrad_bf16<9, 12, 1, 20>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
    return launch_deep_wgrad8_bf16<9, 12, 5, 20>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, ns, st);
  }
  if (d->Hb == 17 && d->Wb == 23 && (pgv_deep_bf16_dbg() & 128))   // (A/B: blocks of 8 samples, three output rows per unit)
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
