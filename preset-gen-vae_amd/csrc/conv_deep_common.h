// Shared by the deep-layer kernels (conv_deep.hip: fp32 LDS images; conv_deep_bf16.hip: bf16-native images).
#pragma once
#include "conv_tile.h"

namespace {

// XCD-aware (channel block, sample group) of a workgroup.  Workgroups are dealt round-robin over the 8 XCDs, each with
// its own L2: with blockIdx = mb * groups + grp every XCD works on ALL channel blocks at once and streams the whole
// weight tensor (up to 8 MB against a 4 MB L2) again and again.  Here the workgroups of one XCD share a channel block
// whenever the block count divides 8 (its weight slice, 1 MB or less, then stays in that XCD's L2 for all sample
// groups); a speed matter only.
__device__ __forceinline__ void deep_block(int nmb, int groups, int& mb, int& grp) {
  const int b = (int)blockIdx.x;
  if (nmb <= 8 && 8 % nmb == 0 && (nmb * groups) % 8 == 0) {
    const int x = b & 7, q = b >> 3, per = 8 / nmb;   // per = XCDs per channel block
    mb = x % nmb;
    grp = q * per + x / nmb;
  } else if (nmb % 8 == 0) {   // more channel blocks than XCDs: nmb / 8 of them per XCD, one after the other
    const int x = b & 7, q = b >> 3;
    mb = x * (nmb / 8) + q / groups;
    grp = q % groups;
  } else {
    mb = b / groups;
    grp = b - mb * groups;
  }
}

// PGV_COMPUTE_F32_SPLIT: x = h + m + l exactly, three bfloat16 terms (8 + 8 + 8 significant bits cover fp32's 24)
__device__ __forceinline__ void pgv_split3(float x, float& h, float& m, float& l) {
  h = (float)(__bf16)x;
  const float r = x - h;
  m = (float)(__bf16)r;
  l = r - m;   // (at most 8 significant bits: exact as bfloat16)
}
__device__ __forceinline__ unsigned pgv_pack_bf16x2(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// two values at once, as the packed bf16 pairs (low half = a) the images hold: 3 packed conversions and 4 residuals per
// pair.  A residual x - bf16(x) is ONE v_dot2c_f32_bf16 on the packed pair itself (x + pair . {-1, 0} for the low half,
// {0, -1} for the high half: one exact product, one exactly representable sum - bit-identical to the subtraction from the
// unpacked term for every finite input below 2^127.99, scratch/ubench/dot2_residual.hip), instead of a shift / mask that
// moves the half to the top of a dword plus a subtraction.  The selectors live in scalar registers: written as literals the
// packed {-1.0, 0} becomes the inline constant "-1.0", which the hardware reads as the fp32 pattern, i.e. as {0, -1.0}.
struct pgv_split_sel {
  unsigned lo, hi;   // packed bf16 {-1, 0} and {0, -1}, in scalar registers
};
__device__ __forceinline__ pgv_split_sel pgv_split_sel_make() {
  unsigned sel_lo = 0x0000bf80u, sel_hi = 0xbf800000u;
  asm volatile("" : "+s"(sel_lo), "+s"(sel_hi));
  return pgv_split_sel{sel_lo, sel_hi};
}
// (sel: made ONCE per kernel - made inside, every call costs two s_mov)
__device__ __forceinline__ void pgv_split3_pair(float a, float b, unsigned& H, unsigned& M, unsigned& L, const pgv_split_sel& sel) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 lo1 = __builtin_bit_cast(bf2, sel.lo), hi1 = __builtin_bit_cast(bf2, sel.hi);
  H = pgv_pack_bf16x2(a, b);
  const float ra = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, H), lo1, a, false);
  const float rb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, H), hi1, b, false);
  M = pgv_pack_bf16x2(ra, rb);
  L = pgv_pack_bf16x2(__builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, M), lo1, ra, false),
                      __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, M), hi1, rb, false));
}
__device__ __forceinline__ void pgv_split3_pair(float a, float b, unsigned& H, unsigned& M, unsigned& L) {
  pgv_split3_pair(a, b, H, M, L, pgv_split_sel_make());
}

// Split weight shadow of a deep k4 s2 p2 layer, DOWN layout = the fragment order of deep_down_split_kernel: the 16 bytes
// (8 channels of slab s at kernel tap (kh, kq)) that lane (m, kq) of wave (half, kh) feeds the matrix instruction as the
// A operand of rows mb*64 + half*32 + mt*16 + m, for plane p, at
//   (((((mb * nslab + s) * 8 + wave) * 3 + p) * 2 + mt) * 64 + lane) * 16 bytes:
// a wave's six fragments of a slab are 6 KB of contiguous memory, read straight into registers (no LDS copy: with K
// split over the waves no two waves share a weight element).  One item = (mb, s, wave, mt, lane).
__device__ __forceinline__ void shadow_split_down_item(int it, const float* __restrict__ w, int CS, int CB,
                                                       unsigned short* __restrict__ down3) {
  const int lane = it & 63, mt = (it >> 6) & 1, wave = (it >> 7) & 7, rest = it >> 10;
  const int nslab = CB / 8, s = rest % nslab, mb = rest / nslab;
  const int m = lane & 15, kq = lane >> 4, half = wave >> 2, kh = wave & 3;
  const int cs = mb * 64 + half * 32 + mt * 16 + m;
  float h[8], mid[8], l[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) pgv_split3(w[((size_t)cs * CB + s * 8 + c) * 16 + kh * 4 + kq], h[c], mid[c], l[c]);
  u32x4* dst = reinterpret_cast<u32x4*>(down3) + ((size_t)((rest * 8 + wave) * 3) * 2 + mt) * 64 + lane;
  dst[0] = u32x4{pgv_pack_bf16x2(h[0], h[1]), pgv_pack_bf16x2(h[2], h[3]), pgv_pack_bf16x2(h[4], h[5]), pgv_pack_bf16x2(h[6], h[7])};
  dst[128] = u32x4{pgv_pack_bf16x2(mid[0], mid[1]), pgv_pack_bf16x2(mid[2], mid[3]), pgv_pack_bf16x2(mid[4], mid[5]), pgv_pack_bf16x2(mid[6], mid[7])};
  dst[256] = u32x4{pgv_pack_bf16x2(l[0], l[1]), pgv_pack_bf16x2(l[2], l[3]), pgv_pack_bf16x2(l[4], l[5]), pgv_pack_bf16x2(l[6], l[7])};
}

// UP layout of the split shadow = the fragment order of deep_up_split_kernel: big-channel blocks of 32, wave wv = 2 * phase +
// M half; the 16 bytes (small channels g*8 .. g*8+7 at the phase's tap kq = 2 th + tw, i.e. kernel tap (ph + 2 th, pw + 2 tw))
// of row mb*32 + half*16 + m, plane p, at ((((mb * (CS/8) + g) * 8 + wv) * 3 + p) * 64 + lane) * 16 bytes.
// One item = (mb, g, wv, lane).
__device__ __forceinline__ void shadow_split_up_item(int it, const float* __restrict__ w, int CS, int CB,
                                                     unsigned short* __restrict__ up3) {
  const int lane = it & 63, wv = (it >> 6) & 7, rest = it >> 9;
  const int ng = CS / 8, g = rest % ng, mb = rest / ng;
  const int m = lane & 15, kq = lane >> 4, phase = wv >> 1, half = wv & 1;
  const int cb = mb * 32 + half * 16 + m, kh = (phase >> 1) + 2 * (kq >> 1), kw = (phase & 1) + 2 * (kq & 1);
  float h[8], mid[8], l[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) pgv_split3(w[((size_t)(g * 8 + c) * CB + cb) * 16 + kh * 4 + kw], h[c], mid[c], l[c]);
  u32x4* dst = reinterpret_cast<u32x4*>(up3) + ((size_t)(rest * 8 + wv) * 3) * 64 + lane;
  dst[0] = u32x4{pgv_pack_bf16x2(h[0], h[1]), pgv_pack_bf16x2(h[2], h[3]), pgv_pack_bf16x2(h[4], h[5]), pgv_pack_bf16x2(h[6], h[7])};
  dst[64] = u32x4{pgv_pack_bf16x2(mid[0], mid[1]), pgv_pack_bf16x2(mid[2], mid[3]), pgv_pack_bf16x2(mid[4], mid[5]), pgv_pack_bf16x2(mid[6], mid[7])};
  dst[128] = u32x4{pgv_pack_bf16x2(l[0], l[1]), pgv_pack_bf16x2(l[2], l[3]), pgv_pack_bf16x2(l[4], l[5]), pgv_pack_bf16x2(l[6], l[7])};
}

// Split shadow of a 1x1 layer, fragment order of k1_fwd_split_kernel: both directions are the product out[m] = sum_k Wt[m][k]
// in[k] (forward: m = cs, k = cb; transposed: m = cb, k = cs).  The 16 bytes (k = ks*32 + kq*8 .. +7) of row r16*16 + m,
// plane p, at (((r16 * (K/32) + ks) * 3 + p) * 64 + lane) * 16 bytes; the forward layout (Cs*Cb*6 bytes), then the transposed
// one.  One item = (direction, r16, ks, lane): Cs*Cb/8 items per direction.
__device__ __forceinline__ void shadow_split_k1_item(int it, const float* __restrict__ w, int CS, int CB,
                                                     unsigned short* __restrict__ sh) {
  const int per = CS * CB / 8, up = it >= per;
  if (up) it -= per;
  const int K = up ? CS : CB, nks = K / 32;
  const int lane = it & 63, rest = it >> 6, ks = rest % nks, r16 = rest / nks;
  const int m = r16 * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
  float h[8], mid[8], l[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) pgv_split3(up ? w[(size_t)(k0 + c) * CB + m] : w[(size_t)m * CB + k0 + c], h[c], mid[c], l[c]);
  u32x4* dst = reinterpret_cast<u32x4*>(sh) + (up ? (size_t)CS * CB * 3 / 8 : 0) + ((size_t)(r16 * nks + ks) * 3) * 64 + lane;
  dst[0] = u32x4{pgv_pack_bf16x2(h[0], h[1]), pgv_pack_bf16x2(h[2], h[3]), pgv_pack_bf16x2(h[4], h[5]), pgv_pack_bf16x2(h[6], h[7])};
  dst[64] = u32x4{pgv_pack_bf16x2(mid[0], mid[1]), pgv_pack_bf16x2(mid[2], mid[3]), pgv_pack_bf16x2(mid[4], mid[5]), pgv_pack_bf16x2(mid[6], mid[7])};
  dst[128] = u32x4{pgv_pack_bf16x2(l[0], l[1]), pgv_pack_bf16x2(l[2], l[3]), pgv_pack_bf16x2(l[4], l[5]), pgv_pack_bf16x2(l[6], l[7])};
}

// Split weight shadow of a LARGE-plane k4 s2 p2 layer (conv_big_split.hip), fragment order of down_q_kernel / up_q_kernel:
// DOWN layout [M tile of 16 small channels][K step ks = kh * (CB/8) + g][plane][lane]: lane (m, kq) feeds the matrix
// instruction, as the A operand of row cs = mt*16 + m, the 8 values (kernel column kw = 0..3) x (big channels 2p, 2p + 1) of
// channel pair p = 4g + kq at kernel row kh - the order in which the B window of the channel-pair planar activation image
// holds them (4 consecutive dwords = 4 input columns, a dword = the pair's two channels).  One item = (mt, ks, lane):
// CS * CB * 2 items.
// NPL = 3: the three planes of the exact split; NPL = 1 (bf16 operand mode): the weight rounded to nearest, one plane.
template <int NPL = 3>
__device__ __forceinline__ void shadow_bigq_down_item(int it, const float* __restrict__ w, int CS, int CB,
                                                      unsigned short* __restrict__ sh) {
  const int lane = it & 63, rest = it >> 6, ksteps = CB / 2, ng = CB / 8;
  const int ks = rest % ksteps, mt = rest / ksteps, kh = ks / ng, g = ks - kh * ng;
  const int cs = mt * 16 + (lane & 15), cp = 4 * g + (lane >> 4);
  float h[8], mid[8], l[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) pgv_split3(w[((size_t)cs * CB + 2 * cp + (c & 1)) * 16 + kh * 4 + (c >> 1)], h[c], mid[c], l[c]);
  u32x4* dst = reinterpret_cast<u32x4*>(sh) + ((size_t)rest * NPL) * 64 + lane;
  dst[0] = u32x4{pgv_pack_bf16x2(h[0], h[1]), pgv_pack_bf16x2(h[2], h[3]), pgv_pack_bf16x2(h[4], h[5]), pgv_pack_bf16x2(h[6], h[7])};
  if constexpr (NPL == 3) {
    dst[64] = u32x4{pgv_pack_bf16x2(mid[0], mid[1]), pgv_pack_bf16x2(mid[2], mid[3]), pgv_pack_bf16x2(mid[4], mid[5]), pgv_pack_bf16x2(mid[6], mid[7])};
    dst[128] = u32x4{pgv_pack_bf16x2(l[0], l[1]), pgv_pack_bf16x2(l[2], l[3]), pgv_pack_bf16x2(l[4], l[5]), pgv_pack_bf16x2(l[6], l[7])};
  }
}
// UP layout [M tile][K step g = group of 8 small channels][plane][lane], M rows r = phase * CB + cb (phase = 2 ph + pw of the
// output pixel): the 16 bytes (small channels g*8 .. +7 at the phase's tap kq = 2 th + tw, i.e. kernel tap (ph + 2 th, pw + 2 tw))
// of row r = mt*16 + m.  One item = (mt, g, lane): CS * CB * 2 items.
template <int NPL = 3>
__device__ __forceinline__ void shadow_bigq_up_item(int it, const float* __restrict__ w, int CS, int CB,
                                                    unsigned short* __restrict__ sh) {
  const int lane = it & 63, rest = it >> 6, ng = CS / 8, g = rest % ng, mt = rest / ng;
  const int r = mt * 16 + (lane & 15), kq = lane >> 4, phase = r / CB, cb = r - phase * CB;
  const int kh = (phase >> 1) + 2 * (kq >> 1), kw = (phase & 1) + 2 * (kq & 1);
  float h[8], mid[8], l[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) pgv_split3(w[((size_t)(g * 8 + c) * CB + cb) * 16 + kh * 4 + kw], h[c], mid[c], l[c]);
  u32x4* dst = reinterpret_cast<u32x4*>(sh) + ((size_t)rest * NPL) * 64 + lane;
  dst[0] = u32x4{pgv_pack_bf16x2(h[0], h[1]), pgv_pack_bf16x2(h[2], h[3]), pgv_pack_bf16x2(h[4], h[5]), pgv_pack_bf16x2(h[6], h[7])};
  if constexpr (NPL == 3) {
    dst[64] = u32x4{pgv_pack_bf16x2(mid[0], mid[1]), pgv_pack_bf16x2(mid[2], mid[3]), pgv_pack_bf16x2(mid[4], mid[5]), pgv_pack_bf16x2(mid[6], mid[7])};
    dst[128] = u32x4{pgv_pack_bf16x2(l[0], l[1]), pgv_pack_bf16x2(l[2], l[3]), pgv_pack_bf16x2(l[4], l[5]), pgv_pack_bf16x2(l[6], l[7])};
  }
}

}  // namespace

// timing aids shared by the bf16-native translation units (defined in conv_deep_bf16.hip)
unsigned long long* pgv_deep_bf16_stamps();   // device buffer of the timing scripts: clock64() at the phase marks, or null
int pgv_deep_bf16_dbg();                      // pgv_dbg_set_deep_bf16_variant
#define BSTAMP(k)                                                                                   \
  do {                                                                                              \
    if (stamps && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 77)) stamps[(blockIdx.x ? 16 : 0) + (k)] = clock64(); \
  } while (0)
