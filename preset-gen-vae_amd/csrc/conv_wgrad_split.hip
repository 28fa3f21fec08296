// Weight gradient of the LARGE-plane k4 s2 p2 layers with fp32 products as SIX bf16 matrix instructions
// (PGV_COMPUTE_F32_SPLIT; model/layer.py:10-46 under train.py:246; conv_big_split.hip has the forward / input-gradient forms):
//   gW[cs][cb][kh][kw] = sum_{b,oh,ow} S[b,cs,oh,ow] * X[b,cb,2oh-2+kh,2ow-2+kw]
// Both operands are activations, so both are split into three exact bfloat16 planes on their way into LDS.  GEMM view (the
// formulation of conv_wgrad_bf16.hip): M = cs, N = (cb, kh, kw), K = output pixels, eight consecutive pixels of a row per
// lane and instruction.  The stride-2 column walk is contiguous in the EVEN / ODD column planes of X (Xp[j] = X[2j + p]):
//   kw = 2 + p:  column 2 ow + p      -> sum_j S[j]     Xp[j]
//   kw = p:      column 2 (ow-1) + p  -> sum_j S[j + 1] Xp[j]      (the A fragment one pixel on: a 16-byte read + the next
//                                                                   dword, moved into place by four v_alignbit_b32 - a
//                                                                   2-byte misaligned ds_read_b128 takes 64 LDS clocks)
// so a column tile is (two big channels) x (4 kernel rows) x (2 parities) of ONE half h of the kernel columns, and a wave owns
// column tiles of one half only (it needs one kind of A fragment).
// Structure as in conv_big_split.hip: one 512-thread workgroup per CU, persistent over its units (sample, band of R output
// rows); per unit a matrix phase (all waves multiply: MT x CTW tiles each, the K steps of a unit dealt over KWAYS waves)
// and a vector phase (the next unit's two bands are split and committed), one barrier each; the band loads travel in
// registers across the matrix phase.  The accumulators stay in registers over all units; at the end the K ways are added up
// through LDS and the workgroup's partial gradient goes to the workspace (the reduce launch of conv_v2_wgrad.hip adds the
// workgroups up, with the tap-sum / bias roles of the step).
#include "conv_tile.h"
#include "conv_deep_common.h"

#ifdef PGV_BIGQ_STAMPS   // scratch builds only (scratch/wgq_stamps.py): clock64() of workgroup 0, wave 0
static unsigned long long* g_wgq_stamps = nullptr;
extern "C" void pgv_dbg_set_wgq_stamps(void* p) { g_wgq_stamps = (unsigned long long*)p; }
#define WSTAMP_ARG , unsigned long long* __restrict__ stamps
#define WSTAMP_PASS , g_wgq_stamps
#define WSTAMP(k)                                                                                  \
  do {                                                                                             \
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0 && (k) < 64) stamps[k] = clock64();          \
  } while (0)
#else
#define WSTAMP_ARG
#define WSTAMP_PASS
#define WSTAMP(k) \
  do {            \
  } while (0)
#endif

namespace {

typedef unsigned short u16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tensor_rsrc(const float* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f4u buffer_load_x4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(f4u, v);
}
__device__ __forceinline__ void ws_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ constexpr int kTermA[6] = {0, 2, 1, 0, 1, 0}, kTermB[6] = {2, 0, 1, 1, 0, 0};   // (plane of a, plane of b), smallest first

// strides in bf16 elements from scratch/wgrad_split_strides.py (conflict-free ds_read_b128 groups where the shape allows)
// NPL = operand planes: 3 = six-instruction fp32 products on exact three-way splits, 1 = bf16 operand mode (PGV_COMPUTE_BF16:
// both operands rounded to nearest where they are committed, one instruction per fragment pair) - as in conv_big_split.hip
template <int CB_, int CS_, int W_, int H_, int R_, int SCH_, int XCH_, int XPL_, int NQ_, int KWAYS_, int NPL_ = 3>
struct WgQ {
  static constexpr int CB = CB_, CS = CS_, W = W_, H = H_, R = R_, SCH = SCH_, XCH = XCH_, XPL = XPL_, NQ = NQ_, KWAYS = KWAYS_;
  static constexpr int NPL = NPL_, NTERM = NPL_ == 3 ? 6 : 1;
  static_assert(NPL_ == 3 || NPL_ == 1, "three planes (six product terms) or one");
  static constexpr int Ws = W / 2 + 1, Hs = H / 2 + 1, BANDS = (Hs + R - 1) / R, XR = 2 * R + 2;
  static constexpr int GPR = (Ws + 7) / 8, XROW = GPR * 8, SROW = GPR * 8 + 8;   // 8-pixel groups per row; row strides (S: + 8 zeros)
  static constexpr int NGRP = R * GPR, KS = (NGRP + 3) / 4, KSW = (KS + KWAYS - 1) / KWAYS;   // K steps of a unit / of a wave
  static constexpr int MT = CS / 16, CTW = (CB / 2) / NQ;      // M tiles (all of them per wave); column tiles of a wave
  static constexpr int S_PLANE = CS * SCH, X_PLANE = 2 * XPL;   // elements of one plane image
  static constexpr int OX = (W + 7) / 8, X_ITEMS = CB * XR * OX, S_ITEMS = CS * R * GPR;
  static constexpr int QX = (X_ITEMS + 511) / 512, QS = (S_ITEMS + 511) / 512;
  static constexpr size_t IMG_BYTES = NPL * 2 * (size_t)(S_PLANE + X_PLANE);
  static constexpr size_t LDS_BYTES = IMG_BYTES + sizeof(float) * 2 * (CB + CS);
  static_assert(2 * NQ * KWAYS == 8 && (CB / 2) % NQ == 0 && CS % 16 == 0, "eight waves: kernel-column half x column groups x K ways");
  static_assert(SCH >= R * SROW + 8 && XCH >= XR * XROW && XPL >= CB * XCH && SCH % 8 == 0 && XCH % 8 == 0 && XPL % 8 == 0, "plane strides");
  static_assert(4 * OX <= XROW && LDS_BYTES <= 160 * 1024, "tile shapes / LDS budget");
  static_assert(KWAYS == 1 || (size_t)(KWAYS - 1) * 2 * NQ * MT * CTW * 1024 <= IMG_BYTES, "K-way reduction fits the images");
};

template <class G, bool BIG_AFF, bool SMALL_AFF>
__global__ __launch_bounds__(512) void conv_wgrad_split_kernel(int B, const float* __restrict__ big, const float* __restrict__ big_scale,
                                                              const float* __restrict__ big_shift, const float* __restrict__ small_in,
                                                              const float* __restrict__ small_scale,
                                                              const float* __restrict__ small_shift, float* __restrict__ partial WSTAMP_ARG) {
  constexpr int CB = G::CB, CS = G::CS, W = G::W, H = G::H, R = G::R, Ws = G::Ws, Hs = G::Hs, MT = G::MT, CTW = G::CTW;
  constexpr int SCH = G::SCH, XCH = G::XCH, XPL = G::XPL, SROW = G::SROW, XROW = G::XROW, GPR = G::GPR, KSW = G::KSW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NPL = G::NPL, NTERM = G::NTERM;
  u16* s_img = reinterpret_cast<u16*>(lds_raw);                     // [NPL][CS][SCH]
  u16* x_img = s_img + NPL * G::S_PLANE;                            // [NPL][2 parities][XPL]
  float* aff_b = reinterpret_cast<float*>(lds_raw + G::IMG_BYTES);   // [2][CB]
  float* aff_s = aff_b + 2 * CB;                                    // [2][CS]
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave & 1, nq = (wave >> 1) % G::NQ, kw0 = (wave >> 1) / G::NQ;   // kernel-column half, column group, K way
  const pgv_split_sel sel = pgv_split_sel_make();

  WSTAMP(0);
  // ---- loader items.  X: (channel, band row, 8 columns) -> 4 even + 4 odd columns = 8 bytes per parity and plane;
  // S: (channel, band row, 8 pixels) = 16 bytes per plane
  int xl_src[G::QX], xl_cr[G::QX], xl_row[G::QX], sl_src[G::QS], sl_cr[G::QS], sl_row[G::QS];
#pragma unroll
  for (int i = 0; i < G::QX; ++i) {
    const int q = min(tid + 512 * i, G::X_ITEMS - 1), c = q / (G::XR * G::OX), rem = q - c * (G::XR * G::OX), r = rem / G::OX, o = rem - r * G::OX;
    xl_src[i] = c * (H * W) + 8 * o;                    // + sample * CB * H * W + image row * W
    xl_row[i] = r * W;
    xl_cr[i] = (c << 16) | (r << 8) | o | ((tid + 512 * i < G::X_ITEMS) ? 0x8000 : 0);
  }
#pragma unroll
  for (int i = 0; i < G::QS; ++i) {
    const int q = min(tid + 512 * i, G::S_ITEMS - 1), c = q / (R * GPR), rem = q - c * (R * GPR), r = rem / GPR, o = rem - r * GPR;
    sl_src[i] = c * (Hs * Ws) + 8 * o;                  // + sample * CS * Hs * Ws + output row * Ws
    sl_row[i] = r * Ws;
    sl_cr[i] = (c << 16) | (r << 8) | o | ((tid + 512 * i < G::S_ITEMS) ? 0x8000 : 0);
  }
  const int grid = (int)gridDim.x, u0 = pgv_xcd_block(), units = B * G::BANDS;
  const int J = u0 < units ? (units - 1 - u0) / grid + 1 : 0;

  f4u xb[G::QX][2], sb[G::QS][2];
  const __amdgpu_buffer_rsrc_t big_rs = tensor_rsrc(big, (int64_t)B * CB * (H * W) * 4);
  const __amdgpu_buffer_rsrc_t small_rs = tensor_rsrc(small_in, (int64_t)B * CS * (Hs * Ws) * 4);
  // The band loads are software-pipelined ITEM BY ITEM through the vector phase: an item of unit j + 1 is converted and
  // committed, and its registers are re-issued at once for the same item of unit j + 2 - loads stay in flight through the
  // vector phase and the next matrix phase (issued in one burst at the start of the matrix phase, a unit's 78 KB were
  // still arriving when the vector phase wanted them: a CU's fair share of HBM moves them in ~6 k clocks, the matrix
  // phase lasts ~3.5 k).  A load for a unit beyond this workgroup's last one gets an offset beyond the buffer: no traffic.
  struct UnitPos { unsigned xs, ss, kill; int band; };
  auto unit_pos = [&](int j) {
    const int u = min(u0 + j * grid, units - 1), b = u / G::BANDS;
    return UnitPos{(unsigned)b * (unsigned)(CB * H * W), (unsigned)b * (unsigned)(CS * Hs * Ws), j < J ? 0u : 0x80000000u, u - b * G::BANDS};
  };
  auto issue_x = [&](int i, const UnitPos& up) {
    // (row offset = scalar band part + the item's constant; a row outside the plane reads outside the buffer)
    const int ih0 = 2 * up.band * R - 2;
    const unsigned rk = (unsigned)(ih0 + ((xl_cr[i] >> 8) & 63)) < (unsigned)H ? up.kill : 0x80000000u;
    const unsigned o = ((up.xs + (unsigned)(xl_src[i] + xl_row[i] + ih0 * W)) * 4u) | rk;
    xb[i][0] = buffer_load_x4(big_rs, o);        // (columns beyond the row: masked at the commit)
    xb[i][1] = buffer_load_x4(big_rs, o + 16u);
  };
  auto issue_s = [&](int i, const UnitPos& up) {
    const unsigned rk = up.band * R + ((sl_cr[i] >> 8) & 63) < Hs ? up.kill : 0x80000000u;
    const unsigned o = ((up.ss + (unsigned)(sl_src[i] + sl_row[i] + up.band * (R * Ws))) * 4u) | rk;
    sb[i][0] = buffer_load_x4(small_rs, o);
    sb[i][1] = buffer_load_x4(small_rs, o + 16u);
  };
  auto commit_x = [&](int i, const UnitPos& up, float a_sc, float a_sh) {
    const int c = xl_cr[i] >> 16, r = (xl_cr[i] >> 8) & 63, o = xl_cr[i] & 255;
    const float mk = (unsigned)(2 * up.band * R - 2 + r) < (unsigned)H ? 1.f : 0.f, sc = a_sc, sh = a_sh * mk;
    if (xl_cr[i] & 0x8000) {   // (kept under the lane mask for full slots too: without it the kernel lost 2 - 4 us)
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? xb[i][0][e] : xb[i][1][e - 4];
        // (columns beyond the row stay zero under an affine too; an operand without an affine is taken as it arrived -
        // rows outside the plane arrived as zeros)
        y[e] = 8 * o + e < W ? (BIG_AFF ? fmaf(v, sc, sh) : v) : 0.f;
      }
      unsigned e1[2], e2[2] = {0, 0}, e3[2] = {0, 0}, o1[2], o2[2] = {0, 0}, o3[2] = {0, 0};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if constexpr (NPL == 3) {
          pgv_split3_pair(y[4 * k], y[4 * k + 2], e1[k], e2[k], e3[k], sel);       // even columns 8o + 4k, + 2
          pgv_split3_pair(y[4 * k + 1], y[4 * k + 3], o1[k], o2[k], o3[k], sel);   // odd columns
        } else {   // bf16 operand mode: rounded to nearest, one plane
          e1[k] = pgv_pack_bf16x2(y[4 * k], y[4 * k + 2]);
          o1[k] = pgv_pack_bf16x2(y[4 * k + 1], y[4 * k + 3]);
        }
      }
      u16* dst = x_img + c * XCH + r * XROW + 4 * o;
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2*>(dst) = u32x2{e1[0], e1[1]};
      *reinterpret_cast<u32x2*>(dst + XPL) = u32x2{o1[0], o1[1]};
      if constexpr (NPL == 3) {
        *reinterpret_cast<u32x2*>(dst + G::X_PLANE) = u32x2{e2[0], e2[1]};
        *reinterpret_cast<u32x2*>(dst + G::X_PLANE + XPL) = u32x2{o2[0], o2[1]};
        *reinterpret_cast<u32x2*>(dst + 2 * G::X_PLANE) = u32x2{e3[0], e3[1]};
        *reinterpret_cast<u32x2*>(dst + 2 * G::X_PLANE + XPL) = u32x2{o3[0], o3[1]};
      }
    }
  };
  auto commit_s = [&](int i, const UnitPos& up, float a_sc, float a_sh) {
    const int c = sl_cr[i] >> 16, r = (sl_cr[i] >> 8) & 63, o = sl_cr[i] & 255;
    const float mk = up.band * R + r < Hs ? 1.f : 0.f, sc = a_sc, sh = a_sh * mk;
    if (sl_cr[i] & 0x8000) {
      u32x4 p1, p2, p3;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v0 = k < 2 ? sb[i][0][2 * k] : sb[i][1][2 * k - 4], v1 = k < 2 ? sb[i][0][2 * k + 1] : sb[i][1][2 * k - 3];
        const float y0 = 8 * o + 2 * k < Ws ? (SMALL_AFF ? fmaf(v0, sc, sh) : v0) : 0.f;
        const float y1 = 8 * o + 2 * k + 1 < Ws ? (SMALL_AFF ? fmaf(v1, sc, sh) : v1) : 0.f;
        unsigned a1, a2 = 0, a3 = 0;
        if constexpr (NPL == 3)
          pgv_split3_pair(y0, y1, a1, a2, a3, sel);
        else
          a1 = pgv_pack_bf16x2(y0, y1);
        p1[k] = a1, p2[k] = a2, p3[k] = a3;
      }
      u16* dst = s_img + c * SCH + r * SROW + 8 * o;
      *reinterpret_cast<u32x4*>(dst) = p1;
      if constexpr (NPL == 3) {
        *reinterpret_cast<u32x4*>(dst + G::S_PLANE) = p2;
        *reinterpret_cast<u32x4*>(dst + 2 * G::S_PLANE) = p3;
      }
    }
  };
  // commit unit jc from the registers, re-issue them for unit jc + 1
  auto vector_phase = [&](int jc) {
    const UnitPos uc = unit_pos(jc), un = unit_pos(jc + 1);
    // (the items' affines are read from LDS before the first commit: read inside an item, the in-order lgkmcnt wait for them
    // was also a wait for the previous item's image stores)
    float axs[G::QX][2], ass[G::QS][2];
#pragma unroll
    for (int i = 0; i < G::QX; ++i) axs[i][0] = aff_b[xl_cr[i] >> 16], axs[i][1] = aff_b[CB + (xl_cr[i] >> 16)];
#pragma unroll
    for (int i = 0; i < G::QS; ++i) ass[i][0] = aff_s[sl_cr[i] >> 16], ass[i][1] = aff_s[CS + (sl_cr[i] >> 16)];
#pragma unroll
    for (int i = 0; i < G::QX; ++i) {
      commit_x(i, uc, axs[i][0], axs[i][1]);
      issue_x(i, un);
    }
#pragma unroll
    for (int i = 0; i < G::QS; ++i) {
      commit_s(i, uc, ass[i][0], ass[i][1]);
      issue_s(i, un);
    }
  };

  // ---- fragment bases (elements) of this wave's K steps: lane group kq of step ks holds pixel group gi = 4 ks + kq
  int offA[KSW], offB[KSW];
#pragma unroll
  for (int k = 0; k < KSW; ++k) {
    const int ks = kw0 + G::KWAYS * k, gi = min(4 * ks + kq, G::NGRP - 1), row = gi / GPR, g8 = gi - row * GPR;
    // (a lane group beyond the unit's last pixel group repeats the last group of the LAST row with an A offset into the
    // zero padding behind it: its products vanish)
    const bool pad = 4 * ks + kq >= G::NGRP;
    offA[k] = n * SCH + row * SROW + (pad ? GPR * 8 : g8 * 8);
    offB[k] = (n & 1) * XPL + (2 * nq * CTW + (n >> 3)) * XCH + (2 * row + ((n >> 1) & 3)) * XROW + g8 * 8;
  }
  f32x4 acc[MT][CTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < CTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (J > 0) {
    const UnitPos u0p = unit_pos(0);
#pragma unroll
    for (int i = 0; i < G::QX; ++i) issue_x(i, u0p);
#pragma unroll
    for (int i = 0; i < G::QS; ++i) issue_s(i, u0p);
  }
  // (the images are cleared and the affines staged behind the first unit's loads: their latency covers it)
  for (int i = tid; i < (int)(G::IMG_BYTES / 16); i += 512) reinterpret_cast<u32x4*>(lds_raw)[i] = u32x4{0, 0, 0, 0};
  for (int i = tid; i < CB; i += 512) {
    aff_b[i] = BIG_AFF ? big_scale[i] : 1.f;
    aff_b[CB + i] = BIG_AFF ? big_shift[i] : 0.f;
  }
  for (int i = tid; i < CS; i += 512) {
    aff_s[i] = SMALL_AFF ? small_scale[i] : 1.f;
    aff_s[CS + i] = SMALL_AFF ? small_shift[i] : 0.f;
  }
  __syncthreads();
  if (J > 0) vector_phase(0);
  __syncthreads();

  WSTAMP(1);
#pragma unroll 1
  for (int j = 0; j < J; ++j) {
    WSTAMP(2 + 3 * j);
    // ================= matrix phase ================= (the loads of unit j + 1 are in flight)
    // K steps whose four pixel groups all lie in rows beyond the plane (a sample's last band: 1 valid row of R at all three
    // sizes) are skipped - 2 of 3 steps of such a unit on 33x45, where they were 13 % of the kernel's matrix instructions
    const int nvalid_groups = min(R, Hs - (min(u0 + j * grid, units - 1) % G::BANDS) * R) * GPR;
#pragma unroll
    for (int k = 0; k < KSW; ++k) {
      if (k == 0 || 4 * (kw0 + G::KWAYS * k) < nvalid_groups) {   // (wave-uniform; the steps of a wave are in row order)
        u32x4 a[MT][NPL];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            const u16* ap = s_img + p * G::S_PLANE + offA[k] + m * 16 * SCH;
            const u32x4 w = *reinterpret_cast<const u32x4*>(ap);
            if (h) {   // kernel columns 2, 3: S itself
              a[m][p] = w;
            } else {   // kernel columns 0, 1: S one pixel on
              const unsigned w4 = *reinterpret_cast<const unsigned*>(ap + 8);
              a[m][p] = u32x4{__builtin_amdgcn_alignbit(w[1], w[0], 16), __builtin_amdgcn_alignbit(w[2], w[1], 16),
                              __builtin_amdgcn_alignbit(w[3], w[2], 16), __builtin_amdgcn_alignbit(w4, w[3], 16)};
            }
          }
#pragma unroll
        for (int t = 0; t < CTW; ++t) {
          u32x4 bfr[NPL];
#pragma unroll
          for (int p = 0; p < NPL; ++p) bfr[p] = *reinterpret_cast<const u32x4*>(x_img + p * G::X_PLANE + offB[k] + 2 * t * XCH);
#pragma unroll
          for (int term = 0; term < NTERM; ++term) {
            const int pa = NPL == 3 ? kTermA[term] : 0, pb = NPL == 3 ? kTermB[term] : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][t] = mfma_bf16_k32(a[m][pa], bfr[pb], acc[m][t]);
          }
        }
      }
    }
    WSTAMP(3 + 3 * j);
    ws_sync();
    WSTAMP(4 + 3 * j);
    // ================= vector phase =================
    if (j + 1 < J) vector_phase(j + 1);
    ws_sync();
  }
  WSTAMP(60);

  // ---- the K ways of a tile are added up through LDS (the images are dead), then this workgroup's partial gradient, layout
  // of gw: D row (lane >> 4) * 4 + reg = cs within the M tile, column n = (channel of the pair, kernel row, parity)
  if (G::KWAYS > 1) {
    f32x4* red = reinterpret_cast<f32x4*>(lds_raw);
    const int slot0 = ((h * G::NQ + nq) * MT * CTW) * 64 + lane;
    if (kw0 > 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < CTW; ++t) red[(size_t)(kw0 - 1) * (2 * G::NQ * MT * CTW * 64) + slot0 + (m * CTW + t) * 64] = acc[m][t];
    }
    ws_sync();
    if (kw0 == 0) {
#pragma unroll
      for (int kx = 1; kx < G::KWAYS; ++kx)   // (fixed order)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < CTW; ++t) acc[m][t] += red[(size_t)(kx - 1) * (2 * G::NQ * MT * CTW * 64) + slot0 + (m * CTW + t) * 64];
    }
  }
  if (kw0 == 0) {
    float* pw = partial + (size_t)blockIdx.x * ((size_t)CS * CB * 16);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < CTW; ++t) {
        const int cb = 2 * (nq * CTW + t) + (n >> 3), kh = (n >> 1) & 3, kw = 2 * h + (n & 1);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int cs = m * 16 + kq * 4 + reg;
          pw[(cs * CB + cb) * 16 + kh * 4 + kw] = acc[m][t][reg];
        }
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WSTAMP(61);
}

template <class G>
int launch_wgq(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift, const float* small_in,
               const float* small_scale, const float* small_shift, float* partial, int64_t partial_bytes, int* nparts,
               hipStream_t st) {
  if (d->Cb != G::CB || d->Cs != G::CS || d->B <= 0) return 0;
  if (big_scale && small_scale) return 0;   // not a case of the train step
  if ((int64_t)d->B * d->Cb * G::H * G::W * 4 >= (int64_t)1 << 31 || (int64_t)d->B * d->Cs * G::Hs * G::Ws * 4 >= (int64_t)1 << 31) return 0;
  const int units = d->B * G::BANDS, grid = min(units, 256);
  if ((int64_t)grid * G::CS * G::CB * 16 * (int64_t)sizeof(float) > partial_bytes) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, const float*, const float*, float* WSTAMP_ARG);
  kern_t kern = big_scale ? (kern_t)conv_wgrad_split_kernel<G, true, false>
                          : (small_scale ? (kern_t)conv_wgrad_split_kernel<G, false, true>
                                         : (kern_t)conv_wgrad_split_kernel<G, false, false>);
  static const void* raised[3];
  const int slot = big_scale ? 0 : (small_scale ? 1 : 2);
  if (raised[slot] != (const void*)kern) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
      pgv_set_error("conv_wgrad_split: cannot raise the dynamic LDS limit");
      return PGV_E_LAUNCH;
    }
    raised[slot] = (const void*)kern;
  }
  *nparts = grid;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS_BYTES, st, d->B, big, big_scale, big_shift, small_in, small_scale, small_shift,
                     partial WSTAMP_PASS);
  PGV_CHECK_LAUNCH("conv_wgrad_split");
  return 1;
}

}  // namespace

// 1 = launched (*nparts partial gradients in ``partial``, one per workgroup, layout of gw), 0 = shape / mode not covered
int pgv_conv_wgrad_split_partial(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                                 const float* small_in, const float* small_scale, const float* small_shift, float* partial,
                                 int64_t partial_bytes, int* nparts, hipStream_t st) {
  if (!partial) return 0;
  if (pgv_big_bf16q_shape(d)) {   // bf16 operand mode: one plane per operand, the same tilings
    if (d->Hb == 129)
      return launch_wgq<WgQ<8, 16, 174, 129, 4, 400, 896, 7232, 1, 4, 1>>(d, big, big_scale, big_shift, small_in, small_scale,
                                                                          small_shift, partial, partial_bytes, nparts, st);
    if (d->Hb == 65)
      return launch_wgq<WgQ<16, 32, 88, 65, 4, 240, 512, 8256, 2, 2, 1>>(d, big, big_scale, big_shift, small_in, small_scale,
                                                                         small_shift, partial, partial_bytes, nparts, st);
    return launch_wgq<WgQ<32, 64, 45, 33, 4, 144, 256, 8256, 4, 1, 1>>(d, big, big_scale, big_shift, small_in, small_scale,
                                                                       small_shift, partial, partial_bytes, nparts, st);
  }
  if (!pgv_big_split_shape(d)) return 0;
  if (d->Hb == 129)
    return launch_wgq<WgQ<8, 16, 174, 129, 4, 400, 896, 7232, 1, 4>>(d, big, big_scale, big_shift, small_in, small_scale, small_shift,
                                                                     partial, partial_bytes, nparts, st);
  if (d->Hb == 65)
    return launch_wgq<WgQ<16, 32, 88, 65, 4, 240, 512, 8256, 2, 2>>(d, big, big_scale, big_shift, small_in, small_scale, small_shift,
                                                                    partial, partial_bytes, nparts, st);
  return launch_wgq<WgQ<32, 64, 45, 33, 4, 144, 256, 8256, 4, 1>>(d, big, big_scale, big_shift, small_in, small_scale, small_shift,
                                                                  partial, partial_bytes, nparts, st);
}
