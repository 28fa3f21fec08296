// Wave-specialised DOWN kernel (Conv2d forward / ConvTranspose2d input gradient) of the stride-2 k=4 layers at the
// reference sizes; structure and measurements: conv_v2_common.h, DESIGN.md section 3.4.
#define PGV_V2_TU down
#include "conv_v2_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// DOWN (Conv2d forward / ConvTranspose2d input-gradient), k = 4, stride 2, pad 2:
//   D[cs][pixel] = sum_{c,kh,kw} W[cs][c][kh][kw] * X[c][2r+kh-2][2col+kw-2]
// Unit = R output rows of one sample; waves = MW (M groups of MTW tiles) x NW (pixel-tile groups of NT tiles); CK input
// channels per LDS chunk; work items (unit, chunk) of a workgroup form ONE pipeline:
//   step loop of item i  ||  commit of item i+1 (first half of the steps)  ||  global loads of item i+2 (second half)
// Every k-step is one scheduling region: NT x MTW MFMAs interleaved 1 : 1 with the ds_read_b32 of step + 2 (three
// rotating operand sets; measured 34.5 clk per MFMA against 53.7 for the compiler's own order, scratch/ubench/v2_loop.hip).
// ---------------------------------------------------------------------------------------------------------------
template <int CB, int CS, int W, int H, int R, int MW, int CK>
struct DownV2Cfg {
  static constexpr int KS = 4;
  static constexpr int Ws = W / 2 + 1, Hs = H / 2 + 1;
  static constexpr int BANDS = (Hs + R - 1) / R;
  static constexpr int NW = 4 / MW;
  static constexpr int MTT = CS / 16, MTW = MTT / MW;
  static constexpr int P = R * Ws;
  static constexpr int NTT = (P + 15) / 16, NT = (NTT + NW - 1) / NW;
  static constexpr int ROWS = 2 * (R - 1) + KS;
  static constexpr int WP = (W + 2 + 3) / 4 * 4;
  static constexpr int PLANE = ROWS * WP;
  static constexpr int NCH = CB / CK;
  static constexpr int S = CK * KS;
  static constexpr int FRONT = 4;
  static constexpr int BUF = CK * PLANE;
  static constexpr size_t LDS_FLOATS = FRONT + 2 * (size_t)BUF + 2 * CB;
  static_assert(CS % 16 == 0 && MTT % MW == 0 && CB % CK == 0 && 4 % MW == 0, "tiling");
  static_assert(2 * (Ws - 1) + KS - 3 < WP, "row stride");
  static_assert(S % 2 == 0 && S >= 4, "k-steps");
};

// ---------------------------------------------------------------------------------------------------------------
// DOWN, wave-specialised form: 512 threads = 4 MFMA waves (one per SIMD) + 4 loader waves (their SIMD partners).
//   MFMA waves:   k-steps (ds_read_b32 + MFMA only, pinned 1 : 1), weights from global memory, epilogue.
//   loader waves: global -> registers -> LDS staging of the channel chunks, TWO items ahead of the multiplication (two
//                 register sets), producer's BatchNorm affine applied on the way.
// One workgroup barrier per item joins the two roles (LDS double buffer).  The MFMA stream carries no staging
// instructions (the sliced single-role kernel above spends 38-44 clk per MFMA in its k-steps, the bare loop 34.5), and
// the loader is ordinary code - loops, branches, no scheduling pragmas.
// ---------------------------------------------------------------------------------------------------------------
// STG: lean loader (StageLean, row tails cleared by the loader) + deferred stores, as in conv_up_ws_kernel: for the
// 129x174 layer, whose StageV2 loader co-limits it and whose output leaves in bursts.
// Two workgroups per CU (V2_DOWN_WPS = 4 waves per SIMD): units of R = 5 rows whose LDS double buffer is half the size.
// Experiment of round 4: two INDEPENDENT wave-specialised workgroups on a CU could fill each other's
// prologue / epilogue / barrier bubbles on the matrix pipe.
template <int R, int W>
constexpr int V2_DOWN_WPS = (W == 88 && R == 5) ? 4 : 2;

// BF16 (STG forms only): PGV_COMPUTE_BF16 - both operands rounded to bfloat16, the input where the lean loader commits
// it to LDS, the weights where they are loaded; fp32 MFMA (products of bf16 values are exact in fp32).
template <int CB, int CS, int W, int H, int R, int MW, int CK, bool FUSE, bool HAS_AFF, int ACT, bool STG = false,
          bool BF16 = false>
__global__ __launch_bounds__(512, (V2_DOWN_WPS<R, W>)) void conv_down_ws_kernel(int B, const float* __restrict__ big,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            int act, float slope, float* __restrict__ out,
                                                            double* __restrict__ stats, pgv_bwd_fuse fuse,
                                                            pgv_bn_src bn, int stat_copies) {
  using G = DownV2Cfg<CB, CS, W, H, R, MW, CK>;
  constexpr int Ws = G::Ws, Hs = G::Hs, BANDS = G::BANDS, NW = G::NW, MTW = G::MTW, P = G::P, NT = G::NT;
  constexpr int WP = G::WP, PLANE = G::PLANE, NCH = G::NCH, S = G::S, BUF = G::BUF;
  using Stage = StageV2<CK, G::ROWS, W, WP, H>;
  constexpr int NPF = Stage::NPF;
  static_assert(!STG || (NCH == 1 && ACT != 2 && Ws % 4 == 0 && S >= MTW * NT), "deferred stores");
  // STG + FUSE ("APRE"): the fused backward epilogue (pgv_bwd_fuse) with the unit's saved activation prefetched into
  // registers by the MFMA waves, one tile per k-step next to the deferred stores of the previous unit (see
  // conv_up_ws_kernel); the class sums of the result (pgv_bwd_fuse.cls) ride along.  Plain products only.
  constexpr bool APRE = STG && FUSE;
  static_assert(!APRE || (!HAS_AFF && ACT == 0), "fused backward epilogue: plain input-gradient products");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile0 = lds + G::FRONT;
  float* aff = tile0 + 2 * BUF;  // [2][CB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int units = B * BANDS;
  const int bid = pgv_xcd_block();  // first unit of this workgroup
  const int my_units = bid < units ? (units - bid + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int my_items = my_units * NCH;
  if (my_items == 0) return;
  if (tid < G::FRONT) lds[tid] = 0.f;
  for (int i = tid; i < CB; i += 512) {
    float sc = 1.f, sh = 0.f;
    // (pgv_conv_down_bn: the producer's BatchNorm is finalized here, from its statistics, instead of by a launch of its own)
    if (HAS_AFF && bn.stats)
      pgv_bn_finalize_dev(bn, CB, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[CB + i] = sh;
  }
  __syncthreads();
  // global source of local item `it` (clamped to the last one: the loader runs ahead unconditionally)
  auto item_src = [&](int it, const float*& plane0, int& ih0) {
    it = min(it, my_items - 1);
    const int u = bid + (it / NCH) * gridDim.x, ch = it % NCH;
    const int b = u / BANDS, band = u - b * BANDS;
    const uint64_t p = (uint64_t)(big + ((int64_t)b * CB + ch * CK) * (H * W));
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    plane0 = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);  // provably wave-uniform: an SGPR pair
    ih0 = band * R * 2 - 2;
  };

  if (STG && wave >= 4) {
    // ======================================= loader waves, lean form (see StageLean) =====================================
    using Lean = StageLean<CK, G::ROWS, W, WP, H, true>;
    constexpr int NL = Lean::NPF;
    const int ltid = tid - 256;
    __builtin_amdgcn_s_setprio(PGV_V2_PRIO_LOADER);
    typename Lean::Geo geo;
    typename Lean::Set sA, sB;
    static_assert(!STG || (BANDS >= 3 && (BANDS - 2) * 2 * R - 2 + G::ROWS <= H), "edge bands");
    const int64_t bytes_in = (int64_t)B * CB * (H * W) * 4;
    auto item_geo = [&](int it, i32x4& rs, unsigned& bad) {
      it = min(it, my_items - 1);
      const int u = bid + it * gridDim.x;
      const int b = u / BANDS, band = u - b * BANDS;
      rs = Lean::band_rsrc(big, bytes_in, ((int64_t)b * CB * H + band * 2 * R - 2) * W);
      bad = band == 0 ? geo.top_bad : (band == BANDS - 1 ? geo.bot_bad : 0u);
    };
    geo.init(ltid, aff, CB, HAS_AFF, 2, H - ((BANDS - 1) * 2 * R - 2), [&](auto jc) {
      i32x4 rs;
      unsigned bad;
      item_geo(0, rs, bad);
      Lean::template issue_slot<decltype(jc)::value, true>(geo, sA, rs, bad);
    });
    auto issue_all = [&](typename Lean::Set& sx, int it) {
      i32x4 rs;
      unsigned bad;
      item_geo(it, rs, bad);
      static_for<0, NL>([&](auto j) { Lean::template issue_slot<decltype(j)::value, true>(geo, sx, rs, bad); });
    };
    auto commit_all = [&](const typename Lean::Set& sx, int it, float* dst) {
      i32x4 rs;
      unsigned bad;
      item_geo(it, rs, bad);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");  // the older set has landed
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NL>([&](auto j) { Lean::template commit_slot<decltype(j)::value, true, HAS_AFF, BF16>(geo, sx, dst, ltid, bad); });
    };
    issue_all(sB, 1);  // (item 0 went out during the set-up)
    commit_all(sA, 0, tile0);
    issue_all(sA, 2);
    ws_barrier();
#pragma unroll 1
    for (int it = 0; it < my_items; it += 2) {
      __builtin_amdgcn_s_sleep(PGV_V2_LOADER_SLEEP);
      commit_all(sB, it + 1, tile0 + BUF);
      issue_all(sB, it + 3);
      ws_barrier();
      if (it + 1 < my_items) {
        __builtin_amdgcn_s_sleep(PGV_V2_LOADER_SLEEP);
        commit_all(sA, it + 2, tile0);
        issue_all(sA, it + 4);
        ws_barrier();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  if (wave >= 4) {
    // ================================================= loader waves =================================================
    const int ltid = tid - 256;
    __builtin_amdgcn_s_setprio(PGV_V2_PRIO_LOADER);
    typename Stage::Geo geo;
    typename Stage::Set sA, sB;
    sA.live = sB.live = 0;
    {   // slot geometry, with item 0's loads going out slot by slot as their constants become ready
      const float* p0;
      int ih0;
      item_src(0, p0, ih0);
      geo.init(ltid, [&](auto jc) { Stage::template issue_slot<decltype(jc)::value>(geo, sA, p0, ih0); });
    }
    auto issue_all = [&](typename Stage::Set& sx, int it) {
      const float* p0;
      int ih0;
      item_src(it, p0, ih0);
      static_for<0, NPF>([&](auto j) { Stage::template issue_slot<decltype(j)::value>(geo, sx, p0, ih0); });
    };
    float sc[NPF], sh[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) sc[j] = 1.f, sh[j] = 0.f;
    if constexpr (HAS_AFF && NCH == 1) Stage::load_affine(geo, aff, CB, 0, sc, sh);  // one chunk: constant per slot
    auto commit_all = [&](const typename Stage::Set& sx, int it, float* dst) {
      if constexpr (HAS_AFF && NCH > 1) Stage::load_affine(geo, aff, CB, (min(it, my_items - 1) % NCH) * CK, sc, sh);
      Stage::wait_set();
      static_for<0, NPF>([&](auto j) {
        constexpr int J = decltype(j)::value;
        Stage::template commit_slot<J>(geo, sx, dst, ltid, HAS_AFF, sc[J], sh[J]);
      });
    };
    // Pipeline: item n lives in register set n & 1 and LDS buffer n & 1; loads are issued two items ahead of their
    // commit; ALWAYS exactly one older and one newer set are in flight when a commit starts (wait_set).
    issue_all(sB, 1);  // (item 0 went out during the set-up)
    commit_all(sA, 0, tile0);
    issue_all(sA, 2);
    V2_T0();
    ws_barrier();  // item 0 committed; the MFMA waves start
#pragma unroll 1
    for (int it = 0; it < my_items; it += 2) {
      V2_ACC(2);
      __builtin_amdgcn_s_sleep(PGV_V2_LOADER_SLEEP);  // let the MFMA waves' first operand reads of the item go first
      V2_ACC(3);
      commit_all(sB, it + 1, tile0 + BUF);  // item it+1 -> buffer 1 while item it is multiplied from buffer 0
      V2_ACC(0);
      issue_all(sB, it + 3);
      V2_ACC(1);
      V2_ITEM();
      ws_barrier();
      if (it + 1 < my_items) {
        V2_ACC(2);
        __builtin_amdgcn_s_sleep(PGV_V2_LOADER_SLEEP);
        V2_ACC(3);
        commit_all(sA, it + 2, tile0);       // item it+2 -> buffer 0 while item it+1 is multiplied from buffer 1
        V2_ACC(0);
        issue_all(sA, it + 4);
        V2_ACC(1);
        V2_ITEM();
        ws_barrier();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be landing in registers at wave exit
    V2_FLUSH();
    return;
  }
  // ==================================================== MFMA waves ===================================================
  const int wm = wave / NW, wn = wave - wm * NW;
  __builtin_amdgcn_s_setprio(PGV_V2_PRIO_MFMA);
  V2_T0();
  // per-lane B base of every pixel tile: pixel (r, c), tap kw = lane>>4: (2r)*WP + 2c - 2 + kw
  int offB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wn * NT + t) * 16 + (lane & 15);
    const int pv = p < P ? p : 0;
    const int r = pv / Ws, c = pv - r * Ws;
    offB[t] = 2 * r * WP + 2 * c - 2 + (lane >> 4);
  }
  // per-lane weight address: w[cs = mt*16 + (lane&15)][c][kh][kw = lane>>4]
  // (uniform base + 32-bit per-lane byte offset: the loads take the scalar-base form, no 64-bit pointers in VGPRs)
  const char* wb = reinterpret_cast<const char*>(w);
  unsigned wl[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) wl[m] = (unsigned)((((wm * MTW + m) * 16 + (lane & 15)) * CB * 16 + (lane >> 4)) * 4);
  auto wload = [&](int m, int elem) { return pgv_opnd(*reinterpret_cast<const float*>(wb + (size_t)elem * 4 + wl[m]), BF16); };
  static_assert(!BF16 || STG, "operand rounding: lean-loader forms only");
  const pgv_act_params actp = pgv_act_setup(act, slope);
  // D^T = X^T W^T: the accumulator of a lane is 4 consecutive pixels (rows (lane>>4)*4 + reg) of one channel (lane & 15)
  const int ech = lane & 15, epx = 4 * (lane >> 4);
  // FUSE (pgv_bwd_fuse): the lower block's BatchNorm + activation backward, g_y = act'(a) * (ka*g + kb*a + kc)
  float bias_r[MTW], ka_r[MTW], kb_r[MTW], kc_r[MTW];
  const pgv_actd_params actd = pgv_actd_setup(FUSE ? fuse.act : 0, FUSE ? fuse.slope : 0.f);
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int cl = (wm * MTW + m) * 16 + ech;
    bias_r[m] = bias ? bias[cl] : 0.f;
    ka_r[m] = FUSE ? fuse.coef[cl] : 0.f;
    kb_r[m] = FUSE ? fuse.coef[CS + cl] : 0.f;
    kc_r[m] = FUSE ? fuse.coef[2 * CS + cl] : 0.f;
  }
  float st_s[MTW], st_q[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) st_s[m] = st_q[m] = 0.f;
  // weights: a lane's A operand of k-step st is ONE dword; they are prefetched HS steps ahead into the other half of a
  // two-halves register ring (an even number of halves per item keeps every index a compile-time constant)
  constexpr int HS = (S % 16 == 0) ? 8 : S / 2;
  static_assert(S % (2 * HS) == 0, "weight ring");
  // STG (one channel chunk per unit): the S weights of a lane are the same for every item - they are loaded ONCE and the
  // loop holds no vector-memory loads at all.  That matters beyond the loads saved: gfx9 counts loads and stores in one
  // counter (vmcnt), so every wait for a weight load issued after a deferred store also waits for that store to
  // complete (10.7 k instead of 7.3 k clocks per item with the ring).
  constexpr bool WRES = STG && NCH == 1;
  float aw[2][MTW][WRES ? 1 : HS];
  float awr[WRES ? MTW : 1][WRES ? S : 1];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    if constexpr (WRES) {
#pragma unroll
      for (int i = 0; i < S; ++i) awr[m][i] = wload(m, i * 4);
    } else {
#pragma unroll
      for (int i = 0; i < HS; ++i) aw[0][m][i] = wload(m, i * 4);
    }
  }
  f32x4 acc[MTW][NT];
  // deferred stores (STG), see conv_up_ws_kernel: the previous unit's tiles, their byte offsets inside the unit (or the
  // out-of-range mark), this lane's channel offsets, the unit's buffer descriptor (zero bytes: nothing pending)
  constexpr unsigned OOR = 0x80000000u;
  f32x4 pend[STG ? MTW : 1][STG ? NT : 1];
  unsigned p4[STG ? NT : 1], choff[STG ? MTW : 1];
  i32x4 prs = {0, 0, 0, 0x00020000};
  if constexpr (STG) {
#pragma unroll
    for (int t = 0; t < NT; ++t) p4[t] = OOR;
#pragma unroll
    for (int m = 0; m < MTW; ++m) choff[m] = (unsigned)(((wm * MTW + m) * 16 + (lane & 15)) * (Hs * Ws) * 4);
  }
  // APRE: the saved-activation tiles of the unit being multiplied, this lane's byte offsets inside its band (or OOR), the
  // band's descriptor; row parity of the lane's 4 pixels in every tile relative to the band's first row (bit t); class
  // sums of the results [2 * row parity + column parity]
  f32x4 apre[APRE ? MTW : 1][APRE ? NT : 1];
  unsigned a4[APRE ? NT : 1];
  i32x4 ars = {0, 0, 0, 0x00020000};
  unsigned rbits = 0;
  float c4[APRE ? MTW : 1][4];
  if constexpr (APRE) {
#pragma unroll
    for (int t = 0; t < NT; ++t) rbits |= (unsigned)(((wn * NT * 16 + 4 * (lane >> 4) + t * 16) / Ws) & 1) << t;
#pragma unroll
    for (int m = 0; m < MTW; ++m) c4[m][0] = c4[m][1] = c4[m][2] = c4[m][3] = 0.f;
  }
  auto load_a = [&](auto qc) {
    constexpr int q = decltype(qc)::value, m = q / NT, t = q - m * NT;
    v2_buffer_load_x4(apre[APRE ? m : 0][APRE ? t : 0], a4[APRE ? t : 0] + choff[m], ars);
  };
  auto store_pending = [&](auto qc) {  // tile q = m * NT + t of the pending unit
    constexpr int q = decltype(qc)::value, m = q / NT, t = q - m * NT;
    const unsigned o4 = p4[t] + choff[m];
    const f32x4 all = pend[m][t];
    const i32x4 rs = prs;
    // (s_nop: a VALU write to the data registers of a > 8-byte store needs a wait state on gfx9; the compiler's hazard
    // recognizer cannot see into inline asm - without it some lanes stored the next instruction's result)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(all), "v"(o4), "s"(rs) : "memory");
  };
  ws_barrier();  // item 0 committed
  V2_ACC(0);
#pragma unroll 1
  for (int it = 0; it < my_items; ++it) {
    const int ch = it % NCH;
    const float* cur = tile0 + (it & 1) * BUF;
    // the operands of the first two k-steps first (LDS latency), the per-item bookkeeping behind them
    float bq[3][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bq[0][t] = cur[offB[t]];
#pragma unroll
    for (int t = 0; t < NT; ++t) bq[1][t] = cur[WP + offB[t]];
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int wc_ = ch * CK * 16, wn_ = ((it + 1) % NCH) * CK * 16;  // element offsets of this / the next item's chunk
    if constexpr (APRE) {   // where this unit's saved activation lies (one channel chunk per unit: it = unit)
      const int u = bid + it * gridDim.x;
      const int b = u / BANDS, band = u - b * BANDS;
      const int Pb = min(R, Hs - band * R) * Ws;
      ars = StageLean<CK, G::ROWS, W, WP, H, true>::band_rsrc(fuse.a, (int64_t)B * CS * (Hs * Ws) * 4,
                                                                ((int64_t)b * CS * Hs + band * R) * Ws);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p0 = wn * NT * 16 + 4 * (lane >> 4) + t * 16;
        a4[t] = p0 + 4 <= Pb ? (unsigned)p0 * 4u : OOR;
      }
    }
    V2_ACC(3);
#ifndef PGV_V2_NO_MFMA
    static_for<0, S>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      constexpr int sn = st + 2, cn = sn / 4, khn = sn - cn * 4;
      __builtin_amdgcn_sched_barrier(0);
      constexpr int SPREAD = S / (MTW * NT);  // the pending tiles leave evenly spread over the k-steps of the item
      if constexpr (STG && st % SPREAD == 0 && st / SPREAD < MTW * NT) {
        store_pending(std::integral_constant<int, st / SPREAD>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (APRE) {
          load_a(std::integral_constant<int, st / SPREAD>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (sn < S) bq[sn % 3][t] = cur[cn * PLANE + khn * WP + offB[t]];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
          acc[m][t] = PGV_MFMA4(bq[st % 3][t], WRES ? awr[m][WRES ? st : 0] : aw[(st / HS) & 1][m][WRES ? 0 : st % HS], acc[m][t]);
      }
      {  // the weight of step st + HS (same item, or the first half of the next one) into the other half of the ring
        constexpr int sp = st + HS;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
          if constexpr (!WRES)
            aw[((st / HS) + 1) & 1][m][st % HS] = sp < S ? wload(m, wc_ + sp * 4) : wload(m, wn_ + (sp - S) * 4);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, MTW, 0);            // MFMA
        if (sn < S) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
      }
    });
#endif
    __builtin_amdgcn_sched_barrier(0);
    V2_ACC(4);
    V2_ITEM();
#ifdef PGV_V2_NO_EPI
    if (false) {
#else
    if (ch == NCH - 1) {
#endif
      const int u = bid + (it / NCH) * gridDim.x;
      const int b = u / BANDS, band = u - b * BANDS;
      const int oh0 = band * R;
      const int Pb = min(R, Hs - oh0) * Ws;  // valid pixels of this band
      if constexpr (APRE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the unit's activation tiles have landed
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (STG)  // [this band of channel 0 of the sample .. end of the tensor)
        prs = StageLean<CK, G::ROWS, W, WP, H, true>::band_rsrc(out, (int64_t)B * CS * (Hs * Ws) * 4,
                                                                  ((int64_t)b * CS * Hs + oh0) * Ws);
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int cl = (wm * MTW + m) * 16 + ech;
        float* orow = out + ((int64_t)b * CS + cl) * (Hs * Ws) + (int64_t)oh0 * Ws;
        const float* arow = FUSE ? fuse.a + ((int64_t)b * CS + cl) * (Hs * Ws) + (int64_t)oh0 * Ws : nullptr;
        if constexpr (FUSE && !APRE) {
        // tiles in groups of 8: the saved-activation loads of a group (FUSE) are all issued before the first one is
        // used - one memory latency per group, not one per tile
        constexpr int TG = 8;
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TG) {
          f4u av[TG];
          if constexpr (FUSE) {
#pragma unroll
            for (int g = 0; g < TG; ++g) {
              const int tp0 = (wn * NT + t0 + g) * 16;
              if (t0 + g < NT && tp0 + 16 <= Pb) av[g] = *reinterpret_cast<const f4u*>(arow + tp0 + epx);
            }
          }
#pragma unroll
          for (int g = 0; g < TG; ++g) {
            const int t = t0 + g;
            if (t >= NT) continue;
            const int tp0 = (wn * NT + t) * 16;  // first pixel of the tile (wave-uniform)
            if (tp0 >= Pb) continue;             // tile entirely beyond the band
            const int p0 = tp0 + epx;
            float x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float y = acc[m][t][k] + bias_r[m];
              x[k] = ACT == 0 ? y : (ACT == 1 ? fmaxf(y, slope * y) : pgv_act_apply(y, actp));
            }
            if (tp0 + 16 <= Pb) {  // (wave-uniform) whole tile inside the band: one 16-byte store per lane
              const float avk[4] = {av[g].x, av[g].y, av[g].z, av[g].w};
#pragma unroll
              for (int k = 0; k < 4; ++k) x[k] = pgv_bwd_apply(x[k], avk[k], ka_r[m], kb_r[m], kc_r[m], actd);
              f4u o;
              o.x = x[0], o.y = x[1], o.z = x[2], o.w = x[3];
              *reinterpret_cast<f4u*>(orow + p0) = o;
              st_s[m] += (x[0] + x[1]) + (x[2] + x[3]);
            } else {  // ragged last tile of the band
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (p0 + k < Pb) {
                  const float y = pgv_bwd_apply(x[k], arow[p0 + k], ka_r[m], kb_r[m], kc_r[m], actd);
                  orow[p0 + k] = y;
                  st_s[m] += y;
                }
              }
            }
          }
        }
        } else {
        // No wave-uniform per-tile branches (each costs more than the tile's arithmetic): a lane whose 4 pixels lie inside
        // the band stores 16 bytes, the lane that straddles the end of the band stores its 1-3 pixels one by one, lanes
        // beyond it do nothing - all by exec mask.  Tiles in groups of 8: the saved-activation loads of a group (FUSE) are
        // all issued before the first one is used - one memory latency per group, not one per tile.
        constexpr int TG = NT > 16 ? 4 : 8;
        const f32x2 bias2 = {bias_r[m], bias_r[m]}, slope2 = {slope, slope};
        f32x2 ss = {0.f, 0.f}, qq = {0.f, 0.f};
        const int pl = wn * NT * 16 + epx;  // first pixel of this lane in tile 0 of the wave
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TG) {
#pragma unroll
          for (int g = 0; g < TG; ++g) {
            const int t = t0 + g;
            if (t >= NT) continue;
            const int p0 = pl + t * 16;
            f32x2 y0 = f32x2{acc[m][t][0], acc[m][t][1]} + bias2, y1 = f32x2{acc[m][t][2], acc[m][t][3]} + bias2;
            if (ACT == 1) {
              const f32x2 z0 = y0 * slope2, z1 = y1 * slope2;
              y0 = f32x2{fmaxf(y0.x, z0.x), fmaxf(y0.y, z0.y)};
              y1 = f32x2{fmaxf(y1.x, z1.x), fmaxf(y1.y, z1.y)};
            } else if (ACT != 0) {
              y0 = f32x2{pgv_act_apply(y0.x, actp), pgv_act_apply(y0.y, actp)};
              y1 = f32x2{pgv_act_apply(y1.x, actp), pgv_act_apply(y1.y, actp)};
            }
            if constexpr (APRE) {   // the lower block's BatchNorm + activation backward against the prefetched tile
              const f32x4 av = apre[m][t];
              y0 = f32x2{pgv_bwd_apply(y0.x, av.x, ka_r[m], kb_r[m], kc_r[m], actd),
                         pgv_bwd_apply(y0.y, av.y, ka_r[m], kb_r[m], kc_r[m], actd)};
              y1 = f32x2{pgv_bwd_apply(y1.x, av.z, ka_r[m], kb_r[m], kc_r[m], actd),
                         pgv_bwd_apply(y1.y, av.w, ka_r[m], kb_r[m], kc_r[m], actd)};
              if (p0 + 4 <= Pb) {    // class sums: the lane's 4 pixels lie in one row and start at an even column
                const bool rodd = (((rbits >> t) ^ (unsigned)oh0) & 1u) != 0;
                const float ev = y0.x + y1.x, od = y0.y + y1.y;
                c4[m][0] += rodd ? 0.f : ev, c4[m][1] += rodd ? 0.f : od, c4[m][2] += rodd ? ev : 0.f, c4[m][3] += rodd ? od : 0.f;
              }
            }
            if constexpr (STG) {
              pend[m][t] = f32x4{y0.x, y0.y, y1.x, y1.y};
              if (m == 0) p4[t] = p0 + 4 <= Pb ? (unsigned)p0 * 4u : OOR;
            }
            if (p0 + 4 <= Pb) {
              f4u o;
              o.x = y0.x, o.y = y0.y, o.z = y1.x, o.w = y1.y;
              if constexpr (!STG) *reinterpret_cast<f4u*>(orow + p0) = o;
              ss += y0 + y1;
              if constexpr (!FUSE) {
                qq = __builtin_elementwise_fma(y0, y0, qq);
                qq = __builtin_elementwise_fma(y1, y1, qq);
              }
            } else if (p0 < Pb) {  // the lane at the ragged end of the band
              const float x[4] = {y0.x, y0.y, y1.x, y1.y};
#pragma unroll
              for (int e = 0; e < 3; ++e) {
                if (p0 + e < Pb) {
                  orow[p0 + e] = x[e];
                  ss.x += x[e];
                  qq.x = fmaf(x[e], x[e], qq.x);
                }
              }
            }
          }
        }
        st_s[m] += ss.x + ss.y;
        st_q[m] += qq.x + qq.y;
        }
      }
    }
    V2_ACC(5);
    ws_barrier();  // everybody is done with buffer (it & 1); buffer (it+1) & 1 is committed
    V2_ACC(2);
  }
  if constexpr (STG) static_for<0, MTW * NT>([&](auto qc) { store_pending(qc); });  // the last unit
  V2_FLUSH();
  // statistics / projections: ONE float64 atomic per channel per workgroup (256 workgroups finishing together on 2*CS
  // addresses: the atomics serialise at the memory side, ~25 ns each - with one per wave they cost 10-25 us per launch).
  // Waves that share channels (NW > 1) are added up through LDS first; the loader waves have left, so this part uses
  // named waits on an LDS flag instead of a workgroup barrier.
  // (PGV_STATS_COPIES: into the partial copy of this workgroup's XCD - the finalize arithmetic adds the copies up)
  double* dst = (stats && stat_copies) ? stats + (blockIdx.x & (PGV_CLS_COPIES - 1)) * 2 * CS : stats;
  if constexpr (FUSE) {
    // bias gradient of the lower block (and the class sums of g_y, APRE): ONE float atomic per value per workgroup - the
    // 256 workgroups finish together and their atomics serialise per address (4 per workgroup cost 25 us per launch,
    // 20 per workgroup 75 us); waves that share channels are added up through LDS first
    constexpr int NV = APRE ? 5 : 1;
    float vals[MTW][NV];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      vals[m][0] = lanegroup_sum(st_s[m]);
      if constexpr (APRE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) vals[m][1 + k] = lanegroup_sum(c4[m][k]);
      }
    }
    auto flush = [&](int cl, const float (&v)[NV]) {
      if (fuse.gbias) atomicAdd(&fuse.gbias[(fuse.gbias_copies ? (blockIdx.x & (PGV_CLS_COPIES - 1)) * CS : 0) + cl], v[0]);
      if constexpr (APRE) {
        if (fuse.cls) {
#pragma unroll
          for (int k = 0; k < 4; ++k)   // (into the partial copy of this workgroup's XCD: 64 addresses x 256 workgroups on one
            atomicAdd(&fuse.cls[(blockIdx.x & (PGV_CLS_COPIES - 1)) * 4 * CS + 4 * cl + k], v[1 + k]);   // copy cost 21 us)
        }
      }
    };
    if constexpr (NW == 1) {
      if (lane < 16) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) flush((wm * MTW + m) * 16 + ech, vals[m]);
      }
    } else {
      float* red = tile0;  // [NW][CS][NV]; the input buffers are dead (the last barrier is behind us)
      if (lane < 16) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int k = 0; k < NV; ++k) red[(wn * CS + (wm * MTW + m) * 16 + ech) * NV + k] = vals[m][k];
      }
      // the 4 MFMA waves rendezvous on an LDS counter (the 4 loader waves never arrive at a barrier again)
      int* flag = reinterpret_cast<int*>(lds);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (wn == 0) {
        while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < 16) {
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            const int cl = (wm * MTW + m) * 16 + ech;
            float v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
              v[k] = 0.f;
#pragma unroll
              for (int j = 0; j < NW; ++j) v[k] += red[(j * CS + cl) * NV + k];
            }
            flush(cl, v);
          }
        }
      }
    }
  } else if (dst) {
    float* red = tile0;  // [NW][MTW*MW*16][2] floats; the input buffers are dead (the last barrier is behind us)
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const float ss = lanegroup_sum(st_s[m]), qq = lanegroup_sum(st_q[m]);
      if (lane < 16) {
        const int cl = (wm * MTW + m) * 16 + ech;
        if constexpr (NW == 1) {
          atomicAdd(&dst[cl], (double)ss);
          atomicAdd(&dst[CS + cl], (double)qq);
        } else {
          red[(wn * CS + cl) * 2 + 0] = ss;
          red[(wn * CS + cl) * 2 + 1] = qq;
        }
      }
    }
    if constexpr (NW > 1) {
      // the 4 MFMA waves rendezvous on an LDS counter (the 4 loader waves never arrive at a barrier again)
      int* flag = reinterpret_cast<int*>(lds);  // FRONT slack word 0 (re-zeroed below is not needed: kernel ends)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (wn == 0) {
        while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          if (lane < 16) {
            const int cl = (wm * MTW + m) * 16 + ech;
            double ss = 0.0, qq = 0.0;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
              ss += (double)red[(k * CS + cl) * 2 + 0];
              qq += (double)red[(k * CS + cl) * 2 + 1];
            }
            atomicAdd(&dst[cl], ss);
            atomicAdd(&dst[CS + cl], qq);
          }
        }
      }
    }
  }
}

template <int CB, int CS, int W, int H, int R, int MW, int CK>
int launch_down_v2(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                   const float* w, const float* bias, int act, float slope, float* out, double* stats,
                   const pgv_bwd_fuse* fuse, const pgv_bn_src* bn, hipStream_t st) {
  using G = DownV2Cfg<CB, CS, W, H, R, MW, CK>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  if (d->Cb != CB || d->Cs != CS) return 0;
  if (stats && fuse) return 0;  // one reduction slot
  // the two ways the train step calls it: forward of a Conv2D block (producer's BatchNorm folded or not, LeakyReLU,
  // statistics) and input gradient of a TConv2D block (plain product, optional BatchNorm-backward projections)
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, const float*, int, float, float*,
                         double*, pgv_bwd_fuse, pgv_bn_src, int);
  kern_t kern;
  const bool leaky = act == PGV_ACT_LEAKY_RELU && slope >= 0.f && slope <= 1.f;
  const int actk = act == PGV_ACT_NONE ? 0 : (leaky ? 1 : 2);
  if (bn && (!in_scale || fuse)) return 0;
#define PGV_DK(F, A, C) (kern_t) conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, F, A, C>
#ifdef PGV_V2_EXPERIMENT
  // tuning builds (scratch/build_dbg.sh): only the forward-call instantiation
  if (fuse || !in_scale || actk != 1) return 0;
  kern = PGV_DK(false, true, 1);
#else
  if (fuse)
    kern = in_scale ? PGV_DK(true, true, 2) : (actk == 0 ? PGV_DK(true, false, 0) : PGV_DK(true, false, 2));
  else if (in_scale)
    kern = actk == 1 ? PGV_DK(false, true, 1) : PGV_DK(false, true, 2);
  else
    kern = actk == 0 ? PGV_DK(false, false, 0) : (actk == 1 ? PGV_DK(false, false, 1) : PGV_DK(false, false, 2));
#endif
#undef PGV_DK
  // lean loader + deferred stores for the layer where they pay (129x174, one channel chunk); plain / LeakyReLU forms
  bool with_cls = false;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  bool bf16_ok = false;
  if constexpr (W == 174 && G::NCH == 1) {
    if (fuse && !in_scale && actk == 0) {   // fused backward epilogue with the saved activation prefetched (APRE)
      kern = bf16 ? (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, true, false, 0, true, true>
                  : (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, true, false, 0, true>;
      with_cls = fuse->cls != nullptr;
      bf16_ok = true;   // (bf16 operand mode: this form only - 80 us band kernel + 39 us class-sum pass otherwise)
    }
    if (!fuse && actk != 2) {
      if (in_scale)
        kern = actk == 1 ? (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, false, true, 1, true>
                         : (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, false, true, 0, true>;
      else
        kern = actk == 1 ? (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, false, false, 1, true>
                         : (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, false, false, 0, true>;
    }
  }
  if (bf16 && !bf16_ok) return 0;
  if (int rc = raise_lds_once((const void*)kern, "conv_down_v2")) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_v2: memset failed");
    return PGV_E_LAUNCH;
  }
  const int units = d->B * G::BANDS;
  const int grid = min(units, 256 * (V2_DOWN_WPS<R, W> / 2));
  const pgv_bwd_fuse fz = {nullptr, nullptr, nullptr, 0, 0.f, nullptr};
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), bytes, st, d->B, big, in_scale, in_shift, w, bias, act,
                     slope, out, stats, fuse ? *fuse : fz, bn ? *bn : pgv_no_bn(), (d->flags & PGV_STATS_COPIES) ? 1 : 0);
  PGV_CHECK_LAUNCH("conv_down_v2");
  return with_cls ? 3 : 1;   // (3: handled, class sums included)
}

}  // namespace

static int g_v2_down_variant = 0;
extern "C" int pgv_dbg_set_v2_down_variant(int v) {
  g_v2_down_variant = v;
  return 0;
}

// Returns 1 when handled, 0 when the shape / mode is not covered (the caller falls back to conv_band.hip), < 0 on error.
// Fused BatchNorm-backward projections (pgv_bwd_fuse) in the wave-specialised kernels: the saved-activation loads sit in
// the epilogue burst next to the stores and are not overlapped with matrix work (one workgroup per CU), so the fused
// form only pays where the alternative is worse (measured inside the train step, us, fused v2 / fused band / unfused v2
// + separate reduce pass):  down 129x174: 115 / 102 / 130 -> band;  down 65x88: 85 / 83 / 107 -> v2 fused;
// up 65x88: 154 / 89 / 138 -> band;  up 33x45: 133 / (no fused band kernel: 140) / 125 -> v2 unfused + reduce pass.
int pgv_conv_down_v2(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* small_out, double* stats,
                     const pgv_bwd_fuse* fuse, hipStream_t st, const pgv_bn_src* bn) {
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4) return 0;
  // bf16 operand mode with a weight shadow: the bf16-native kernels of the 33x45 / 65x88 layers (conv_deep_bf16.hip)
  if (int rc = pgv_conv_down_big_bf16(d, big, in_scale, in_shift, bias, act, slope, small_out, stats, fuse, st, bn)) return rc;
  // (bf16 operand mode: only the fused 129x174 input gradient has an operand-rounding instantiation, see launch_down_v2)
  if ((d->flags & PGV_COMPUTE_BF16) && !(d->Hb == 129 && d->Wb == 174 && fuse)) return 0;
  if (d->Hb == 33 && d->Wb == 45)   // 32 -> 64 channels, 17x23 outputs: the whole sample per unit, M split 4 ways
    return launch_down_v2<32, 64, 45, 33, 17, 4, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, bn, st);
  if (d->Hb == 65 && d->Wb == 88 && g_v2_down_variant == 1)   // (experiment: 7 bands of 5 rows, two workgroups per CU)
    return launch_down_v2<16, 32, 88, 65, 5, 2, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, bn, st);
  if (d->Hb == 65 && d->Wb == 88)   // 16 -> 32 channels, 33x45 outputs: 3 bands of 11 rows, waves 2 (M) x 2 (pixels)
    return launch_down_v2<16, 32, 88, 65, 11, 2, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, bn, st);
  if (d->Hb == 129 && d->Wb == 174)  // 8 -> 16 channels, 65x88 outputs: 13 bands of 5 rows, waves split the pixels
    return launch_down_v2<8, 16, 174, 129, 5, 1, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, bn, st);
  return 0;
}
