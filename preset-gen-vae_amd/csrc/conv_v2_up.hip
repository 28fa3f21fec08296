// Wave-specialised UP kernel (ConvTranspose2d forward / Conv2d input gradient) of the stride-2 k=4 layers at the
// reference sizes; structure and measurements: conv_v2_common.h, DESIGN.md section 3.4.
#define PGV_V2_TU up
#include "conv_v2_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// UP (ConvTranspose2d forward / Conv2d input-gradient), k = 4, stride 2, pad 2, by sub-pixel phases:
//   out[cb][2u+ph][2v+pw] = sum_{cs,th,tw} w[cs][cb][ph+2th][pw+2tw] * X[cs][u+1-th][v+1-tw]
// GEMM rows m = (cb, ph, pw) (M tile = 4 output channels x 4 phases), columns = grid positions (u, v) of the
// Hg x Wg = ceil(H/2) x ceil(W/2) sub-pixel grid (rows padded to an even width Wgp), one k-step per input channel
// (k lane = (th, tw)).  Same wave-specialised pipeline as conv_down_ws_kernel.  A lane's accumulator holds the 2x2 output
// block of one channel at one grid position; one exchange with the neighbouring lane (DPP) turns it into 4 consecutive
// pixels of one output row: a 16-byte store.
// ---------------------------------------------------------------------------------------------------------------
template <int CB, int CS, int W, int H, int R, int MW, int CK>
struct UpV2Cfg {
  static constexpr int Ws = W / 2 + 1, Hs = H / 2 + 1;      // input (small) size
  static constexpr int Wg = (W + 1) / 2, Hg = (H + 1) / 2;  // sub-pixel grid
  static constexpr int Wgp = (Wg + 1) / 2 * 2;
  static constexpr int BANDS = (Hg + R - 1) / R;
  static constexpr int NW = 4 / MW;
  static constexpr int MTT = CB / 4, MTW = MTT / MW;
  static constexpr int P = R * Wgp;
  static constexpr int NTT = (P + 15) / 16, NT = (NTT + NW - 1) / NW;
  static constexpr int ROWS = R + 1;
  static constexpr int WsP = (Ws + 1 + 3) / 4 * 4;
  static constexpr int PLANE = ROWS * WsP;
  static constexpr int NCH = CS / CK;
  static constexpr int S = CK;
  static constexpr int FRONT = 4;
  static constexpr int BUF = CK * PLANE;
  // + the epilogue's per-lane store geometry, [2][NT][256] ints (kept in LDS: the accumulators leave no registers for it)
  static constexpr size_t LDS_FLOATS = FRONT + 2 * (size_t)BUF + 2 * CS + 2 * (size_t)NT * 256;
  static_assert(CB % 4 == 0 && MTT % MW == 0 && CS % CK == 0 && 4 % MW == 0 && S >= 4, "tiling");
};

// STG ("deferred stores"): the epilogue leaves the finished band in REGISTERS and the stores go out one tile per k-step
// of the NEXT unit, from the MFMA waves themselves.  For the 129x174 layer the output of a unit (56 KB) leaving in one
// burst while the matrix pipe idles was 35 % of the kernel (the store path of a CU moves ~10 bytes per clock).  Handing
// the band to the loader waves through LDS does not work: their ~100 instruction slots per unit are used up by the input
// stage.  The stores are buffer stores with the hardware range check, so they need no branch inside the pinned k-step
// regions: lanes without (4 / 2) valid pixels carry an out-of-range offset, and a descriptor of zero bytes drops the
// stores of the first unit, which has nothing pending.  Needs NCH == 1 and an even image width; uses the lean loader.
// BF16: PGV_COMPUTE_BF16 - both operands rounded to bfloat16 (the weights where they are loaded, the input where it is
// committed to LDS), products on the fp32 MFMA.  Compile-time: as a run-time flag its test sat in the loader's commit
// loop and cost the fp32 kernels 20-30 % (the loader waves have no instruction slots to spare).
template <int CB, int CS, int W, int H, int R, int MW, int CK, bool FUSE, bool HAS_AFF, int ACT, bool STG = false,
          bool BF16 = false>
__global__ __launch_bounds__(512, 2) void conv_up_ws_kernel(int B, const float* __restrict__ small_in,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          int act, float slope, float* __restrict__ out,
                                                          double* __restrict__ stats, pgv_bwd_fuse fuse, pgv_bn_src bn,
                                                          int stat_copies) {
  using G = UpV2Cfg<CB, CS, W, H, R, MW, CK>;
  constexpr int Ws = G::Ws, Hs = G::Hs, Wg = G::Wg, Hg = G::Hg, Wgp = G::Wgp, BANDS = G::BANDS, NW = G::NW;
  constexpr int MTW = G::MTW, P = G::P, NT = G::NT, WsP = G::WsP, PLANE = G::PLANE, NCH = G::NCH, S = G::S, BUF = G::BUF;
  using Stage = StageV2<CK, G::ROWS, Ws, WsP, Hs, 1>;
  constexpr int NPF = Stage::NPF;
  static_assert(!STG || (NCH == 1 && ACT != 2 && W % 2 == 0 && S >= MTW * NT), "deferred stores");
  // STG + FUSE ("APRE"): the fused backward epilogue (pgv_bwd_fuse) with the saved activation of the unit PREFETCHED into
  // registers by the MFMA waves themselves, one tile per k-step (buffer loads with the hardware range check: no branch
  // inside the pinned k-step regions), so the epilogue finds it in registers; the stores then leave from the epilogue
  // (registers hold the activation tile instead of a pending output tile).  Plain products only (no bias-side affine).
  constexpr bool APRE = STG && FUSE, DEFER = STG && !FUSE;
  static_assert(!APRE || (!HAS_AFF && ACT == 0), "fused backward epilogue: plain input-gradient products");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile0 = lds + G::FRONT;
  float* aff = tile0 + 2 * BUF;  // [2][CS]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int units = B * BANDS;
  const int bid = pgv_xcd_block();  // first unit of this workgroup
  const int my_units = bid < units ? (units - bid + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int my_items = my_units * NCH;
  if (my_items == 0) return;
  if (tid < G::FRONT) lds[tid] = 0.f;
  for (int i = tid; i < CS; i += 512) {
    float sc = 1.f, sh = 0.f;
    // (pgv_conv_up_bn: the producer's BatchNorm is finalized here, from its statistics, instead of by a launch of its own)
    if (HAS_AFF && bn.stats)
      pgv_bn_finalize_dev(bn, CS, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[CS + i] = sh;
  }
  __syncthreads();
  auto item_src = [&](int it, const float*& plane0, int& ih0) {
    it = min(it, my_items - 1);
    const int u = bid + (it / NCH) * gridDim.x, ch = it % NCH;
    const int b = u / BANDS, band = u - b * BANDS;
    const uint64_t p = (uint64_t)(small_in + ((int64_t)b * CS + ch * CK) * (Hs * Ws));
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    plane0 = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
    ih0 = band * R;
  };

  if (STG && wave >= 4) {
    // ======================================= loader waves, lean form (see StageLean) =====================================
    // (an instruction of these waves gets an issue slot every ~70 clocks while the SIMD partner streams MFMAs: the
    // StageV2 loader needs ~180 of them per item of this layer, the budget is ~100)
    using Lean = StageLean<CK, G::ROWS, Ws, WsP, Hs>;
    constexpr int NL = Lean::NPF;
    const int ltid = tid - 256;
    __builtin_amdgcn_s_setprio(PGV_V2_PRIO_LOADER);
    typename Lean::Geo geo;
    typename Lean::Set sA, sB;
    static_assert(!STG || (BANDS >= 2 && (BANDS - 1) * R <= Hs), "edge bands");
    const int64_t bytes_in = (int64_t)B * CS * (Hs * Ws) * 4;
    auto item_geo = [&](int it, i32x4& rs, unsigned& bad) {
      it = min(it, my_items - 1);
      const int u = bid + it * gridDim.x;
      const int b = u / BANDS, band = u - b * BANDS;
      rs = Lean::band_rsrc(small_in, bytes_in, ((int64_t)b * CS * Hs + band * R) * Ws);
      bad = band == BANDS - 1 ? geo.bot_bad : 0u;  // rows below the input plane exist only in the last band
    };
    geo.init(ltid, aff, CS, HAS_AFF, 0, Hs - (BANDS - 1) * R, [&](auto jc) {
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
    if constexpr (HAS_AFF && NCH == 1) Stage::load_affine(geo, aff, CS, 0, sc, sh);
    auto commit_all = [&](const typename Stage::Set& sx, int it, float* dst) {
      if constexpr (HAS_AFF && NCH > 1) Stage::load_affine(geo, aff, CS, (min(it, my_items - 1) % NCH) * CK, sc, sh);
      Stage::wait_set();
      static_for<0, NPF>([&](auto j) {
        constexpr int J = decltype(j)::value;
        Stage::template commit_slot<J>(geo, sx, dst, ltid, HAS_AFF, sc[J], sh[J], BF16);
      });
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
  // ==================================================== MFMA waves ===================================================
  const int wm = wave / NW, wn = wave - wm * NW;
  __builtin_amdgcn_s_setprio(PGV_V2_PRIO_MFMA);
  // per-lane B base of every position tile: position (u, v), tap (th, tw) = lane>>4: (u+1-th)*WsP + (v+1-tw)
  int offB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wn * NT + t) * 16 + (lane & 15);
    const int pv = p < P ? p : 0;
    const int u = pv / Wgp, v = min(pv - u * Wgp, Wg - 1);
    offB[t] = (u + 1 - (lane >> 5)) * WsP + (v + 1 - ((lane >> 4) & 1));
  }
  // per-lane weight address: row (lane&15) = (cb = mt*4 + (row>>2), ph, pw), k = lane>>4 = (th, tw):
  // w[cs][cb][ph + 2 th][pw + 2 tw]
  const char* wb = reinterpret_cast<const char*>(w);
  unsigned wl[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int row = lane & 15, k = lane >> 4;
    const int cb = (wm * MTW + m) * 4 + (row >> 2), kh = ((row >> 1) & 1) + 2 * (k >> 1), kw = (row & 1) + 2 * (k & 1);
    wl[m] = (unsigned)((cb * 16 + kh * 4 + kw) * 4);
  }
  // weight of input channel cs (uniform base + 32-bit per-lane byte offset: scalar-base loads)
  auto wload = [&](int m, int cs) {
    return pgv_opnd(*reinterpret_cast<const float*>(wb + (size_t)cs * (CB * 16 * 4) + wl[m]), BF16);
  };
  const pgv_act_params actp = pgv_act_setup(act, slope);
  // accumulator layout: column (lane&15) = grid position, rows (lane>>4)*4 + reg = (channel lane>>4 of the M tile,
  // phase reg = ph*2 + pw); after the lane-pair exchange a lane holds 4 consecutive pixels of output row 2u + (lane&1)
  const int ech = lane >> 4, odd = lane & 1;
  float bias_r[MTW], ka_r[MTW], kb_r[MTW], kc_r[MTW];  // FUSE: see conv_down_ws_kernel
  const pgv_actd_params actd = pgv_actd_setup(FUSE ? fuse.act : 0, FUSE ? fuse.slope : 0.f);
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int cl = (wm * MTW + m) * 4 + ech;
    bias_r[m] = bias ? bias[cl] : 0.f;
    ka_r[m] = FUSE ? fuse.coef[cl] : 0.f;
    kb_r[m] = FUSE ? fuse.coef[CB + cl] : 0.f;
    kc_r[m] = FUSE ? fuse.coef[2 * CB + cl] : 0.f;
  }
  float st_s[MTW], st_q[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) st_s[m] = st_q[m] = 0.f;
  // weight ring (see conv_down_ws_kernel); 4-step halves where the accumulators leave no room for 8-step ones
  constexpr int HS = (S % 16 == 0 && MTW * NT < 28) ? 8 : (S % 8 == 0 ? 4 : S / 2);
  static_assert(S % (2 * HS) == 0, "weight ring");
  constexpr bool WRES = STG && NCH == 1;  // weights resident for the whole kernel (see conv_down_ws_kernel)
  float aw[2][MTW][WRES ? 1 : HS];
  float awr[WRES ? MTW : 1][WRES ? S : 1];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    if constexpr (WRES) {
#pragma unroll
      for (int i = 0; i < S; ++i) awr[m][i] = wload(m, i);
    } else {
#pragma unroll
      for (int i = 0; i < HS; ++i) aw[0][m][i] = wload(m, i);
    }
  }
  f32x4 acc[MTW][NT];
  // deferred stores (STG): the previous unit's output tiles, their byte offsets inside the unit (or an out-of-range mark)
  // for the lanes that store 16 / 8 bytes, this lane's channel offsets, and the unit's buffer descriptor
  constexpr unsigned OOR = 0x80000000u;  // stays out of range after the channel offset is added
  f32x4 pend[DEFER ? MTW : 1][DEFER ? NT : 1];
  unsigned p4[STG ? NT : 1], p2[DEFER ? NT : 1], choff[STG ? MTW : 1];
  i32x4 prs = {0, 0, 0, 0x00020000};  // zero bytes: nothing pending yet, every store is dropped
  // APRE: the unit's saved-activation tiles (p4 = this lane's byte offset inside the unit's band, OOR without >= 2 pixels)
  f32x4 apre[APRE ? MTW : 1][APRE ? NT : 1];
  if constexpr (STG) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      p4[t] = OOR;
      if constexpr (DEFER) p2[t] = OOR;
    }
#pragma unroll
    for (int m = 0; m < MTW; ++m) choff[m] = (unsigned)(((wm * MTW + m) * 4 + (lane >> 4)) * (H * W) * 4);
  }
  auto load_a = [&](auto qc) {  // APRE: saved activation of tile q = m * NT + t of the unit being multiplied
    constexpr int q = decltype(qc)::value, m = q / NT, t = q - m * NT;
    // (a lane with 2 valid pixels at the end of a row reads 2 floats of the next row along with them: in range or zero)
    v2_buffer_load_x4(apre[APRE ? m : 0][APRE ? t : 0], p4[t] + choff[m], prs);
  };
  auto store_pending = [&](auto qc) {  // tile q = m * NT + t of the pending unit
    constexpr int q = decltype(qc)::value, m = q / NT, t = q - m * NT;
    const unsigned o4 = p4[t] + choff[m], o2 = p2[DEFER ? t : 0] + choff[m];
    const f32x2 lo = {pend[DEFER ? m : 0][DEFER ? t : 0].x, pend[DEFER ? m : 0][DEFER ? t : 0].y};
    const f32x4 all = pend[DEFER ? m : 0][DEFER ? t : 0];
    const i32x4 rs = prs;
    // (s_nop: a VALU write to the data registers of a > 8-byte store needs a wait state on gfx9; the compiler's hazard
    // recognizer cannot see into inline asm - without it some lanes stored the next instruction's result)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(all), "v"(o4), "s"(rs) : "memory");
    asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(lo), "v"(o2), "s"(rs) : "memory");
  };
  // Epilogue geometry of a FULL band (R grid rows, 2R output rows), per pixel tile of this lane: byte-less offset of the
  // lane's 4 output pixels inside the band of one channel and the number of them that exist (0: tile position beyond the
  // band / padded grid column), packed as offset | count << 28.  Loop-invariant: computed once.
  // table 0: a full band (R grid rows, 2R output rows); table 1: the last band of a sample (fewer rows)
  int* tofl = reinterpret_cast<int*>(tile0 + 2 * BUF + 2 * CS) + tid;  // [2][NT][256], this lane's column
  constexpr int RB_LAST = Hg - (BANDS - 1) * R, HB_LAST = H - 2 * (BANDS - 1) * R < 2 * RB_LAST ? H - 2 * (BANDS - 1) * R : 2 * RB_LAST;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wn * NT + t) * 16 + (lane & 15);
    const int pu = p / Wgp, pv = p - pu * Wgp;
    const int orow = 2 * pu + odd, ocol = 2 * (pv & ~1);
    const int nv = min(max(W - ocol, 0), 4);
    tofl[t * 256] = (orow * W + ocol) | ((pu < R ? nv : 0) << 28);
    tofl[(NT + t) * 256] = (orow * W + ocol) | ((pu < RB_LAST && orow < HB_LAST ? nv : 0) << 28);
  }
  V2_T0();
  ws_barrier();  // item 0 committed
  V2_ACC(0);
#pragma unroll 1
  for (int it = 0; it < my_items; ++it) {
    const int ch = it % NCH;
    const float* cur = tile0 + (it & 1) * BUF;
    float bq[3][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bq[0][t] = cur[offB[t]];
#pragma unroll
    for (int t = 0; t < NT; ++t) bq[1][t] = cur[PLANE + offB[t]];
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int wc_ = ch * CK, wn_ = ((it + 1) % NCH) * CK;  // first input channel of this / the next item's chunk
    if constexpr (APRE) {   // where this unit's saved activation lies (one channel chunk per unit: it = unit)
      const int un = bid + it * gridDim.x;
      const int b = un / BANDS, band = un - b * BANDS;
      prs = StageLean<CK, G::ROWS, Ws, WsP, Hs>::band_rsrc(fuse.a, (int64_t)B * CB * (H * W) * 4,
                                                            ((int64_t)b * CB * H + 2 * band * R) * W);
      const int* tof = tofl + (band == BANDS - 1 ? NT * 256 : 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tv = tof[t * 256];
        p4[t] = ((unsigned)tv >> 28) >= 2u ? (unsigned)(tv & 0x0FFFFFFF) * 4u : OOR;
      }
    }
    V2_ACC(3);
    static_for<0, S>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      constexpr int sn = st + 2;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DEFER && st < MTW * NT) {
        store_pending(st_c);  // one tile of the previous unit leaves per k-step
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (APRE && st < MTW * NT) {
        load_a(st_c);         // one tile of this unit's saved activation arrives per k-step
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (sn < S) bq[sn % 3][t] = cur[sn * PLANE + offB[t]];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
          acc[m][t] = PGV_MFMA4(WRES ? awr[m][WRES ? st : 0] : aw[(st / HS) & 1][m][WRES ? 0 : st % HS], bq[st % 3][t], acc[m][t]);
      }
      {
        constexpr int sp = st + HS;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
          if constexpr (!WRES) aw[((st / HS) + 1) & 1][m][st % HS] = sp < S ? wload(m, wc_ + sp) : wload(m, wn_ + sp - S);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, MTW, 0);            // MFMA
        if (sn < S) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    V2_ACC(4);
    V2_ITEM();
    if (ch == NCH - 1) {
      // ---- epilogue of the unit
      const int un = bid + (it / NCH) * gridDim.x;
      const int b = un / BANDS, band = un - b * BANDS;
      const int u0 = band * R;
      const int Rb = min(R, Hg - u0);     // grid rows of this band
      const int Hb = min(2 * Rb, H - 2 * u0);  // output rows of this band
      if constexpr (APRE) {
        // ---- fused backward epilogue, table-driven like the plain one below; the activation tiles are in registers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int* tof = tofl + (band == BANDS - 1 ? NT * 256 : 0);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          const int cl = (wm * MTW + m) * 4 + ech;
          float* obase = out + (((int64_t)b * CB + cl) * H + 2 * u0) * W;
          const f32x2 bias2 = {bias_r[m], bias_r[m]};
          float ss = 0.f;
          int tvn = tof[0];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int tv = tvn;
            if (t + 1 < NT) tvn = tof[(t + 1) * 256];
            const f32x2 y0 = f32x2{acc[m][t][0], acc[m][t][1]} + bias2, y1 = f32x2{acc[m][t][2], acc[m][t][3]} + bias2;
            const float s0 = odd ? y0.x : y1.x, s1 = odd ? y0.y : y1.y;
            const float r0 = dpp_mov<0xB1>(s0), r1 = dpp_mov<0xB1>(s1);
            const f32x2 rr = {r0, r1};
            const f32x2 o01 = odd ? rr : y0, o23 = odd ? y1 : rr;
            const f32x4 av = apre[m][t];
            const float g0 = pgv_bwd_apply(o01.x, av.x, ka_r[m], kb_r[m], kc_r[m], actd);
            const float g1 = pgv_bwd_apply(o01.y, av.y, ka_r[m], kb_r[m], kc_r[m], actd);
            const float g2 = pgv_bwd_apply(o23.x, av.z, ka_r[m], kb_r[m], kc_r[m], actd);
            const float g3 = pgv_bwd_apply(o23.y, av.w, ka_r[m], kb_r[m], kc_r[m], actd);
            const int off = tv & 0x0FFFFFFF;
            const unsigned nv = (unsigned)tv >> 28;
            if (nv == 4) {
              f4u o;
              o.x = g0, o.y = g1, o.z = g2, o.w = g3;
              *reinterpret_cast<f4u*>(obase + off) = o;
              ss += (g0 + g1) + (g2 + g3);
            } else if (nv == 2) {   // (even width: the last two pixels of a row)
              *reinterpret_cast<float2*>(obase + off) = float2{g0, g1};
              ss += g0 + g1;
            }
          }
          st_s[m] += ss;
        }
      } else if constexpr (FUSE && ACT == 0 && !HAS_AFF) {
        // ---- fused backward epilogue of the multi-chunk layers ("WIN"): no registers are free during the k-steps to
        // prefetch the saved activation (the accumulators alone are 112-128), so its tiles come in through a ring of
        // D buffer loads that runs D tiles ahead of the arithmetic: compiler-visible raw buffer loads / stores with the
        // hardware range check (no control flow: the s_waitcnt counts stay exact), geometry from the tables of the
        // plain epilogue.  Only the first D loads of a unit expose their latency.
        const int* tof = tofl + (band == BANDS - 1 ? NT * 256 : 0);
        constexpr int NPART = W % 4;             // pixels of the lane at the end of a row of an odd-width image
        // (ring depth: what the accumulators leave of the 256 registers - deeper rings spill)
        constexpr int QN = MTW * NT, DMAX = QN * 4 >= 128 ? 6 : (QN * 4 >= 112 ? 9 : 12), D = QN < DMAX ? QN : DMAX;
        constexpr unsigned OOB = 0x80000000u;
        const int64_t elem0 = ((int64_t)b * CB * H + 2 * u0) * W;
        const unsigned left = (unsigned)min((int64_t)0x7FFFFFFF, ((int64_t)B * CB * (H * W) - elem0) * 4);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(fuse.a + elem0), 0, left, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + elem0), 0, left, 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 ar[D];
        unsigned ap[D][NPART ? NPART : 1];
        unsigned chb[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) chb[m] = (unsigned)(((wm * MTW + m) * 4 + ech) * (H * W) * 4);
        auto fetch = [&](auto qc, auto slot) {   // saved activation of tile q = m * NT + t into ring slot
          constexpr int q = decltype(qc)::value, sl = decltype(slot)::value, m = q / NT, t = q - m * NT;
          const int tv = tof[t * 256];
          const unsigned nv = (unsigned)tv >> 28, o4 = chb[m] + (unsigned)(tv & 0x0FFFFFFF) * 4u;
          ar[sl] = __builtin_amdgcn_raw_buffer_load_b128(ra, nv == 4 ? o4 : OOB, 0, 0);
#pragma unroll
          for (int e = 0; e < NPART; ++e)
            ap[sl][e] = __builtin_amdgcn_raw_buffer_load_b32(ra, (nv < 4 && (unsigned)e < nv) ? o4 + 4u * e : OOB, 0, 0);
        };
        static_for<0, D>([&](auto qc) { fetch(qc, qc); });
        float ssm[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) ssm[m] = 0.f;
        static_for<0, QN>([&](auto qc) {
          constexpr int q = decltype(qc)::value, sl = q % D, m = q / NT, t = q - m * NT;
          const int tv = tof[t * 256];
          const unsigned nv = (unsigned)tv >> 28, o4 = chb[m] + (unsigned)(tv & 0x0FFFFFFF) * 4u;
          const f32x2 bias2 = {bias_r[m], bias_r[m]};
          const f32x2 y0 = f32x2{acc[m][t][0], acc[m][t][1]} + bias2, y1 = f32x2{acc[m][t][2], acc[m][t][3]} + bias2;
          const float s0 = odd ? y0.x : y1.x, s1 = odd ? y0.y : y1.y;
          const float r0 = dpp_mov<0xB1>(s0), r1 = dpp_mov<0xB1>(s1);
          const f32x2 rr = {r0, r1};
          const f32x2 o01 = odd ? rr : y0, o23 = odd ? y1 : rr;
          // (out-of-range loads return 0: the 16-byte and the per-pixel loads of a lane never both hit)
          float av[4] = {__uint_as_float(ar[sl].x), __uint_as_float(ar[sl].y), __uint_as_float(ar[sl].z),
                         __uint_as_float(ar[sl].w)};
#pragma unroll
          for (int e = 0; e < NPART; ++e) av[e] += __uint_as_float(ap[sl][e]);
          const float g0 = pgv_bwd_apply(o01.x, av[0], ka_r[m], kb_r[m], kc_r[m], actd);
          const float g1 = pgv_bwd_apply(o01.y, av[1], ka_r[m], kb_r[m], kc_r[m], actd);
          const float g2 = pgv_bwd_apply(o23.x, av[2], ka_r[m], kb_r[m], kc_r[m], actd);
          const float g3 = pgv_bwd_apply(o23.y, av[3], ka_r[m], kb_r[m], kc_r[m], actd);
          const u32x4 gv = {__float_as_uint(g0), __float_as_uint(g1), __float_as_uint(g2), __float_as_uint(g3)};
          __builtin_amdgcn_raw_buffer_store_b128(gv, ro, nv == 4 ? o4 : OOB, 0, 0);
          const float ge[3] = {g0, g1, g2};
          float part = 0.f;
#pragma unroll
          for (int e = 0; e < NPART; ++e) {
            const bool on = nv < 4 && (unsigned)e < nv;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ge[e]), ro, on ? o4 + 4u * e : OOB, 0, 0);
            part += on ? ge[e] : 0.f;
          }
          ssm[m] += nv == 4 ? (g0 + g1) + (g2 + g3) : part;
          if constexpr (q + D < QN) fetch(std::integral_constant<int, q + D>{}, std::integral_constant<int, sl>{});
        });
#pragma unroll
        for (int m = 0; m < MTW; ++m) st_s[m] += ssm[m];
      } else if (!FUSE && ACT != 2) {
        const int* tof = tofl + (band == BANDS - 1 ? NT * 256 : 0);
        // No wave-uniform per-tile branches and no address arithmetic (a uniform branch per tile costs more than the
        // tile's arithmetic: the general path below spends ~480 clocks per tile): the geometry of the two kinds of band
        // comes from the tables computed at kernel start.  Lanes whose 4 pixels exist store 16 bytes; the lane at a row end of an odd-width image
        // stores its 1-3 pixels one by one; statistics ride in the same exec-masked blocks.
        const f32x2 slope2 = {slope, slope};
        if constexpr (DEFER)  // descriptor of [this band of channel 0 of the sample .. end of the tensor)
          prs = StageLean<CK, G::ROWS, Ws, WsP, Hs>::band_rsrc(out, (int64_t)B * CB * (H * W) * 4,
                                                                ((int64_t)b * CB * H + 2 * u0) * W);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          const int cl = (wm * MTW + m) * 4 + ech;
          float* obase = out + (((int64_t)b * CB + cl) * H + 2 * u0) * W;
          const f32x2 bias2 = {bias_r[m], bias_r[m]};
          f32x2 ss = {0.f, 0.f}, qq = {0.f, 0.f};
          int tvn = tof[0];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int tv = tvn;
            if (t + 1 < NT) tvn = tof[(t + 1) * 256];  // one tile ahead: the LDS latency hides under this tile's arithmetic
            f32x2 y0 = f32x2{acc[m][t][0], acc[m][t][1]} + bias2, y1 = f32x2{acc[m][t][2], acc[m][t][3]} + bias2;
            if (ACT == 1) {
              const f32x2 z0 = y0 * slope2, z1 = y1 * slope2;
              y0 = f32x2{fmaxf(y0.x, z0.x), fmaxf(y0.y, z0.y)};
              y1 = f32x2{fmaxf(y1.x, z1.x), fmaxf(y1.y, z1.y)};
            }
            // exchange with the neighbouring grid column: even lanes end up with output row 2u, odd lanes with row 2u+1
            const float s0 = odd ? y0.x : y1.x, s1 = odd ? y0.y : y1.y;
            const float r0 = dpp_mov<0xB1>(s0), r1 = dpp_mov<0xB1>(s1);
            const f32x2 rr = {r0, r1};
            const f32x2 o01 = odd ? rr : y0, o23 = odd ? y1 : rr;
            const int off = tv & 0x0FFFFFFF;
            const unsigned nv = (unsigned)tv >> 28;
            if constexpr (DEFER) {
              if (m == 0) {
                p4[t] = nv == 4 ? (unsigned)off * 4u : OOR;
                p2[DEFER ? t : 0] = nv == 2 ? (unsigned)off * 4u : OOR;
              }
            }
            if (nv == 4) {
              f4u o;
              o.x = o01.x, o.y = o01.y, o.z = o23.x, o.w = o23.y;
              if constexpr (DEFER) {
                pend[DEFER ? m : 0][DEFER ? t : 0] = f32x4{o.x, o.y, o.z, o.w};
              } else {
#ifndef PGV_V2_NO_STORE
                *reinterpret_cast<f4u*>(obase + off) = o;
#endif
              }
              ss += o01 + o23;
              qq = __builtin_elementwise_fma(o01, o01, qq);
              qq = __builtin_elementwise_fma(o23, o23, qq);
            } else if ((W % 4 != 0 || Wg != Wgp) && nv != 0) {
              const float ov[4] = {o01.x, o01.y, o23.x, o23.y};
#pragma unroll
              for (int e = 0; e < 3; ++e)
                if (e < (int)nv) {
                  if constexpr (DEFER)
                    pend[DEFER ? m : 0][DEFER ? t : 0] = f32x4{ov[0], ov[1], ov[2], ov[3]};  // (even width: 2 valid pixels, stored as 8 bytes)
                  else
                    obase[off + e] = ov[e];
                  ss.x += ov[e];
                  qq.x = fmaf(ov[e], ov[e], qq.x);
                }
            }
          }
          st_s[m] += ss.x + ss.y;
          st_q[m] += qq.x + qq.y;
        }
      } else
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int cl = (wm * MTW + m) * 4 + ech;
        float* obase = out + (((int64_t)b * CB + cl) * H + 2 * u0) * W;
        const float* abase = FUSE ? fuse.a + (((int64_t)b * CB + cl) * H + 2 * u0) * W : nullptr;
        // grid position of this lane in the wave's first tile, advanced by 16 positions per tile (Wgp > 16: at most one
        // row wrap per step)
        int pu, pv;
        {
          const int p = wn * NT * 16 + (lane & 15);
          pu = p / Wgp;
          pv = p - pu * Wgp;
        }
        constexpr int TG = 8;  // tiles per group: saved-activation loads of a group issued together (FUSE)
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TG) {
          int offs[TG];
          bool fulls[TG];
          f4u av[TG];
#pragma unroll
          for (int g = 0; g < TG; ++g) {
            const int orow = 2 * pu + odd, ocol = 2 * (pv & ~1);
            offs[g] = orow * W + ocol;
            const bool full = orow < Hb && ocol + 4 <= W;
            const int tp0 = (wn * NT + t0 + g) * 16;
            fulls[g] = t0 + g < NT && tp0 < Rb * Wgp && __builtin_amdgcn_ballot_w64(full) == ~0ull;
            if constexpr (FUSE) {
              if (fulls[g]) av[g] = *reinterpret_cast<const f4u*>(abase + offs[g]);
            }
            if (!fulls[g]) offs[g] = (orow < Hb) ? offs[g] | (min(max(W - ocol, 0), 4) << 28) : offs[g];  // nv in the top bits
            pv += 16;
            if (pv >= Wgp) {
              pv -= Wgp;
              ++pu;
            }
          }
#pragma unroll
          for (int g = 0; g < TG; ++g) {
            const int t = t0 + g;
            if (t >= NT) continue;
            const int tp0 = (wn * NT + t) * 16;
            if (tp0 >= Rb * Wgp) continue;  // (wave-uniform) tile entirely beyond the band
            float x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float y = acc[m][t][k] + bias_r[m];
              x[k] = ACT == 0 ? y : (ACT == 1 ? fmaxf(y, slope * y) : pgv_act_apply(y, actp));
            }
            // exchange with the neighbouring grid column: even lanes end up with output row 2u, odd lanes with row 2u+1
            const float s0 = odd ? x[0] : x[2], s1 = odd ? x[1] : x[3];
            const float r0 = dpp_mov<0xB1>(s0), r1 = dpp_mov<0xB1>(s1);
            float o0 = odd ? r0 : x[0], o1 = odd ? r1 : x[1], o2 = odd ? x[2] : r0, o3 = odd ? x[3] : r1;
            if (fulls[g]) {  // (wave-uniform) every lane stores 4 valid pixels
              const int off = offs[g];
              if constexpr (FUSE) {
                o0 = pgv_bwd_apply(o0, av[g].x, ka_r[m], kb_r[m], kc_r[m], actd);
                o1 = pgv_bwd_apply(o1, av[g].y, ka_r[m], kb_r[m], kc_r[m], actd);
                o2 = pgv_bwd_apply(o2, av[g].z, ka_r[m], kb_r[m], kc_r[m], actd);
                o3 = pgv_bwd_apply(o3, av[g].w, ka_r[m], kb_r[m], kc_r[m], actd);
              }
              f4u o;
              o.x = o0, o.y = o1, o.z = o2, o.w = o3;
#ifndef PGV_V2_NO_STORE
              *reinterpret_cast<f4u*>(obase + off) = o;
#endif
              st_s[m] += (o0 + o1) + (o2 + o3);
              if constexpr (!FUSE) {
                st_q[m] = fmaf(o0, o0, st_q[m]);
                st_q[m] = fmaf(o1, o1, st_q[m]);
                st_q[m] = fmaf(o2, o2, st_q[m]);
                st_q[m] = fmaf(o3, o3, st_q[m]);
              }
            } else {  // row ends of odd-width images, last row of odd-height images, padded grid column
              const int off = offs[g] & 0x0FFFFFFF, nv = (unsigned)offs[g] >> 28;
              const float ov[4] = {o0, o1, o2, o3};
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (k < nv) {
                  float y = ov[k];
                  if constexpr (FUSE) y = pgv_bwd_apply(y, abase[off + k], ka_r[m], kb_r[m], kc_r[m], actd);
                  obase[off + k] = y;
                  st_s[m] += y;
                  if constexpr (!FUSE) st_q[m] = fmaf(y, y, st_q[m]);
                }
              }
            }
          }
        }
      }
    }
    V2_ACC(5);
    ws_barrier();
    V2_ACC(2);
  }
  if constexpr (DEFER) static_for<0, MTW * NT>([&](auto qc) { store_pending(qc); });  // the last unit
  V2_FLUSH();
  // statistics / projections: one float64 atomic per channel per workgroup (see conv_down_ws_kernel)
  // (PGV_STATS_COPIES: into the partial copy of this workgroup's XCD - the finalize arithmetic adds the copies up)
  double* dst = (stats && stat_copies) ? stats + (blockIdx.x & (PGV_CLS_COPIES - 1)) * 2 * CB : stats;
  if constexpr (FUSE) {
    // bias gradient of the lower block: ONE float atomic per channel per workgroup (see conv_down_ws_kernel); waves that
    // share channels are added up through LDS first
    float vals[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) vals[m] = group16_sum(st_s[m]);
    if (fuse.gbias) {
      if constexpr (NW == 1) {
        if ((lane & 15) == 0) {
#pragma unroll
          for (int m = 0; m < MTW; ++m) atomicAdd(&fuse.gbias[(fuse.gbias_copies ? (blockIdx.x & (PGV_CLS_COPIES - 1)) * CB : 0) + (wm * MTW + m) * 4 + ech], vals[m]);
        }
      } else {
        float* red = tile0;  // [NW][CB]
        if ((lane & 15) == 0) {
#pragma unroll
          for (int m = 0; m < MTW; ++m) red[wn * CB + (wm * MTW + m) * 4 + ech] = vals[m];
        }
        int* flag = reinterpret_cast<int*>(lds);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (wn == 0) {
          while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if ((lane & 15) == 0) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
              const int cl = (wm * MTW + m) * 4 + ech;
              float v = 0.f;
#pragma unroll
              for (int j = 0; j < NW; ++j) v += red[j * CB + cl];
              atomicAdd(&fuse.gbias[(fuse.gbias_copies ? (blockIdx.x & (PGV_CLS_COPIES - 1)) * CB : 0) + cl], v);
            }
          }
        }
      }
    }
  } else if (dst) {
    float* red = tile0;  // [NW][CB][2]
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const float ss = group16_sum(st_s[m]), qq = group16_sum(st_q[m]);
      if ((lane & 15) == 0) {
        const int cl = (wm * MTW + m) * 4 + ech;
        if constexpr (NW == 1) {
          atomicAdd(&dst[cl], (double)ss);
          atomicAdd(&dst[CB + cl], (double)qq);
        } else {
          red[(wn * CB + cl) * 2 + 0] = ss;
          red[(wn * CB + cl) * 2 + 1] = qq;
        }
      }
    }
    if constexpr (NW > 1) {
      int* flag = reinterpret_cast<int*>(lds);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (wn == 0) {
        while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          if ((lane & 15) == 0) {
            const int cl = (wm * MTW + m) * 4 + ech;
            double ss = 0.0, qq = 0.0;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
              ss += (double)red[(k * CB + cl) * 2 + 0];
              qq += (double)red[(k * CB + cl) * 2 + 1];
            }
            atomicAdd(&dst[cl], ss);
            atomicAdd(&dst[CB + cl], qq);
          }
        }
      }
    }
  }
}

template <int CB, int CS, int W, int H, int R, int MW, int CK>
int launch_up_v2(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                 const float* w, const float* bias, int act, float slope, float* out, double* stats,
                 const pgv_bwd_fuse* fuse, const pgv_bn_src* bn, hipStream_t st) {
  using G = UpV2Cfg<CB, CS, W, H, R, MW, CK>;
  // deferred stores for the layer whose output bursts bound it (129x174: 56 KB per unit), where the variant exists
  constexpr bool STG = W == 174 && G::NCH == 1;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  if (d->Cb != CB || d->Cs != CS) return 0;
  if (stats && fuse) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, const float*, int, float, float*,
                         double*, pgv_bwd_fuse, pgv_bn_src, int);
  kern_t kern;
  const bool leaky = act == PGV_ACT_LEAKY_RELU && slope >= 0.f && slope <= 1.f;
  const int actk = act == PGV_ACT_NONE ? 0 : (leaky ? 1 : 2);
  if (bn && (!in_scale || fuse)) return 0;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
#define PGV_UK(F, A, C) (kern_t) conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, F, A, C>
#ifdef PGV_V2_EXPERIMENT
  if (fuse || !in_scale || actk != 1) return 0;
  kern = PGV_UK(false, true, 1);
#else
  if (fuse)
    kern = in_scale ? PGV_UK(true, true, 2) : (actk == 0 ? PGV_UK(true, false, 0) : PGV_UK(true, false, 2));
  else if (in_scale)
    kern = actk == 1 ? PGV_UK(false, true, 1) : PGV_UK(false, true, 2);
  else
    kern = actk == 0 ? PGV_UK(false, false, 0) : (actk == 1 ? PGV_UK(false, false, 1) : PGV_UK(false, false, 2));
#endif
#undef PGV_UK
  if constexpr (STG) {  // (the LeakyReLU / linear forms exist with deferred stores, the plain product with the fused epilogue)
    if (actk == 2 || (fuse && (in_scale || actk != 0))) return 0;
    if (fuse)
      kern = (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, true, false, 0, true>;
    else if (in_scale)
      kern = actk == 1 ? (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, false, true, 1, true>
                       : (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, false, true, 0, true>;
    else
      kern = actk == 1 ? (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, false, false, 1, true>
                       : (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, false, false, 0, true>;
  }
  if (bf16) {   // operand-rounding instantiations exist for the 33x45 layer in the forms the train step issues
    if constexpr (STG) {   // ... and for the 129x174 input gradient with the fused backward epilogue
      if (!(fuse && !in_scale && actk == 0)) return 0;
      kern = (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, true, false, 0, true, true>;
    } else if constexpr (W == 45 && !STG) {
#define PGV_UKB(F, A, C) (kern_t) conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, F, A, C, false, true>
      if (fuse && !in_scale && actk == 0)
        kern = PGV_UKB(true, false, 0);
      else if (!fuse && !in_scale && actk == 0)
        kern = PGV_UKB(false, false, 0);
      else if (!fuse && !in_scale && actk == 1)
        kern = PGV_UKB(false, false, 1);
      else if (!fuse && in_scale && actk == 1)
        kern = PGV_UKB(false, true, 1);
      else
        return 0;
#undef PGV_UKB
    } else {
      return 0;
    }
  }
  if (int rc = raise_lds_once((const void*)kern, "conv_up_v2")) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_v2: memset failed");
    return PGV_E_LAUNCH;
  }
  const int units = d->B * G::BANDS;
  const int grid = min(units, 256);
  const pgv_bwd_fuse fz = {nullptr, nullptr, nullptr, 0, 0.f, nullptr};
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), bytes, st, d->B, small_in, in_scale, in_shift, w, bias, act, slope,
                     out, stats, fuse ? *fuse : fz, bn ? *bn : pgv_no_bn(), (d->flags & PGV_STATS_COPIES) ? 1 : 0);
  PGV_CHECK_LAUNCH("conv_up_v2");
  return 1;
}

}  // namespace

// (returns 2 when it handled the call but left the requested projections to a separate reduce pass)
int pgv_conv_up_v2(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                   const float* w, const float* bias, int act, float slope, float* big_out, double* stats,
                   const pgv_bwd_fuse* fuse, hipStream_t st, const pgv_bn_src* bn) {
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4) return 0;
  // bf16 operand mode with a weight shadow: the bf16-native kernel of the 33x45 layer (conv_deep_bf16.hip)
  if (int rc = pgv_conv_up_big_bf16(d, small_in, in_scale, in_shift, bias, act, slope, big_out, stats, fuse, st, bn)) return rc;
  // bf16 operand mode: only the 33x45 layer comes here (operands rounded at the LDS commit / weight load, fp32 MFMA: 95 us
  // against 157 us for the band kernel's bf16 loop at this shape; the other shapes' band kernels are faster than this form)
  // (and the fused 129x174 input gradient: 204 us on the band kernel's bf16 loop)
  if ((d->flags & PGV_COMPUTE_BF16) && !(d->Hb == 33 && d->Wb == 45) && !(d->Hb == 129 && d->Wb == 174 && fuse)) return 0;
  // fused backward epilogue (pgv_bwd_fuse): 129x174 prefetches the saved activation during the k-steps (APRE), the
  // multi-chunk layers run a ring of buffer loads through the epilogue (WIN); only plain input-gradient products
  // (65x88: 128 accumulator registers leave no room for the ring - it spills; the band kernel's fused epilogue stays)
  if (fuse && d->Hb == 65 && d->Wb == 88) return 0;
  if (d->Hb == 33 && d->Wb == 45)   // 64 -> 32 channels onto 33x45: 2 bands of 9 / 8 grid rows, M split 4 ways
    return launch_up_v2<32, 64, 45, 33, 9, 4, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, bn, st);
  if (d->Hb == 65 && d->Wb == 88)   // 32 -> 16 channels onto 65x88: 3 bands of 11 grid rows, waves split the positions
    return launch_up_v2<16, 32, 88, 65, 11, 1, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, bn, st);
#ifndef PGV_V2_NO_UP_L2
  // 16 -> 8 channels onto 129x174 (13 bands of 5 grid rows, waves split the positions).  This layer is bound by the CU's
  // store path (56 KB of output per unit): with direct stores from the epilogue the matrix pipe idled 35 % of the time
  // and the band kernel's two co-resident workgroups were faster; with the output staged through LDS and moved out by
  // the loader waves during the next unit's k-steps (STG) this form wins.  Fused projections stay on the band kernel.
  if (d->Hb == 129 && d->Wb == 174)
    return launch_up_v2<8, 16, 174, 129, 5, 1, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, bn, st);
#endif
  return 0;
}
