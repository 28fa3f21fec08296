// Wave-specialised weight-gradient kernels (k=4 layers and the 5x5 layers of the 1 <-> 8 channel pair) and the
// partial-gradient reduce pass; structure and measurements: conv_v2_common.h, DESIGN.md section 3.4.
#define PGV_V2_TU wgrad
#include "conv_v2_common.h"
#include "bn_taps.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// WGRAD, k = 4, stride 2, pad 2:  gw[cs][cb][kh][kw] = sum_{b,oh,ow} small[b,cs,oh,ow] * big[b,cb,2oh-2+kh,2ow-2+kw]
// GEMM with M = cs (16 per tile), N = (cb, 16 taps) = one N tile per big channel, K = output pixels (4 consecutive ow per
// MFMA).  A[cs][pixel] from the small tile, B[pixel][tap] straight from the raw big tile.  Every MFMA wave holds ALL M
// tiles for CB/4 big channels (MT + CB/4 operand reads per MT*CB/4 MFMAs: the LDS is idle most of the time, bank
// conflicts of the A reads do not matter); the accumulators live in registers over all units of the persistent
// workgroup.  Work item = R output rows of one sample (both tiles double-buffered in LDS, two items in flight in the
// loader's registers).  Flush: per-workgroup partial sums go to a workspace with plain stores and a second kernel adds
// them up (256 workgroups x the whole gradient as float atomics cost 26 us on the 64x32-channel layer: the atomics
// execute at the memory side at 1.3 TB/s).
// ---------------------------------------------------------------------------------------------------------------
template <int CB, int CS, int W, int H, int R>
struct WgradV2Cfg {
  static constexpr int Ws = W / 2 + 1, Hs = H / 2 + 1;
  static constexpr int BANDS = (Hs + R - 1) / R;
  static constexpr int MT = CS / 16, NBW = CB / 4;            // M tiles per wave (all), big channels per wave
  static constexpr int ROWS_B = 2 * (R - 1) + 4;
  // row stride of the big tile = 8 mod 16: the four kernel rows of a B fragment (6 consecutive floats each per 32-lane
  // half) then fall on disjoint bank ranges of the 32 ds_read_b32 banks
  static constexpr int WP = (W + 2 - 8 + 15) / 16 * 16 + 8;
  static constexpr int WsP = (Ws + 3) / 4 * 4;
  // small-tile plane stride = an odd number of 16-byte groups: the A fragment reads one pixel of 16 channels per 16
  // lanes - with the planes back to back (72 / 144 / 264 floats: 8, 16, 8 mod 32) that is a 4- to 8-way bank conflict,
  // with an odd group count the 16 channels fall on 8 different bank groups (2-way)
  static constexpr int PLANE_B = ROWS_B * WP, PLANE_S = R * WsP + ((R * WsP / 4) % 2 == 0 ? 4 : 0);
  static constexpr int SPR = WsP / 4;                         // k-steps per output row
  static constexpr int S = R * SPR;
  static constexpr int FRONT = 4;
  // one item: FRONT zero floats (what column -2 of the first row of the first plane reads), big tile, small tile
  static constexpr int BUF = FRONT + CB * PLANE_B + CS * PLANE_S;
  static constexpr size_t LDS_FLOATS = 2 * (size_t)BUF + 2 * (CB + CS);
  static_assert(CS % 16 == 0 && CB % 4 == 0 && WP >= W + 2 && WP % 4 == 0 && 2 * WsP <= WP, "tiling");
};

template <int CB, int CS, int W, int H, int R, bool AFF_B, bool AFF_S>
__global__ __launch_bounds__(512, 2) void conv_wgrad_ws_kernel(int B, const float* __restrict__ big,
                                                             const float* __restrict__ big_scale,
                                                             const float* __restrict__ big_shift,
                                                             const float* __restrict__ small_in,
                                                             const float* __restrict__ small_scale,
                                                             const float* __restrict__ small_shift,
                                                             float* __restrict__ partial) {
  using G = WgradV2Cfg<CB, CS, W, H, R>;
  constexpr int Ws = G::Ws, Hs = G::Hs, BANDS = G::BANDS, MT = G::MT, NBW = G::NBW, WP = G::WP, WsP = G::WsP;
  constexpr int PLANE_B = G::PLANE_B, PLANE_S = G::PLANE_S, SPR = G::SPR, BUF = G::BUF;
  using StageB = StageLean<CB, G::ROWS_B, W, WP, H>;
  using StageS = StageLean<CS, R, Ws, WsP, Hs, false, PLANE_S>;
  constexpr int NPB = StageB::NPF, NPS = StageS::NPF;
  static_assert(NPB + NPS < 64, "vmcnt range");
  // columns >= W / >= Ws (what lies behind the end of a row in its last chunk) are only reached in the last k-step of a row
  static_assert(SPR >= 2 && SPR % 2 == 0 && 8 * (SPR - 2) + 7 < W && 4 * (SPR - 1) <= Ws, "tail masking");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile0 = lds + G::FRONT;
  float* aff_b = tile0 + 2 * BUF;  // [2][CB]
  float* aff_s = aff_b + 2 * CB;   // [2][CS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int units = B * BANDS;
  const int bid = pgv_xcd_block();  // first unit of this workgroup
  const int my_items = bid < units ? (units - bid + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (tid < 2 * G::FRONT) lds[(tid / G::FRONT) * BUF + tid % G::FRONT] = 0.f;
  if (AFF_B)
    for (int i = tid; i < CB; i += 512) aff_b[i] = big_scale[i], aff_b[CB + i] = big_shift[i];
  if (AFF_S)
    for (int i = tid; i < CS; i += 512) aff_s[i] = small_scale[i], aff_s[CS + i] = small_shift[i];
  __syncthreads();
  if (wave >= 4) {
    // ================================================= loader waves =================================================
    if (my_items == 0) return;
    const int ltid = tid - 256;
    __builtin_amdgcn_s_setprio(PGV_V2_PRIO_LOADER);
    V2_T0();
    typename StageB::Geo geoB;
    typename StageS::Geo geoS;
    typename StageB::Set bA, bB;
    typename StageS::Set sA, sB;
    // only the first and the last band of a sample touch rows outside the image
    static_assert(BANDS >= 3 && (BANDS - 2) * 2 * R - 2 + G::ROWS_B <= H && (BANDS - 1) * R <= Hs, "edge bands");
    auto first_item = [&](auto stage_is_big, auto jc) {  // the loads of item 0, slot by slot out of the set-up
      constexpr int J = decltype(jc)::value;
      const int u = bid, b = u / BANDS, band = u - b * BANDS;
      if constexpr (decltype(stage_is_big)::value) {
        const i32x4 rb = StageB::band_rsrc(big, (int64_t)B * CB * (H * W) * 4, ((int64_t)b * CB * H + band * 2 * R - 2) * W);
        StageB::template issue_slot<J, true>(geoB, bA, rb, band == 0 ? geoB.top_bad : (band == BANDS - 1 ? geoB.bot_bad : 0u));
      } else {
        const i32x4 rs = StageS::band_rsrc(small_in, (int64_t)B * CS * (Hs * Ws) * 4, ((int64_t)b * CS * Hs + band * R) * Ws);
        StageS::template issue_slot<J, true>(geoS, sA, rs, band == 0 ? geoS.top_bad : (band == BANDS - 1 ? geoS.bot_bad : 0u));
      }
    };
    geoB.init(ltid, aff_b, CB, AFF_B, 2, H - ((BANDS - 1) * 2 * R - 2), [&](auto jc) { first_item(std::true_type{}, jc); });
    geoS.init(ltid, aff_s, CS, AFF_S, 0, Hs - (BANDS - 1) * R, [&](auto jc) { first_item(std::false_type{}, jc); });
    const int64_t bytes_b = (int64_t)B * CB * (H * W) * 4, bytes_s = (int64_t)B * CS * (Hs * Ws) * 4;
    auto band_of = [&](int it, int& b, int& band) {
      it = min(it, my_items - 1);
      const int u = bid + it * gridDim.x;
      b = u / BANDS;
      band = u - b * BANDS;
    };
    auto is_edge = [&](int band) { return band == 0 || band == BANDS - 1; };
    auto issue_all = [&](typename StageB::Set& bx, typename StageS::Set& sx, int it) {
      int b, band;
      band_of(it, b, band);
      const int ihb = band * 2 * R - 2, ihs = band * R;
      const i32x4 rb = StageB::band_rsrc(big, bytes_b, ((int64_t)b * CB * H + ihb) * W);
      const i32x4 rs = StageS::band_rsrc(small_in, bytes_s, ((int64_t)b * CS * Hs + ihs) * Ws);
      if (is_edge(band)) {
        const unsigned badb = band == 0 ? geoB.top_bad : geoB.bot_bad, bads = band == 0 ? geoS.top_bad : geoS.bot_bad;
        static_for<0, NPB>([&](auto j) { StageB::template issue_slot<decltype(j)::value, true>(geoB, bx, rb, badb); });
        static_for<0, NPS>([&](auto j) { StageS::template issue_slot<decltype(j)::value, true>(geoS, sx, rs, bads); });
      } else {
        static_for<0, NPB>([&](auto j) { StageB::template issue_slot<decltype(j)::value, false>(geoB, bx, rb, 0u); });
        static_for<0, NPS>([&](auto j) { StageS::template issue_slot<decltype(j)::value, false>(geoS, sx, rs, 0u); });
      }
    };
    auto commit_all = [&](const typename StageB::Set& bx, const typename StageS::Set& sx, int it, float* dst) {
      int b, band;
      band_of(it, b, band);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB + NPS) : "memory");  // the older item has landed
      __builtin_amdgcn_sched_barrier(0);
      V2_ACC(5);
      if ((AFF_B || AFF_S) && is_edge(band)) {
        const unsigned badb = band == 0 ? geoB.top_bad : geoB.bot_bad, bads = band == 0 ? geoS.top_bad : geoS.bot_bad;
        static_for<0, NPB>([&](auto j) { StageB::template commit_slot<decltype(j)::value, true, AFF_B>(geoB, bx, dst, ltid, badb); });
        static_for<0, NPS>([&](auto j) {
          StageS::template commit_slot<decltype(j)::value, true, AFF_S>(geoS, sx, dst + CB * PLANE_B, ltid, bads);
        });
      } else {
        static_for<0, NPB>([&](auto j) { StageB::template commit_slot<decltype(j)::value, false, AFF_B>(geoB, bx, dst, ltid, 0u); });
        static_for<0, NPS>([&](auto j) {
          StageS::template commit_slot<decltype(j)::value, false, AFF_S>(geoS, sx, dst + CB * PLANE_B, ltid, 0u);
        });
      }
    };
    issue_all(bB, sB, 1);  // (item 0 went out during the set-up)
    commit_all(bA, sA, 0, tile0);
    issue_all(bA, sA, 2);
    V2_ACC(6);
    ws_barrier();
#pragma unroll 1
    for (int it = 0; it < my_items; it += 2) {
      V2_ACC(2);
      __builtin_amdgcn_s_sleep(PGV_V2_LOADER_SLEEP);
      V2_ACC(3);
      commit_all(bB, sB, it + 1, tile0 + BUF);
      V2_ACC(0);
      issue_all(bB, sB, it + 3);
      V2_ACC(1);
      V2_ITEM();
      ws_barrier();
      if (it + 1 < my_items) {
        V2_ACC(2);
        __builtin_amdgcn_s_sleep(PGV_V2_LOADER_SLEEP);
        V2_ACC(3);
        commit_all(bA, sA, it + 2, tile0);
        V2_ACC(0);
        issue_all(bA, sA, it + 4);
        V2_ACC(1);
        V2_ITEM();
        ws_barrier();
      }
    }
    V2_ACC(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V2_FLUSH();
    return;
  }
  // ==================================================== MFMA waves ===================================================
  __builtin_amdgcn_s_setprio(PGV_V2_PRIO_MFMA);
  V2_T0();
  // A: lane (m = cs = lane&15, k = pixel lane>>4) reads small[cs][r][4i + k]; B: lane (n = tap = lane&15, k) reads
  // big[cb][2r + kh][2(4i + k) + kw - 2]  (column -2 of a row = the zero tail of the row before / the zero front)
  int offA[MT], offB[NBW];
#pragma unroll
  for (int m = 0; m < MT; ++m) offA[m] = CB * PLANE_B + (m * 16 + (lane & 15)) * PLANE_S + (lane >> 4);
#pragma unroll
  for (int n = 0; n < NBW; ++n) {
    const int tap = lane & 15;
    offB[n] = (wave * NBW + n) * PLANE_B + (tap >> 2) * WP + (tap & 3) - 2 + 2 * (lane >> 4);
  }
  // operand lanes of the last k-step of a row that lie behind the end of the row (the loader does not clean them)
  const bool keepA = 4 * (SPR - 1) + (lane >> 4) < Ws;
  const bool keepB = 8 * (SPR - 1) + 2 * (lane >> 4) + (lane & 3) - 2 < W;
  f32x4 acc[MT][NBW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NBW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  V2_ACC(0);
  if (my_items > 0) {
    ws_barrier();  // item 0 committed
#pragma unroll 1
    for (int it = 0; it < my_items; ++it) {
      V2_ACC(2);
      const float* cur = tile0 + (it & 1) * BUF;
      const int u = bid + it * gridDim.x;
      const int nrows = min(R, Hs - (u % BANDS) * R);  // the last band of a sample may be short
      // k-steps go in pairs (two consecutive 4-pixel groups of a row): the two operand values of a lane lie 4 (A) / 8 (B)
      // floats apart and come from ONE ds_read2_b32 - an LDS instruction of the MFMA wave costs MFMA issue time
      // (3-7 clocks each, they do not hide under the 32 clocks of an MFMA), so there should be few of them
      constexpr int NSET = MT * NBW >= 32 ? 2 : 3;  // operand pairs in flight + 1 (the big tile has no registers for 3)
      float av[NSET][2][MT], bv[NSET][2][NBW];
      auto load_pair = [&](int ps, float (&a)[2][MT], float (&b)[2][NBW]) {
        const int r = ps / (SPR / 2), i = 2 * (ps - r * (SPR / 2));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int m = 0; m < MT; ++m) a[h][m] = cur[offA[m] + r * WsP + 4 * (i + h)];
#pragma unroll
          for (int n = 0; n < NBW; ++n) b[h][n] = cur[offB[n] + 2 * r * WP + 8 * (i + h)];
        }
        if (i + 1 == SPR - 1) {
          if (Ws % 4 != 0)
#pragma unroll
            for (int m = 0; m < MT; ++m) a[1][m] = keepA ? a[1][m] : 0.f;
          if (W % 4 != 0)
#pragma unroll
            for (int n = 0; n < NBW; ++n) b[1][n] = keepB ? b[1][n] : 0.f;
        }
      };
      constexpr int PPR = SPR / 2, PS = R * PPR;  // pairs per row, per item
      load_pair(0, av[0], bv[0]);
      if (NSET == 3) load_pair(1, av[1], bv[1]);
      static_for<0, R>([&](auto r_c) {
        constexpr int r = decltype(r_c)::value;
        if (Hs % R == 0 || r < nrows) {
          static_for<0, PPR>([&](auto i_c) {
            constexpr int ps = r * PPR + decltype(i_c)::value;
            constexpr int pn = ps + NSET - 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (pn < PS) load_pair(pn, av[pn % NSET], bv[pn % NSET]);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int n = 0; n < NBW; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][n] = PGV_MFMA4(av[ps % NSET][h][m], bv[ps % NSET][h][n], acc[m][n]);
            // MT + NBW reads under 2 * MT * NBW MFMAs: one read behind each of the first MFMAs
            static_for<0, 2 * MT * NBW>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if constexpr (pn < PS && k < MT + NBW)
                __builtin_amdgcn_sched_group_barrier(0x100, k + 1 == 2 * MT * NBW ? MT + NBW - k : 1, 0);
            });
          });
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      V2_ACC(4);
      V2_ITEM();
      ws_barrier();
    }
  }
  V2_ACC(2);
  // ---- this workgroup's partial gradient: D column = lane&15 = tap, rows (lane>>4)*4 + reg = cs within the M tile
  float* pw = partial + (size_t)blockIdx.x * (CS * CB * 16);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NBW; ++n)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int cs = m * 16 + (lane >> 4) * 4 + reg, cb = wave * NBW + n;
        pw[(cs * CB + cb) * 16 + (lane & 15)] = acc[m][n][reg];
      }
  V2_ACC(5);
  V2_FLUSH();
}

// gw[e] (+)= sum over the workgroups' partial gradients.  A block = 8 float4 elements x 32 slices of the partials: with
// 256 partials every thread has its 8 loads in flight at once - the pass costs about one memory round trip (a thread
// that walks 32 partials one after the other made this kernel take 10 us).
// One more role of the reduce launches (pgv_bias_req): the block's bias gradient from the per-XCD partial copies its
// producer kept (pgv_bwd_fuse.gbias_copies) - workgroups behind the reduce (and border) ones, a thread per channel.
struct BiasFin {
  const float* copies; float* gbias; int C, accumulate;
};
__device__ __forceinline__ void bias_finish_role(const BiasFin& b, int blk) {
  const int c = blk * 256 + threadIdx.x;
  if (c >= b.C) return;
  float t = b.accumulate ? b.gbias[c] : 0.f;
#pragma unroll
  for (int r = 0; r < PGV_CLS_COPIES; ++r) t += b.copies[r * b.C + c];
  b.gbias[c] = t;
}
inline BiasFin bias_fin(const pgv_bias_req* b) {
  BiasFin f = {nullptr, nullptr, 0, 0};
  if (b) f.copies = b->copies, f.gbias = b->gbias, f.C = b->C, f.accumulate = b->accumulate;
  return f;
}
inline int bias_blocks(const pgv_bias_req* b) { return b ? (b->C + 255) / 256 : 0; }

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nparts, int n4,
                                                           float* __restrict__ gw, int accumulate, BiasFin bf, int nred) {
  if ((int)blockIdx.x >= nred) {
    bias_finish_role(bf, blockIdx.x - nred);
    return;
  }
  __shared__ f32x4 red[32][8];
  const int el = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int e = blockIdx.x * 8 + el;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(partial) + e;
    int k = sl;
    for (; k + 7 * 32 < nparts; k += 8 * 32) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + 32 * u) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nparts; k += 32) s += p[(size_t)k * n4];
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && e < n4) {
#pragma unroll
    for (int k = 1; k < 32; ++k) s += red[k][el];
    f32x4* o = reinterpret_cast<f32x4*>(gw) + e;
    if (accumulate) s += *o;
    *o = s;
  }
}

// The reduce pass above and the border tap sums of the block's output gradient (bn_taps.h) in ONE launch - two roles
// that share nothing but the launch (pgv_conv_wgrad_coef): each is a dependent launch of ~5 us otherwise, and the tap
// sums only need gy, not the weight gradient.
//   workgroups [0, nred): reduce role, as wgrad_reduce_kernel;
//   workgroups [nred, ...): border role, as tap_border_kernel of bn.hip (T[c][tap] += class sum - unpaired border sum).
// (Tried and dropped: also forming the coefficients here, S_o and S_1 accumulated with float64 atomics and finished by
// the last workgroup to arrive.  Same-address atomics from a thousand workgroups retire one per ~90 ns, the arrival
// counter alone cost 25 us, and a device-scope __threadfence() per workgroup - an L2 write-back on this chip - 100 us.)
struct WgradTapsArgs {
  const float* partial; int nparts, n4; float* gw; int accumulate, nred;
  const float* gy; int B, Cgy, H, W, per, nsplit, gy_is_big, s, p; TapBorder tb; const float* cls; double* T; int trep;
  int cls_copies, ntap; BiasFin bf;
};
template <int K>
__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(WgradTapsArgs a) {
  constexpr int KK = K * K;
  __shared__ f32x4 red[32][8];
  __shared__ float res[KK];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < a.nred) {
    const int el = tid & 7, sl = tid >> 3;
    const int e = blockIdx.x * 8 + el;
    f32x4 sv = {0.f, 0.f, 0.f, 0.f};
    if (e < a.n4) {
      const f32x4* pp = reinterpret_cast<const f32x4*>(a.partial) + e;
      int k = sl;
      for (; k + 7 * 32 < a.nparts; k += 8 * 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)(k + 32 * u) * a.n4];
#pragma unroll
        for (int u = 0; u < 8; ++u) sv += v[u];
      }
      for (; k < a.nparts; k += 32) sv += pp[(size_t)k * a.n4];
    }
    red[sl][el] = sv;
    __syncthreads();
    if (sl == 0 && e < a.n4) {
#pragma unroll
      for (int k = 1; k < 32; ++k) sv += red[k][el];
      f32x4* o = reinterpret_cast<f32x4*>(a.gw) + e;
      if (a.accumulate) sv += *o;
      *o = sv;
    }
  } else if ((int)blockIdx.x >= a.nred + a.ntap) {
    bias_finish_role(a.bf, blockIdx.x - a.nred - a.ntap);
  } else {
    const int bid = blockIdx.x - a.nred;
    const int c = bid % a.Cgy, r = bid / a.Cgy, by = r % a.nsplit, bz = r / a.nsplit;
    tap_border_block<K>(a.gy, a.B, a.Cgy, a.H, a.W, a.per, a.gy_is_big, a.s, a.p, a.tb, c, by, bz, res);
    if (tid < KK) {
      const int kh = tid / K, kw = tid - kh * K;
      double t = -(double)res[tid];
      if (by == 0 && bz == 0) {   // the class total enters once per channel
        const int m = a.gy_is_big ? a.s : 1;
        const int rho = a.gy_is_big ? (((kh - a.p) % a.s) + a.s) % a.s : 0, kap = a.gy_is_big ? (((kw - a.p) % a.s) + a.s) % a.s : 0;
        const int ncopy = a.cls_copies;   // (partial copies by XCD of the producers: pgv_bwd_fuse.cls / gbias_copies)
        for (int q = 0; q < ncopy; ++q) t += (double)a.cls[(q * a.Cgy + c) * m * m + rho * m + kap];
      }
      atomicAdd(&a.T[(int64_t)(r % a.trep) * a.Cgy * KK + (int64_t)c * KK + tid], t);   // (copies: tap_replicas)
    }
  }
}

// arguments of the border role (as tap_sums_launch of bn.hip sizes them); false when the border form does not apply
inline bool wgrad_taps_setup(const pgv_conv_desc* d, const pgv_coef_req* req, const float* big, const float* small_in,
                             WgradTapsArgs* a, int* nblocks) {
  const int gy_is_big = !req->lower_is_big;
  const int C = gy_is_big ? d->Cb : d->Cs, H = gy_is_big ? d->Hb : d->Hs, W = gy_is_big ? d->Wb : d->Ws;
  const int oH = gy_is_big ? d->Hs : d->Hb, oW = gy_is_big ? d->Ws : d->Wb;
  const int K = d->kh;
  if (!req->cls || d->kh != d->kw || (K != 4 && K != 5) || H > 1024 || W > 1024 || (gy_is_big && d->stride > 3)) return false;
  if (!tap_axis(gy_is_big != 0, H, K, d->stride, d->pad, oH, &a->tb.ra, &a->tb.rb, a->tb.rm) ||
      !tap_axis(gy_is_big != 0, W, K, d->stride, d->pad, oW, &a->tb.ca, &a->tb.cb, a->tb.cm))
    return false;
  const int NE = (a->tb.ra + a->tb.rb) * W + (H - a->tb.ra - a->tb.rb) * (a->tb.ca + a->tb.cb);
  const int nz = (int)max((int64_t)1, pgv_cdiv(NE, kTapChunk));
  a->per = (int)max((int64_t)1, min((int64_t)16, pgv_cdiv((int64_t)d->B * C * nz, 512)));
  a->nsplit = (int)pgv_cdiv(d->B, a->per);
  a->gy = gy_is_big ? big : small_in;
  a->B = d->B, a->Cgy = C, a->H = H, a->W = W, a->gy_is_big = gy_is_big, a->s = d->stride, a->p = d->pad;
  a->cls = req->cls, a->T = req->scratch, a->trep = tap_replicas(C, K * K);
  a->cls_copies = req->cls_copies > 0 ? req->cls_copies : (gy_is_big ? PGV_CLS_COPIES : 1);
  a->ntap = C * a->nsplit * nz;
  *nblocks = a->nred + a->ntap;
  return true;
}

// reduce + border launch; 1 when launched, 0 when the border form does not apply (the caller reduces on its own)
inline int launch_wgrad_reduce_taps(const pgv_conv_desc* d, const pgv_coef_req* req, const pgv_bias_req* bias,
                                    const float* big, const float* small_in, const float* partial, int nparts, int n4,
                                    float* gw, hipStream_t st) {
  WgradTapsArgs a;
  a.bf = bias_fin(bias);
  a.partial = partial, a.nparts = nparts, a.n4 = n4, a.gw = gw;
  a.accumulate = (d->flags & PGV_PREZEROED) ? 1 : 0, a.nred = (n4 + 7) / 8;
  int nblocks = 0;
  if (!wgrad_taps_setup(d, req, big, small_in, &a, &nblocks)) return 0;
  nblocks += bias_blocks(bias);
  if (d->kh == 4)
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel<4>, dim3(nblocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel<5>, dim3(nblocks), dim3(256), 0, st, a);
  PGV_CHECK_LAUNCH("conv_wgrad_v2 reduce + tap sums");
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// WGRAD of the 1 <-> 8 channel 5x5 layers (enc1 / dec8: big = [B,1,257,347], small = [B,8,129,174]), same structure as
// conv_wgrad_ws_kernel:  gw[cs][kh][kw] = sum_{b,oh,ow} small[b,cs,oh,ow] * big[b,0,2oh-2+kh,2ow-2+kw].
// M = (cs, column shift s) = 16 rows, N = (kh, kernel column 2..4) = 15 of 16 columns, K = output pixels: the shifted copy
// of the small operand turns kernel columns 2..4 into 0..2, so ONE 16x16 tile holds the 200 gradients (round 6; before:
// M = cs in 8 of 16 rows, N = 25 taps in two tiles - twice the matrix instructions, whose issue starved the loader
// wave of the same SIMD).  The four MFMA waves split K: wave w
// multiplies output row w of the unit (R = 4 rows) and keeps its own accumulators; every wave's partial sum goes to the
// workspace (4 per workgroup) and the reduce pass adds them.  The band kernel this replaces spends 37 % of a
// workgroup's time in its MFMA phase (commit 1.4 us + issue 1.0 us against 1.6 us of MFMAs per item).
// ---------------------------------------------------------------------------------------------------------------
template <int CS, int W, int H, int R>
struct Wgrad5Cfg {
  static constexpr int KS = 5, NTAP = 25;
  static constexpr int Ws = W / 2 + 1, Hs = H / 2 + 1;
  static constexpr int BANDS = (Hs + R - 1) / R;
  static constexpr int ROWS_B = 2 * (R - 1) + KS;
  // big-tile row stride = 8 mod 32 floats: the kernel rows of a B fragment (7 consecutive floats each per 32-lane half)
  // fall on disjoint bank ranges
  static constexpr int WP = (W + 2 - 8 + 31) / 32 * 32 + 8;
  static constexpr int WsP = (Ws + 3) / 4 * 4;
  // plane stride of the small tile = 4 mod 32 floats: the A fragment reads the same pixel of all 8 channels at once -
  // with the planes back to back (704 floats) that is an 8-way bank conflict on every read, and this kernel has only
  // 3 operand reads per 4 MFMAs to hide it behind (MFMA-side time 74 us instead of ~40)
  static constexpr int PLANE_S = R * WsP + 4;
  static_assert(PLANE_S % 32 == 4, "bank spreading");
  static constexpr int SPR = WsP / 4, PPR = SPR / 2;  // k-steps / pairs of k-steps per output row
  static constexpr int FRONT = 4;
  static constexpr int BUF = FRONT + ROWS_B * WP + CS * PLANE_S;
  static constexpr size_t LDS_FLOATS = 2 * (size_t)BUF + 2 * (1 + CS) + 8;
  static_assert(R == 4 && CS == 8 && SPR % 2 == 0 && WP >= W + 2 && 2 * WsP <= WP, "tiling");
};

// BF16: PGV_COMPUTE_BF16 - both operands rounded to bfloat16 where the loader commits them (products of rounded operands
// are exact on the fp32 matrix pipe; this 1-channel layer has little matrix work)
template <int CS, int W, int H, int R, bool AFF_S, bool BF16 = false>
__global__ __launch_bounds__(512, 2) void conv_wgrad5_ws_kernel(int B, const float* __restrict__ big,
                                                              const float* __restrict__ small_in,
                                                              const float* __restrict__ small_scale,
                                                              const float* __restrict__ small_shift,
                                                              float* __restrict__ partial) {
  using G = Wgrad5Cfg<CS, W, H, R>;
  constexpr int Ws = G::Ws, Hs = G::Hs, BANDS = G::BANDS, WP = G::WP, WsP = G::WsP, PLANE_S = G::PLANE_S;
  constexpr int SPR = G::SPR, PPR = G::PPR, BUF = G::BUF, PLANE_B = G::ROWS_B * WP;
  using StageB = StageLean<1, G::ROWS_B, W, WP, H>;
  using StageS = StageLean<CS, R, Ws, WsP, Hs, false, PLANE_S>;
  constexpr int NPB = StageB::NPF, NPS = StageS::NPF;
  static_assert(NPB + NPS < 64, "vmcnt range");
  static_assert(8 * (SPR - 2) + 7 + 4 < W + 2 && 4 * (SPR - 1) <= Ws, "tail masking");
  static_assert(BANDS >= 3 && (BANDS - 2) * 2 * R - 2 + G::ROWS_B <= H && (BANDS - 1) * R <= Hs, "edge bands");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile0 = lds + G::FRONT;
  float* aff_s = lds + 2 * BUF;  // [2][CS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int units = B * BANDS;
  const int bid = pgv_xcd_block();
  const int my_items = bid < units ? (units - bid + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (tid < 2 * G::FRONT) lds[(tid / G::FRONT) * BUF + tid % G::FRONT] = 0.f;
  if (AFF_S)
    for (int i = tid; i < CS; i += 512) aff_s[i] = small_scale[i], aff_s[CS + i] = small_shift[i];
  __syncthreads();
  if (wave >= 4) {
    // ================================================= loader waves =================================================
    if (my_items == 0) return;
    const int ltid = tid - 256;
    __builtin_amdgcn_s_setprio(PGV_V2_PRIO_LOADER);
    typename StageB::Geo geoB;
    typename StageS::Geo geoS;
    typename StageB::Set bA, bB;
    typename StageS::Set sA, sB;
    const int64_t bytes_b = (int64_t)B * (H * W) * 4, bytes_s = (int64_t)B * CS * (Hs * Ws) * 4;
    auto band_of = [&](int it, int& b, int& band) {
      it = min(it, my_items - 1);
      const int u = bid + it * gridDim.x;
      b = u / BANDS;
      band = u - b * BANDS;
    };
    auto is_edge = [&](int band) { return band == 0 || band == BANDS - 1; };
    auto first_item = [&](auto stage_is_big, auto jc) {  // the loads of item 0, slot by slot out of the set-up
      constexpr int J = decltype(jc)::value;
      int b, band;
      band_of(0, b, band);
      if constexpr (decltype(stage_is_big)::value) {
        const i32x4 rb = StageB::band_rsrc(big, bytes_b, ((int64_t)b * H + band * 2 * R - 2) * W);
        StageB::template issue_slot<J, true>(geoB, bA, rb, band == 0 ? geoB.top_bad : (band == BANDS - 1 ? geoB.bot_bad : 0u));
      } else {
        const i32x4 rs = StageS::band_rsrc(small_in, bytes_s, ((int64_t)b * CS * Hs + band * R) * Ws);
        StageS::template issue_slot<J, true>(geoS, sA, rs, band == BANDS - 1 ? geoS.bot_bad : 0u);
      }
    };
    geoB.init(ltid, nullptr, 1, false, 2, H - ((BANDS - 1) * 2 * R - 2), [&](auto jc) { first_item(std::true_type{}, jc); });
    geoS.init(ltid, aff_s, CS, AFF_S, 0, Hs - (BANDS - 1) * R, [&](auto jc) { first_item(std::false_type{}, jc); });
    auto issue_all = [&](typename StageB::Set& bx, typename StageS::Set& sx, int it) {
      int b, band;
      band_of(it, b, band);
      const i32x4 rb = StageB::band_rsrc(big, bytes_b, ((int64_t)b * H + band * 2 * R - 2) * W);
      const i32x4 rs = StageS::band_rsrc(small_in, bytes_s, ((int64_t)b * CS * Hs + band * R) * Ws);
      if (is_edge(band)) {
        const unsigned badb = band == 0 ? geoB.top_bad : geoB.bot_bad, bads = band == 0 ? 0u : geoS.bot_bad;
        static_for<0, NPB>([&](auto j) { StageB::template issue_slot<decltype(j)::value, true>(geoB, bx, rb, badb); });
        static_for<0, NPS>([&](auto j) { StageS::template issue_slot<decltype(j)::value, true>(geoS, sx, rs, bads); });
      } else {
        static_for<0, NPB>([&](auto j) { StageB::template issue_slot<decltype(j)::value, false>(geoB, bx, rb, 0u); });
        static_for<0, NPS>([&](auto j) { StageS::template issue_slot<decltype(j)::value, false>(geoS, sx, rs, 0u); });
      }
    };
    auto commit_all = [&](const typename StageB::Set& bx, const typename StageS::Set& sx, int it, float* dst) {
      int b, band;
      band_of(it, b, band);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB + NPS) : "memory");  // the older item has landed
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NPB>([&](auto j) { StageB::template commit_slot<decltype(j)::value, false, false, BF16>(geoB, bx, dst, ltid, 0u); });
      if (AFF_S && band == BANDS - 1) {
        static_for<0, NPS>([&](auto j) {
          StageS::template commit_slot<decltype(j)::value, true, AFF_S, BF16>(geoS, sx, dst + PLANE_B, ltid, geoS.bot_bad);
        });
      } else {
        static_for<0, NPS>([&](auto j) {
          StageS::template commit_slot<decltype(j)::value, false, AFF_S, BF16>(geoS, sx, dst + PLANE_B, ltid, 0u);
        });
      }
    };
    issue_all(bB, sB, 1);  // (item 0 went out during the set-up)
    commit_all(bA, sA, 0, tile0);
    issue_all(bA, sA, 2);
    ws_barrier();
#pragma unroll 1
    for (int it = 0; it < my_items; it += 2) {
#ifndef PGV_W5_NO_LOAD
      __builtin_amdgcn_s_sleep(PGV_V2_LOADER_SLEEP);
      commit_all(bB, sB, it + 1, tile0 + BUF);
      issue_all(bB, sB, it + 3);
#endif
      ws_barrier();
      if (it + 1 < my_items) {
#ifndef PGV_W5_NO_LOAD
        __builtin_amdgcn_s_sleep(PGV_V2_LOADER_SLEEP);
        commit_all(bA, sA, it + 2, tile0);
        issue_all(bA, sA, it + 4);
#endif
        ws_barrier();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // ==================================================== MFMA waves ===================================================
  __builtin_amdgcn_s_setprio(PGV_V2_PRIO_MFMA);
  // ONE 16x16 tile holds all 200 gradients: wave w multiplies output row w of the unit with
  //   M row m = (cs = m & 7, s = m >> 3):  A[m][k] = small[cs][w][v + s]            (the same pixel row, one column on)
  //   N col n = (kh = n / 3, kwn = 2 + n % 3), n < 15:  B[k][n] = big[2w + kh][2v + kwn - 2]
  // so that D[(cs, s)][(kh, kwn)] = sum_v small[cs][w][v + s] * big[2w + kh][2(v + s) - 2 + (kwn - 2s)] is the gradient of
  // tap (kh, kw = kwn - 2s): s = 0 gives kernel columns 2..4, s = 1 columns 0..2 (column 2 twice: the s = 1 copy is not
  // stored).  One MFMA per 4 pixels instead of two with half-empty tiles (M = 8 channels, N = 25 taps in two tiles).
  // K slot k = lane >> 4 of step i is pixel v = 4i + k.  The s = 1 rows see pixels 1 .. Ws of the row; pixel 0 (v = -1)
  // rides in the last step's slot k = 3, whose own pixel (4 SPR - 1) lies behind the row end.
  const int kq = lane >> 4, sh = (lane >> 3) & 1;
  const int tapn = min(lane & 15, 14), khn = tapn / 3, kwn = 2 + tapn % 3;
  const int offA = PLANE_B + (lane & 7) * PLANE_S + wave * WsP + kq + sh;
  const int offB = (2 * wave + khn) * WP + kwn - 2 + 2 * kq;
  static_assert(4 * (SPR - 1) + 3 >= Ws, "slot 3 of the last step is free for pixel -1");
  // last step: which operand lanes hold real pixels (the loader does not clean the floats behind a row end)
  const bool keepA = 4 * (SPR - 1) + kq + sh < Ws, keepB = 8 * (SPR - 1) + 2 * kq + kwn - 2 < W;
  const int offA_m1 = PLANE_B + (lane & 7) * PLANE_S + wave * WsP;         // pixel v + s = 0 of the row
  const int offB_m1 = (2 * wave + khn) * WP + kwn - 4;                      // columns -2 .. 0 (left zero padding)
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (my_items > 0) {
    ws_barrier();  // item 0 committed
#pragma unroll 1
    for (int it = 0; it < my_items; ++it) {
      const float* cur = tile0 + (it & 1) * BUF;
      const int u = bid + it * gridDim.x;
      const int nrows = min(R, Hs - (u % BANDS) * R);  // the last band of a sample is short
#ifdef PGV_W5_NO_MFMA
      if (false) {
#else
      if (wave < nrows) {
#endif
        float av[3][2], bv[3][2];
        auto load_pair = [&](int ps, float (&a)[2], float (&b)[2]) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            a[h] = cur[offA + 4 * (2 * ps + h)];
            b[h] = cur[offB + 8 * (2 * ps + h)];
          }
          if (2 * ps + 1 == SPR - 1) {
            const float a0 = cur[offA_m1], b0 = cur[offB_m1];
            a[1] = kq == 3 ? (sh ? a0 : 0.f) : (keepA ? a[1] : 0.f);
            b[1] = kq == 3 ? b0 : (keepB ? b[1] : 0.f);
          }
        };
        load_pair(0, av[0], bv[0]);
        load_pair(1, av[1], bv[1]);
        static_for<0, PPR>([&](auto pc) {
          constexpr int ps = decltype(pc)::value, pn = ps + 2;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (pn < PPR) load_pair(pn, av[pn % 3], bv[pn % 3]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc = PGV_MFMA4(av[ps % 3][h], bv[ps % 3][h], acc);
          // 2 operand reads (one per operand: the two k-steps of a pair come from one ds_read2) under 2 MFMAs
          static_for<0, 2>([&](auto kc) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (pn < PPR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          });
        });
        __builtin_amdgcn_sched_barrier(0);
      }
      ws_barrier();
    }
  }
  // ---- this wave's partial gradient: D column = lane & 15 = (kh, kwn), rows (lane>>4)*4 + reg = (cs, s)
  float* pw = partial + ((size_t)blockIdx.x * 4 + wave) * (CS * G::NTAP);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int m = (lane >> 4) * 4 + reg, cs = m & 7, s1 = m >> 3;
    if ((lane & 15) < 15 && !(s1 && kwn == 4)) pw[cs * G::NTAP + khn * G::KS + kwn - 2 * s1] = acc[reg];
  }
}

template <int R>
int launch_wgrad5_v2(const pgv_conv_desc* d, const float* big, const float* small_in, const float* small_scale,
                     const float* small_shift, float* gw, void* workspace, int64_t workspace_bytes,
                     const pgv_coef_req* req, const pgv_bias_req* bias, hipStream_t st) {
  using G = Wgrad5Cfg<8, 347, 257, R>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  const int units = d->B * G::BANDS;
  const int grid = min(units, 256);
  const int nparts = 4 * grid;
  const int64_t need = (int64_t)nparts * 8 * G::NTAP * sizeof(float);
  if (!workspace || workspace_bytes < need || ((uintptr_t)gw & 15) || ((uintptr_t)workspace & 15)) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, float*);
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  kern_t kern = small_scale ? (bf16 ? (kern_t)conv_wgrad5_ws_kernel<8, 347, 257, R, true, true>
                                    : (kern_t)conv_wgrad5_ws_kernel<8, 347, 257, R, true>)
                            : (bf16 ? (kern_t)conv_wgrad5_ws_kernel<8, 347, 257, R, false, true>
                                    : (kern_t)conv_wgrad5_ws_kernel<8, 347, 257, R, false>);
  if (int rc = raise_lds_once((const void*)kern, "conv_wgrad5_v2")) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), bytes, st, d->B, big, small_in, small_scale, small_shift, (float*)workspace);
  PGV_CHECK_LAUNCH("conv_wgrad5_v2");
  const int n4 = 8 * G::NTAP / 4;
  if (req) {   // reduce + the tap sums of the output gradient in one launch (3: the caller goes on with the coefficients)
    const int rc = launch_wgrad_reduce_taps(d, req, bias, big, small_in, (const float*)workspace, nparts, n4, gw, st);
    if (rc) return rc < 0 ? rc : 3;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n4 + 7) / 8 + bias_blocks(bias)), dim3(256), 0, st, (const float*)workspace,
                     nparts, n4, gw, (d->flags & PGV_PREZEROED) ? 1 : 0, bias_fin(bias), (n4 + 7) / 8);
  PGV_CHECK_LAUNCH("conv_wgrad5_v2 reduce");
  return 1;
}

template <int CB, int CS, int W, int H, int R>
int launch_wgrad_v2(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                    const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                    void* workspace, int64_t workspace_bytes, const pgv_coef_req* req, const pgv_bias_req* bias,
                    hipStream_t st) {
  using G = WgradV2Cfg<CB, CS, W, H, R>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  if (d->Cb != CB || d->Cs != CS) return 0;
  const int units = d->B * G::BANDS;
  const int grid = min(units, 256);
  const int64_t need = (int64_t)grid * CS * CB * 16 * sizeof(float);
  if (!workspace || workspace_bytes < need || ((uintptr_t)gw & 15) || ((uintptr_t)workspace & 15)) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, const float*, const float*, float*);
  if (big_scale && small_scale) return 0;  // not a case of the train step (the loader would spill registers)
  kern_t kern = big_scale ? (kern_t)conv_wgrad_ws_kernel<CB, CS, W, H, R, true, false>
                          : (small_scale ? (kern_t)conv_wgrad_ws_kernel<CB, CS, W, H, R, false, true>
                                         : (kern_t)conv_wgrad_ws_kernel<CB, CS, W, H, R, false, false>);
  if (int rc = raise_lds_once((const void*)kern, "conv_wgrad_v2")) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), bytes, st, d->B, big, big_scale, big_shift, small_in, small_scale,
                     small_shift, (float*)workspace);
  PGV_CHECK_LAUNCH("conv_wgrad_v2");
  const int n4 = CS * CB * 16 / 4;
  if (req) {   // reduce + the tap sums of the output gradient in one launch (3: the caller goes on with the coefficients)
    const int rc = launch_wgrad_reduce_taps(d, req, bias, big, small_in, (const float*)workspace, grid, n4, gw, st);
    if (rc) return rc < 0 ? rc : 3;
  }
  // PGV_PREZEROED: gw holds zeros or an earlier partial sum to add to; otherwise it is overwritten
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n4 + 7) / 8 + bias_blocks(bias)), dim3(256), 0, st, (const float*)workspace,
                     grid, n4, gw, (d->flags & PGV_PREZEROED) ? 1 : 0, bias_fin(bias), (n4 + 7) / 8);
  PGV_CHECK_LAUNCH("conv_wgrad_v2 reduce");
  return 1;
}

}  // namespace

// workspace: one partial gradient per workgroup
int64_t pgv_conv_wgrad_v2_workspace(const pgv_conv_desc* d) {
  if (d->stride == 2 && d->pad == 2 && d->kh == 5 && d->kw == 5 && d->Cb == 1 &&
      d->Cs == 8 && d->Hb == 257 && d->Wb == 347)
    return (int64_t)4 * 256 * 8 * 25 * sizeof(float);  // one partial gradient per MFMA wave
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4) return 0;
  // (bf16 operand mode: the band kernels leave up to 512 partial gradients - two workgroups per CU - for the same reduce)
  if ((d->Hb == 33 && d->Wb == 45 && d->Cb == 32 && d->Cs == 64) ||
      (d->Hb == 65 && d->Wb == 88 && d->Cb == 16 && d->Cs == 32) ||
      (d->Hb == 129 && d->Wb == 174 && d->Cb == 8 && d->Cs == 16))
    return (int64_t)((d->flags & PGV_COMPUTE_BF16) ? 512 : 256) * d->Cs * d->Cb * 16 * sizeof(float);
  return 0;
}

__global__ __launch_bounds__(256) void bias_finish_kernel(BiasFin bf) { bias_finish_role(bf, blockIdx.x); }
int pgv_bias_finish(const pgv_bias_req* bias, hipStream_t st) {
  if (!bias || bias->C <= 0) return PGV_OK;
  hipLaunchKernelGGL(bias_finish_kernel, dim3(bias_blocks(bias)), dim3(256), 0, st, bias_fin(bias));
  PGV_CHECK_LAUNCH("bias_finish");
  return PGV_OK;
}

// debugging / A-B knob: bit 0 = bf16 operand mode stays on the band kernels (conv_band.hip) instead of conv_wgrad_bf16.hip
static int g_wgrad_bf16_variant = 0;
extern "C" int pgv_dbg_set_wgrad_bf16_variant(int v) {
  const int old = g_wgrad_bf16_variant;
  g_wgrad_bf16_variant = v;
  return old;
}

int pgv_conv_wgrad_v2(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                      const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                      void* workspace, int64_t workspace_bytes, const pgv_coef_req* req, const pgv_bias_req* bias,
                      hipStream_t st) {
  if (d->stride == 2 && d->pad == 2 && d->kh == 5 && d->kw == 5 && d->B > 0 &&
      d->Cb == 1 && d->Cs == 8 && d->Hb == 257 && d->Wb == 347 && !big_scale)
    return launch_wgrad5_v2<4>(d, big, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, req, bias, st);
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4) return 0;
  if (d->B <= 0) return 0;
  // (the large-plane kernels of conv_wgrad_split.hip: fp32 products as six bf16 instructions, or bf16 operand mode with one plane)
  const bool split = (pgv_big_split_shape(d) || pgv_big_bf16q_shape(d)) && workspace && !((uintptr_t)gw & 15) &&
                     !((uintptr_t)workspace & 15) && !(big_scale && small_scale);
  if ((d->flags & PGV_COMPUTE_BF16) || split) {
    // bf16 operand mode: the band kernels (v_mfma_f32_16x16x32_bf16) leave per-workgroup partial gradients in the workspace
    // and the reduce launch of this file adds them up - with the tap sums / bias roles the fp32 step folds into it.
    // PGV_COMPUTE_F32_SPLIT: the same contract, products as six bf16 instructions (conv_wgrad_split.hip)
    if (!workspace || ((uintptr_t)gw & 15) || ((uintptr_t)workspace & 15)) return 0;
    int nparts = 0;
    int rc = 0;
    if (split)
      rc = pgv_conv_wgrad_split_partial(d, big, big_scale, big_shift, small_in, small_scale, small_shift, (float*)workspace,
                                        workspace_bytes, &nparts, st);
    if (rc == 0 && (d->flags & PGV_COMPUTE_BF16)) {   // (bf16 operand mode: the round-4 kernels, for what the above did not take)
      rc = (g_wgrad_bf16_variant & 1) ? 0
                                      : pgv_conv_wgrad_bf16_partial(d, big, big_scale, big_shift, small_in, small_scale,
                                                                    small_shift, (float*)workspace, workspace_bytes, &nparts, st);
      if (rc == 0)
        rc = pgv_conv_wgrad_band_partial(d, big, big_scale, big_shift, small_in, small_scale, small_shift, (float*)workspace,
                                         workspace_bytes, &nparts, st);
    }
    if (rc < 0) return rc;
    if (rc > 0) {
      const int n4 = d->Cs * d->Cb * 16 / 4;
      if (req) {
        rc = launch_wgrad_reduce_taps(d, req, bias, big, small_in, (const float*)workspace, nparts, n4, gw, st);
        if (rc) return rc < 0 ? rc : 3;
      }
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n4 + 7) / 8 + bias_blocks(bias)), dim3(256), 0, st, (const float*)workspace,
                         nparts, n4, gw, (d->flags & PGV_PREZEROED) ? 1 : 0, bias_fin(bias), (n4 + 7) / 8);
      PGV_CHECK_LAUNCH("conv_wgrad_band reduce");
      return 1;
    }
    if (d->flags & PGV_COMPUTE_BF16) return 0;
    // fp32, six-instruction form not applicable to this call (workspace smaller than one partial gradient per workgroup, or a
    // tensor of 2 GB and more - B >= ~3000 on 129x174): the native wave-specialised kernels below, not the generic fallbacks
  }
  if (d->Hb == 33 && d->Wb == 45)
    return launch_wgrad_v2<32, 64, 45, 33, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw,
                                              workspace, workspace_bytes, req, bias, st);
  if (d->Hb == 65 && d->Wb == 88)
    return launch_wgrad_v2<16, 32, 88, 65, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw,
                                              workspace, workspace_bytes, req, bias, st);
  if (d->Hb == 129 && d->Wb == 174)
    return launch_wgrad_v2<8, 16, 174, 129, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw,
                                               workspace, workspace_bytes, req, bias, st);
  return 0;
}
