// Shape-specialised band kernels for the stride-2 k=4 layers of the reference tables (model/encoder.py:241-255,
// model/decoder.py:205-218) at the reference spectrogram size (257x347 -> 129x174 -> 65x88 -> 33x45 -> 17x23).
//
// Same implicit GEMM on v_mfma_f32_16x16x4_f32 as conv_mfma.hip, but every tile dimension is a template constant:
//   * the LDS tile keeps image rows at a padded stride WP (multiple of 4, >= W+2): the pad columns are stored as
//     zeros, so the column masks (one v_cndmask + hazard nop in front of every MFMA) disappear, and
//   * every LDS address of the MFMA loop is  per-lane base + immediate offset : no address arithmetic is left in
//     the loop, which is fully unrolled and software-pipelined (reads of step s+1 issued before the MFMAs of step s).
// Measured on MI355X (scratch/ubench/mfma_loop.hip): 34 clk per MFMA from one wave per SIMD, against 74 clk for
// the runtime-stride loop of conv_mfma.hip (the matrix pipe needs 32).
// Shapes that are not instantiated here fall through to conv_mfma.hip (return 0).
#include <stdlib.h>
#include <type_traits>
#include "conv_tile.h"
#include "band_prefetch.h"

#ifndef PGV_BAND_U
#define PGV_BAND_U 4  // 16-byte loads in flight per lane while staging a band
#endif

#ifdef PGV_PHASE_TIMING
extern "C" int pgv_dbg_set_tlog_band(void* p);
__device__ unsigned long long* pgv_tlog_band = nullptr;
extern "C" int pgv_dbg_set_tlog_band(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(pgv_tlog_band), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
// per-wave accumulated phase durations (100 MHz ticks): slot i = time spent before BAND_ACC(i) since the previous
// stamp, summed over the work items of a persistent workgroup; slot 7 = number of items
#define BAND_T0() unsigned long long band_tp = wall_clock64(), band_sum[7] = {0, 0, 0, 0, 0, 0, 0}, band_n = 0
#define BAND_ACC(i)                                   \
  do {                                                \
    const unsigned long long now = wall_clock64();    \
    band_sum[i] += now - band_tp;                     \
    band_tp = now;                                    \
  } while (0)
#define BAND_ITEM() (++band_n)
#define BAND_FLUSH()                                                                                         \
  do {                                                                                                       \
    if ((threadIdx.x & 63) == 0 && pgv_tlog_band) {                                                          \
      unsigned long long* o = pgv_tlog_band + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;            \
      for (int i = 0; i < 7; ++i) o[i] = band_sum[i];                                                        \
      o[7] = band_n;                                                                                         \
    }                                                                                                        \
  } while (0)
#else
#define BAND_T0()
#define BAND_ACC(i)
#define BAND_ITEM()
#define BAND_FLUSH()
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------
// DOWN (Conv2d forward / ConvTranspose2d input-gradient), k = 4, stride 2, pad 2.
//   D[cs][pixel] = sum_{c,kh,kw} W[cs][c][kh][kw] * X[c][2r+kh-2][2col+kw-2]
// MT M-tiles of 16 output channels, NT N-tiles of 16 pixels per wave, CK input channels per LDS chunk, R output rows
// per (sample, band) unit, W/H = input image size.  Workgroups are persistent: work items are (unit, channel chunk)
// pairs, the loads of item i+1 are in flight while item i is multiplied (and, on the last chunk of a unit, while
// its epilogue runs).
// ---------------------------------------------------------------------------------------------------------------
template <int MT, int NT, int CK, int NCH, bool WRES, int R, int W, int H>
struct DownCfg {
  static constexpr int KS = 4;
  static constexpr int Ws = (W + 4 - KS) / 2 + 1, Hs = (H + 4 - KS) / 2 + 1;
  static constexpr int ROWS = 2 * (R - 1) + KS;
  static constexpr int WP = (W + 2 + 3) / 4 * 4;
  static constexpr int PLANE = ROWS * WP;
  static constexpr int CSP = MT * 16 + 1;  // odd: conflict-free transposing writes, <= 2-way conflicts on the A reads
  static constexpr int KC = CK * KS * 4;  // k rows per chunk
  static constexpr int PS = 64 * NT + 4;  // out tile row stride
  static constexpr int EM = MT > 2 ? 2 : MT;
  static constexpr int TILE = (CK * PLANE > EM * 16 * PS ? CK * PLANE : EM * 16 * PS);
  static constexpr int FRONT = 4;  // zero slack in front of the tile (col -2,-1 of row 0 of channel 0)
  static constexpr int WT = (WRES ? NCH : 1) * KC * CSP;  // weight tile: all chunks resident, or the current one
  static constexpr int NWQ = KC * MT * 16 / 4 / 256;       // 16-byte weight loads per lane and chunk
  static constexpr size_t LDS_FLOATS = FRONT + TILE + WT + 4 * MT * 16 * 2 + 2 * NCH * CK;
  static constexpr int BANDS = (Hs + R - 1) / R;
  static_assert(R * Ws <= 64 * NT, "band does not fit the wave tiles");
  static_assert((KC * MT * 16) % 1024 == 0, "weight chunk must split into whole 16-byte loads per lane");
};

template <int MT, int NT, int CK, int NCH, bool WRES, int R, int W, int H, bool FUSE, bool BF16, bool CLS = false>
__global__ __launch_bounds__(256, 2) void conv_down_band_kernel(int B, int Cb, int Cs, const float* __restrict__ big,
                                                              const float* __restrict__ in_scale,
                                                              const float* __restrict__ in_shift,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias, int act, float slope,
                                                              float* __restrict__ out, double* __restrict__ stats,
                                                              pgv_bwd_fuse fuse) {
  using G = DownCfg<MT, NT, CK, NCH, WRES, R, W, H>;
  constexpr int KS = 4, Ws = G::Ws, Hs = G::Hs, WP = G::WP, PLANE = G::PLANE, CSP = G::CSP, KC = G::KC;
  constexpr int PS = G::PS, EM = G::EM, BANDS = G::BANDS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* in_tile = lds + G::FRONT;
  float* w_tile = in_tile + G::TILE;
  float* st_tile = w_tile + G::WT;         // [4 waves][MT*16][2]
  float* aff = st_tile + 4 * MT * 16 * 2;  // [2][Cb]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int units = B * BANDS;
  const int nchunk = (Cb + CK - 1) / CK;

  // per-lane B base: pixel (r, c) of tile t, tap kw = lane>>4:  (2r)*WP + 2c - 2 + kw   (+ c*PLANE + kh*WP immediates)
  int offB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wave * NT + t) * 16 + (lane & 15);
    const int pv = p < R * Ws ? p : 0;
    const int r = pv / Ws, c = pv - r * Ws;
    offB[t] = 2 * r * WP + 2 * c - 2 + (lane >> 4);
  }
  const int offA = (lane >> 4) * CSP + (lane & 15);

  typename PickPrefetch<CK, G::ROWS, W, WP, H>::type pf;
  pf.init(tid);
  if (tid < G::FRONT) lds[tid] = 0.f;
  for (int i = tid; i < 4 * MT * 16 * 2; i += 256) st_tile[i] = 0.f;
  stage_affine(aff, in_scale, in_shift, Cb, tid);  // visible after the first barrier of the item loop
  const pgv_act_params actp = pgv_act_setup(act, slope);
  const pgv_actd_params actd = pgv_actd_setup(FUSE ? fuse.act : 0, FUSE ? fuse.slope : 0.f);

  // weights: 16-byte loads of 4 consecutive k = (c, kh, kw0..3) of one output channel, transposed to [k][cs] in LDS.
  // WRES: every chunk is staged once per workgroup; otherwise the chunk of the next item is prefetched into
  // registers together with its band.
  constexpr int NWQ = G::NWQ;
  f32x4 wv[WRES ? 1 : NWQ];
  auto load_weights = [&](int cb0, f32x4 (&dst)[NWQ]) {
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
      const int idx = (tid + j * 256) * 4;
      const int cs = idx / KC, k = idx - cs * KC;
      const bool ok = cs < Cs && cb0 + (k >> 4) < Cb;
      const float* g = w + ((int64_t)(ok ? cs : 0) * Cb + (ok ? cb0 : 0)) * 16 + (ok ? k : 0);
      const f32x4 t = *reinterpret_cast<const f32x4*>(g);
      dst[j] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_weights = [&](float* wt, const f32x4 (&src)[NWQ]) {
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
      const int idx = (tid + j * 256) * 4;
      const int cs = idx / KC, k = idx - cs * KC;
#pragma unroll
      for (int i = 0; i < 4; ++i) wt[(k + i) * CSP + cs] = src[j][i];
    }
  };
  float bias_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int cl = m * 16 + (lane >> 4) * 4 + reg;
      bias_r[m][reg] = (bias && cl < Cs) ? bias[cl] : 0.f;
    }
  auto issue_item = [&](int u, int ch) {
    const int b = u / BANDS, band = u - b * BANDS;
    const int ih0 = band * R * 2 - 2;
    pf.issue(big + ((int64_t)b * Cb + ch * CK) * (H * W), ih0, Cb - ch * CK);
    if constexpr (!WRES) load_weights(ch * CK, wv);
  };

  constexpr bool kRegStats = MT <= 2;
  float st_s[kRegStats ? MT : 1][4], st_q[kRegStats ? MT : 1][4];
#pragma unroll
  for (int m = 0; m < (kRegStats ? MT : 1); ++m)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) st_s[m][reg] = st_q[m][reg] = 0.f;
  int u = pgv_xcd_block();
  if (u >= units) return;
  BAND_T0();
#ifdef PGV_SETPRIO
  __builtin_amdgcn_s_setprio(3);
#endif
  issue_item(u, 0);
  if constexpr (WRES) {
    for (int c = 0; c < nchunk; ++c) {
      f32x4 tmp[NWQ];
      load_weights(c * CK, tmp);
      store_weights(w_tile + c * KC * CSP, tmp);
    }
  }
  f32x4 acc[MT][NT];
  int ch = 0;
#pragma unroll 1
  while (true) {
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();  // previous item's MFMA reads / epilogue copy of the tile region are complete
    BAND_ACC(0);
    pf.commit(in_tile, in_scale ? aff : nullptr, Cb, ch * CK, Cb - ch * CK, tid);
    BAND_ACC(1);
    if constexpr (!WRES) store_weights(w_tile, wv);
    const float* wt = WRES ? w_tile + ch * KC * CSP : w_tile;
    BAND_ACC(2);
    // next item: next chunk of this unit, else first chunk of this workgroup's next unit
    int nu = u, nch = ch + 1;
    if (nch == nchunk) {
      nch = 0;
      nu = u + gridDim.x;
    }
    if (nu < units) issue_item(nu, nch);
    BAND_ACC(3);
    __syncthreads();
    BAND_ACC(4);
#ifdef PGV_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // ragged last band (e.g. 17 output rows = 8 + 8 + 1): a wave whose pixel tiles all lie beyond the band's rows has
    // nothing to multiply - it leaves the matrix pipe of its SIMD to the co-resident workgroup
    const bool wave_active = wave * NT * 16 < min(R, G::Hs - (u % BANDS) * R) * G::Ws;
    if (!wave_active) {
    } else if constexpr (!BF16) {
      constexpr int S = CK * KS;  // steps (c, kh); one MFMA per (step, m, t) consumes the 4 kw taps
      float a0[MT], a1[MT], b0[NT], b1[NT];
      auto load_step = [&](int st, float (&av)[MT], float (&bv)[NT]) {
        const int c = st / KS, kh = st - c * KS;
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = wt[st * 4 * CSP + m * 16 + offA];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = in_tile[c * PLANE + kh * WP + offB[t]];
      };
      auto compute_step = [&](const float (&av)[MT], const float (&bv)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], acc[m][t], 0, 0, 0);
      };
      load_step(0, a0, b0);
#pragma unroll
      for (int st = 0; st < S; st += 2) {
        load_step(st + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < S) load_step(st + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // bf16 compute (PGV_COMPUTE_BF16): one bf16 MFMA step per (channel, m, t) - the lane group supplies its
      // kw tap for the four kernel rows kh = 0..3 (the operands of four fp32 steps), rounded to bf16 while packing
      constexpr int S = CK;
      u32x4 a0[MT], b0[NT];
      auto load_step = [&](auto half_c, int c, u32x4 (&av)[MT], u32x4 (&bv)[NT]) {
        constexpr int HALF = decltype(half_c)::value;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float* ap = wt + c * 16 * CSP + m * 16 + offA;
          set_half<HALF>(av[m], pack_bf16x4(ap[0], ap[4 * CSP], ap[8 * CSP], ap[12 * CSP]));
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float* bp = in_tile + c * PLANE + offB[t];
          set_half<HALF>(bv[t], pack_bf16x4(bp[0], bp[WP], bp[2 * WP], bp[3 * WP]));
        }
      };
      // two 16-deep steps per v_mfma_f32_16x16x32_bf16 (conv_tile.h); pairs are double-buffered where the registers allow
      // it: the operands of pair p + 1 are read and packed while pair p multiplies.  The kernels with large tiles (33x45
      // planes) would spill: they read a pair, multiply it, read the next - the co-resident workgroup covers the latency.
      auto compute_pair = [&](const u32x4 (&a8)[MT], const u32x4 (&b8)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][t] = mfma_bf16_k32(a8[m], b8[t], acc[m][t]);
      };
      auto load_pair = [&](int st, u32x4 (&a8)[MT], u32x4 (&b8)[NT]) {
        load_step(std::integral_constant<int, 0>{}, st, a8, b8);
        if (st + 1 < S) {
          load_step(std::integral_constant<int, 1>{}, st + 1, a8, b8);
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m) a8[m][2] = a8[m][3] = 0u;
#pragma unroll
          for (int t = 0; t < NT; ++t) b8[t][2] = b8[t][3] = 0u;
        }
      };
      constexpr bool DB = MT + NT <= 6;
      if constexpr (DB) {
        u32x4 a1[MT], b1[NT];
        load_pair(0, a0, b0);
#pragma unroll
        for (int st = 0; st < S; st += 4) {
          if (st + 2 < S) load_pair(st + 2, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          compute_pair(a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 4 < S) load_pair(st + 4, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 2 < S) compute_pair(a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int st = 0; st < S; st += 2) {
          load_pair(st, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          compute_pair(a0, b0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#ifdef PGV_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    BAND_ACC(5);
    BAND_ITEM();
    if (ch == nchunk - 1) {
      // ---- epilogue: bias + activation, statistics, band through LDS, 16-byte stores of contiguous NCHW segments
      const int b = u / BANDS, band = u - b * BANDS;
      const int oh0 = band * R;
      const int Pb = min(R, Hs - oh0) * Ws;
      float* out_tile = in_tile;
#pragma unroll
      for (int m0 = 0; m0 < MT; m0 += EM) {
        constexpr int BN_IT = (16 * NT * EM * 16 + 255) / 256;
        f4u bn_av[FUSE ? BN_IT : 1];
        if constexpr (FUSE) {  // saved activation of this channel group: in flight while the band goes through LDS
          const int nchf = min(Cs - m0 * 16, EM * 16);
          const int64_t offf = ((int64_t)b * Cs + m0 * 16) * Hs * Ws + (int64_t)oh0 * Ws;
          bnred_fetch<EM * 16, BN_IT>(fuse.a + offf, (int64_t)Hs * Ws, nchf, Pb, tid, bn_av);
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < EM; ++mm) {
          const int m = m0 + mm;
          float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int cl = m * 16 + (lane >> 4) * 4 + reg;
            const float bv = bias_r[m][reg];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int p = (wave * NT + t) * 16 + (lane & 15);
              const float v = pgv_act_apply(acc[m][t][reg] + bv, actp);
              out_tile[(cl - m0 * 16) * PS + p] = v;
              const float vm = p < Pb ? v : 0.f;
              s[reg] += vm;
              q[reg] = fmaf(vm, vm, q[reg]);
            }
          }
          if (stats) {
            if constexpr (kRegStats) {  // per-lane partial sums live in registers over all units of the workgroup
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                st_s[m][reg] += s[reg];
                st_q[m][reg] += q[reg];
              }
            } else {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                const float ss = group16_sum(s[reg]), qq = group16_sum(q[reg]);
                if ((lane & 15) == 0) {
                  const int cl = m * 16 + (lane >> 4) * 4 + reg;
                  st_tile[(wave * MT * 16 + cl) * 2 + 0] += ss;  // slot owned by this lane: no atomics needed
                  st_tile[(wave * MT * 16 + cl) * 2 + 1] += qq;
                }
              }
            }
          }
        }
        __syncthreads();
        const int nchn = min(Cs - m0 * 16, EM * 16);
        if (nchn > 0) {
          const int64_t off = ((int64_t)b * Cs + m0 * 16) * Hs * Ws + (int64_t)oh0 * Ws;
          if constexpr (FUSE)  // BatchNorm + activation backward of the lower block on the way out (pgv_bwd_fuse)
            store_rows_bwd<EM * 16, BN_IT, CLS ? Ws : 0>(out_tile, PS, out + off, fuse.a + off, (int64_t)Hs * Ws, nchn, Pb, tid,
                                                         fuse.coef + m0 * 16, Cs, actd, st_tile + (CLS ? 4 : 1) * m0 * 16,
                                                         bn_av, oh0);
          else
            store_rows_contig(out_tile, PS, out + off, (int64_t)Hs * Ws, nchn, Pb, tid);
        }
      }
      BAND_ACC(6);
    }
    u = nu;
    ch = nch;
    if (u >= units) break;
  }
  // BatchNorm statistics: the per-wave partial sums of all units of this workgroup sit in st_tile; one float64 atomic
  // per channel per WORKGROUP (per unit they serialise on 2*Cs addresses: +24 us on the 129x174 layer)
  if (stats) {
    if constexpr (kRegStats) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float ss = group16_sum(st_s[m][reg]), qq = group16_sum(st_q[m][reg]);
          if ((lane & 15) == 0) {
            const int cl = m * 16 + (lane >> 4) * 4 + reg;
            st_tile[(wave * MT * 16 + cl) * 2 + 0] = ss;
            st_tile[(wave * MT * 16 + cl) * 2 + 1] = qq;
          }
        }
    }
    __syncthreads();
    if (tid < MT * 16 && tid < Cs) {
      double ss = 0.0, qq = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        ss += (double)st_tile[(wv * MT * 16 + tid) * 2 + 0];
        qq += (double)st_tile[(wv * MT * 16 + tid) * 2 + 1];
      }
      atomicAdd(&stats[tid], ss);
      atomicAdd(&stats[Cs + tid], qq);
    }
  }
  if constexpr (FUSE) {  // (never together with stats: the launcher falls back to a separate pass then)
    __syncthreads();
    if constexpr (CLS) {   // class sums (pgv_bwd_fuse.cls) and, from them, the bias gradient
      static_assert(Ws % 4 == 0, "class sums");
      if (tid < MT * 16 && tid < Cs) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = st_tile[4 * tid + k];
          atomicAdd(&fuse.cls[4 * tid + k], v);
          t += v;
        }
        if (fuse.gbias) atomicAdd(&fuse.gbias[(fuse.gbias_copies ? (blockIdx.x & (PGV_CLS_COPIES - 1)) * Cs : 0) + tid], t);
      }
    } else {
      if (fuse.gbias && tid < MT * 16 && tid < Cs)
        atomicAdd(&fuse.gbias[(fuse.gbias_copies ? (blockIdx.x & (PGV_CLS_COPIES - 1)) * Cs : 0) + tid], st_tile[tid]);
    }
  }
  BAND_FLUSH();
}

template <int MT, int NT, int CK, int NCH, bool WRES, int R, int W, int H, bool CANFUSE>
int launch_down_band(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* out, double* stats,
                     const pgv_bwd_fuse* fuse, hipStream_t st) {
  using G = DownCfg<MT, NT, CK, NCH, WRES, R, W, H>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  if (d->Cb > NCH * CK) return 0;
  // The fused-epilogue variant keeps the saved-activation loads in registers: only instantiated where it pays
  // (CANFUSE), and only launched when asked for, so the plain kernel's register allocation is unaffected.
  if (fuse && !CANFUSE) return 0;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  auto kern = conv_down_band_kernel<MT, NT, CK, NCH, WRES, R, W, H, false, false>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_down_band");
  if (rc) return rc;
  auto kernf = conv_down_band_kernel<MT, NT, CK, NCH, WRES, R, W, H, CANFUSE, false>;
  if (CANFUSE) {
    static bool attr_done_f = false;
    if ((rc = raise_lds_limit(kernf, &attr_done_f, "conv_down_band"))) return rc;
  }
  auto kernb = conv_down_band_kernel<MT, NT, CK, NCH, WRES, R, W, H, false, true>;
  static bool attr_done_b = false;
  if ((rc = raise_lds_limit(kernb, &attr_done_b, "conv_down_band"))) return rc;
  auto kernfb = conv_down_band_kernel<MT, NT, CK, NCH, WRES, R, W, H, CANFUSE, true>;
  if (CANFUSE) {
    static bool attr_done_fb = false;
    if ((rc = raise_lds_limit(kernfb, &attr_done_fb, "conv_down_band"))) return rc;
  }
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_band: memset failed");
    return PGV_E_LAUNCH;
  }
  const int units = d->B * G::BANDS;
  int per_cu = (int)min((size_t)2, (size_t)kMaxLds / bytes);
#ifdef PGV_PHASE_TIMING
  if (getenv("PGV_WG_PER_CU")) per_cu = atoi(getenv("PGV_WG_PER_CU"));
#endif
  const int grid = min(units, 256 * per_cu);
  const pgv_bwd_fuse fz = {nullptr, nullptr, nullptr, 0, 0.f, nullptr};
  // class sums of the written tensor (pgv_bwd_fuse.cls) ride in the store pass where its rows are whole 16-byte pieces
  if constexpr (CANFUSE && G::Ws % 4 == 0) {
    if (fuse && fuse->cls && !bf16) {
      auto kernc = conv_down_band_kernel<MT, NT, CK, NCH, WRES, R, W, H, CANFUSE, false, true>;
      static bool attr_done_c = false;
      if ((rc = raise_lds_limit(kernc, &attr_done_c, "conv_down_band"))) return rc;
      hipLaunchKernelGGL(kernc, dim3(grid), dim3(256), bytes, st, d->B, d->Cb, d->Cs, big, in_scale, in_shift, w, bias,
                         act, slope, out, stats, *fuse);
      PGV_CHECK_LAUNCH("conv_down_band");
      return 3;   // handled, class sums included
    }
  }
  hipLaunchKernelGGL(bf16 ? (fuse ? kernfb : kernb) : (fuse ? kernf : kern), dim3(grid), dim3(256), bytes, st, d->B,
                     d->Cb, d->Cs, big,
                     in_scale, in_shift, w, bias, act, slope, out, stats, fuse ? *fuse : fz);
  PGV_CHECK_LAUNCH("conv_down_band");
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// WGRAD, k = 4, stride 2, pad 2:  gw[cs][cb][kh][kw] = sum_{b,oh,ow} small[b,cs,oh,ow] * big[b,cb,2oh-2+kh,2ow-2+kw]
// GEMM with M = cs (MT tiles of 16), N = (cb, 16 taps) = one N tile per big channel, K = output pixels, 4 consecutive
// ow per MFMA.  A[cs][pixel] from the small tile (row stride WsP, zero pad columns), B[pixel][tap] straight from the raw
// big tile: lane (tap j, pixel k) reads  cb*PLANE_B + (2r+kh)*WP + 2*(ow0+k) + kw - 2.
// Waves split the N tiles WN ways and the rows of the band WK = 4/WN ways (R = WK*RW rows per unit); the (row, step)
// loop of a wave is fully unrolled, so all LDS addresses are per-lane base + immediate.  Persistent workgroups keep
// the accumulators in registers over all their (sample, band) units and flush once with float atomics; the tiles of
// unit i+1 are prefetched into registers while unit i is multiplied.
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int MT, int CSL, int CB, int NSPLIT, int WN, int RW, int W, int H, bool BF16>
struct WgradCfg {
  static constexpr int KK = KS * KS;
  static constexpr int NTAP = (KK + 15) / 16;  // N tiles per big channel
  static constexpr int NB = CB * NTAP / WN;    // N tiles per wave
  static constexpr int WK = 4 / WN;
  static constexpr int R = WK * RW;
  static constexpr int CS = MT * 16;           // M rows; rows >= CSL alias existing channels and are discarded
  static constexpr int Ws = (W + 4 - KS) / 2 + 1, Hs = (H + 4 - KS) / 2 + 1;
  static constexpr int ROWS_B = 2 * (R - 1) + KS;
  static constexpr int WP = (W + 2 + 3) / 4 * 4;
  // bf16 compute: 16 output pixels per MFMA (four fp32 k-steps merged), so rows are padded to a multiple of 16
  static constexpr int WsP = BF16 ? (Ws + 15) / 16 * 16 : (Ws + 3) / 4 * 4;
  // channel stride of the small tile: the A operand is read at  cs*PLANE_S + pixel  by 16 channels x 4 pixels per wave,
  // stride % 64 == 4 spreads them over all 64 LDS banks (R*WsP itself is 0 or 32 mod 64: 8- to 16-way conflicts)
  static constexpr int PLANE_B = ROWS_B * WP, PLANE_S = (R * WsP - 4 + 63) / 64 * 64 + 4;
  static constexpr int SPR = WsP / 4;  // MFMA k-steps per output row
  static constexpr int FRONT = 4;
  static constexpr int TILES = CB * PLANE_B + CSL * PLANE_S;
  static constexpr int RED = WK > 1 ? WK * CS * CB * NTAP * 16 : 0;  // cross-wave reduction buffer of the final flush
  static constexpr int BODY = TILES > RED ? TILES : RED;
  static constexpr size_t LDS_FLOATS = FRONT + BODY + 2 * (CB + CSL);
  static constexpr int BANDS = (Hs + R - 1) / R;
  // right-most B read of a real pixel: 2*(Ws-1) + KS-1 - 2 must hit image or zero pad columns of the same row
  static_assert(2 * (Ws - 1) + KS - 3 < WP, "row stride");
  static_assert((CB * NTAP) % WN == 0 && (CSL & (CSL - 1)) == 0 && CSL <= CS, "tiling");
};

template <int KS, int MT, int CSL, int CB, int NSPLIT, int WN, int RW, int W, int H, bool BF16>
__global__ __launch_bounds__(256, 2) void conv_wgrad_band_kernel(int B, int Cb, int Cs, const float* __restrict__ big,
                                                               const float* __restrict__ big_scale,
                                                               const float* __restrict__ big_shift,
                                                               const float* __restrict__ small_in,
                                                               const float* __restrict__ small_scale,
                                                               const float* __restrict__ small_shift,
                                                               float* __restrict__ gw, float* __restrict__ partial) {
  using G = WgradCfg<KS, MT, CSL, CB, NSPLIT, WN, RW, W, H, BF16>;
  constexpr int KK = G::KK, NTAP = G::NTAP, NB = G::NB, WK = G::WK, R = G::R, CS = G::CS, Ws = G::Ws, Hs = G::Hs;
  constexpr int WP = G::WP, WsP = G::WsP, PLANE_B = G::PLANE_B, PLANE_S = G::PLANE_S, SPR = G::SPR, BANDS = G::BANDS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* big_tile = lds + G::FRONT;
  float* small_tile = big_tile + CB * PLANE_B;
  float* aff_b = big_tile + G::BODY;  // [2][Cb]
  float* aff_s = aff_b + 2 * CB;      // [2][Cs]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gn = wave % WN, wk = wave / WN;
  const int units = B * BANDS;
  // NSPLIT > 1: the big channels are divided among workgroups (workgroup g keeps split g % NSPLIT for all its units),
  // which divides the register-resident accumulators and the final atomic flush by NSPLIT at the price of reading
  // the small tensor NSPLIT times
  const int split = blockIdx.x % NSPLIT, cb0 = split * CB;
  const int ncb = min(CB, Cb - cb0);  // big channels of this split

  // per-lane bases (rows r = wk + WK*rw and steps i are immediates); taps >= KK of the last tap tile read tap 0 and
  // are discarded at the flush
  int offB[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int nt = gn * NB + n;
    const int cb = nt / NTAP, tau = (nt - cb * NTAP) * 16 + (lane & 15);
    const int kh = tau < KK ? tau / KS : 0, kw = tau < KK ? tau - (tau / KS) * KS : 0;
    offB[n] = cb * PLANE_B + (2 * wk + kh) * WP + kw - 2 + 2 * (lane >> 4);
  }
  int offA[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) offA[m] = ((m * 16 + (lane & 15)) & (CSL - 1)) * PLANE_S + wk * WsP + (lane >> 4);

  f32x4 acc[MT][NB];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  typename PickPrefetch<CB, G::ROWS_B, W, WP, H>::type pfb;
  typename PickPrefetch<CSL, R, Ws, WsP, Hs, 0, PLANE_S>::type pfs;
  pfb.init(tid);
  pfs.init(tid);
  if (tid < G::FRONT) lds[tid] = 0.f;
  if (ncb > 0) stage_affine(aff_b, big_scale ? big_scale + cb0 : nullptr, big_shift ? big_shift + cb0 : nullptr, ncb, tid);
  stage_affine(aff_s, small_scale, small_shift, Cs, tid);
  auto issue_unit = [&](int u) {
    const int b = u / BANDS, band = u - b * BANDS;
    pfb.issue(big + ((int64_t)b * Cb + cb0) * (H * W), band * R * 2 - 2, ncb);
    pfs.issue(small_in + (int64_t)b * Cs * (Hs * Ws), band * R, Cs);
  };

  const int ustep = gridDim.x / NSPLIT;  // the launcher makes the grid a multiple of NSPLIT
  int u = NSPLIT == 1 ? pgv_xcd_block() : blockIdx.x / NSPLIT;  // consecutive units on one XCD (conv_tile.h)
  BAND_T0();
  if (u < units) issue_unit(u);
#pragma unroll 1
  for (; u < units; u += ustep) {
    __syncthreads();  // the previous unit's MFMA reads are complete (and the affine tables are visible)
    BAND_ACC(0);
    pfb.commit(big_tile, big_scale ? aff_b : nullptr, max(ncb, 1), 0, ncb, tid);
    pfs.commit(small_tile, small_scale ? aff_s : nullptr, Cs, 0, Cs, tid);
    BAND_ACC(1);
    if (u + ustep < units) issue_unit(u + ustep);
    BAND_ACC(3);
    __syncthreads();
    BAND_ACC(4);
    if constexpr (!BF16) {
      constexpr int S = RW * SPR;
      float a0[MT], a1[MT], b0[NB], b1[NB];
      auto load_step = [&](int st, float (&av)[MT], float (&bv)[NB]) {
        const int rw = st / SPR, i = st - rw * SPR;
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = small_tile[offA[m] + rw * WK * WsP + 4 * i];
#pragma unroll
        for (int n = 0; n < NB; ++n) bv[n] = big_tile[offB[n] + rw * WK * 2 * WP + 8 * i];
      };
      auto compute_step = [&](const float (&av)[MT], const float (&bv)[NB]) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
      };
      load_step(0, a0, b0);
#pragma unroll
      for (int st = 0; st < S; st += 2) {
        if (st + 1 < S) load_step(st + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < S) load_step(st + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < S) compute_step(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // bf16 compute: 32 output pixels per v_mfma_f32_16x16x32_bf16 (two 16-pixel steps paired in one operand, conv_tile.h) -
      // the lane group keeps its pixel offset k and supplies pixels k, k+4, k+8, k+12 of each 16-pixel step (the operands
      // of four fp32 steps), rounded while packing
      constexpr int SPR4 = SPR / 4;
      static_assert(SPR % 4 == 0, "bf16 rows are padded to 16 pixels");
      constexpr int S = RW * SPR4;
      u32x4 a0[MT], b0[NB];
      auto load_step = [&](auto half_c, int st, u32x4 (&av)[MT], u32x4 (&bv)[NB]) {
        constexpr int HALF = decltype(half_c)::value;
        const int rw = st / SPR4, i = st - rw * SPR4;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float* ap = small_tile + offA[m] + rw * WK * WsP + 16 * i;
          set_half<HALF>(av[m], pack_bf16x4(ap[0], ap[4], ap[8], ap[12]));
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const float* bp = big_tile + offB[n] + rw * WK * 2 * WP + 32 * i;
          set_half<HALF>(bv[n], pack_bf16x4(bp[0], bp[8], bp[16], bp[24]));
        }
      };
      // two 16-deep steps per v_mfma_f32_16x16x32_bf16 (conv_tile.h); pairs are double-buffered where the registers allow
      // it: the operands of pair p + 1 are read and packed while pair p multiplies.  The kernels with large tiles (33x45
      // planes) would spill: they read a pair, multiply it, read the next - the co-resident workgroup covers the latency.
      auto compute_pair = [&](const u32x4 (&a8)[MT], const u32x4 (&b8)[NB]) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = mfma_bf16_k32(a8[m], b8[n], acc[m][n]);
      };
      auto load_pair = [&](int st, u32x4 (&a8)[MT], u32x4 (&b8)[NB]) {
        load_step(std::integral_constant<int, 0>{}, st, a8, b8);
        if (st + 1 < S) {
          load_step(std::integral_constant<int, 1>{}, st + 1, a8, b8);
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m) a8[m][2] = a8[m][3] = 0u;
#pragma unroll
          for (int n = 0; n < NB; ++n) b8[n][2] = b8[n][3] = 0u;
        }
      };
      constexpr bool DB = MT + NB <= 6;
      if constexpr (DB) {
        u32x4 a1[MT], b1[NB];
        load_pair(0, a0, b0);
#pragma unroll
        for (int st = 0; st < S; st += 4) {
          if (st + 2 < S) load_pair(st + 2, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          compute_pair(a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 4 < S) load_pair(st + 4, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 2 < S) compute_pair(a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int st = 0; st < S; st += 2) {
          load_pair(st, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          compute_pair(a0, b0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    BAND_ACC(5);
    BAND_ITEM();
  }
  BAND_FLUSH();
  // ---- flush: D col = lane&15 = tap within the tap tile, row = (lane>>4)*4 + reg = cs within the M tile.  Waves that
  // split the rows of the band (WK > 1) hold partial sums of the same elements: they are added up through LDS first,
  // so the workgroup issues one coalesced float atomic per weight element.
  // ``partial`` (pgv_conv_wgrad_band_partial): the workgroup's sums go to slot blockIdx.x / NSPLIT of a workspace in the
  // layout of gw, with plain stores, and the reduce launch of the wave-specialised path (conv_v2_wgrad.hip) adds the
  // slots up.  Measured on the bf16 operand mode (phase stamps, scratch/phase_wgrad_bf16.py): the item loops of the
  // 65x88 / 33x45 kernels took 29 / 27 us of 60 / 72 us launches - the rest was this flush as float atomics, 512
  // workgroups x up to 128 KB = 67 MB of atomic traffic at the ~1.3 TB/s the memory side retires them.
  float* const dst = partial ? partial + (size_t)(blockIdx.x / NSPLIT) * ((size_t)Cs * Cb * KK) : gw;
  constexpr int NTT = CB * NTAP;  // N tiles in total
  if constexpr (WK > 1) {
    __syncthreads();
    float* red = big_tile;  // [WK][CS][NTT][16]
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int cs = m * 16 + (lane >> 4) * 4 + reg, nt = gn * NB + n;
          red[((wk * CS + cs) * NTT + nt) * 16 + (lane & 15)] = acc[m][n][reg];
        }
    __syncthreads();
    for (int e = tid; e < CS * NTT * 16; e += 256) {
      const int cs = e / (NTT * 16), rem = e - cs * (NTT * 16), nt = rem >> 4;
      const int cb = nt / NTAP, tau = (nt - cb * NTAP) * 16 + (rem & 15);
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < WK; ++k) v += red[k * CS * NTT * 16 + e];
      if (cs < Cs && cb < ncb && tau < KK) {
        float* o = &dst[((int64_t)cs * Cb + cb0 + cb) * KK + tau];
        if (partial) *o = v;
        else atomicAdd(o, v);
      }
    }
  } else {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int nt = gn * NB + n;
      const int cb = nt / NTAP, tau = (nt - cb * NTAP) * 16 + (lane & 15);
      if (cb < ncb && tau < KK) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int cs = m * 16 + (lane >> 4) * 4 + reg;
            if (cs < Cs) {
              float* o = &dst[((int64_t)cs * Cb + cb0 + cb) * KK + tau];
              if (partial) *o = acc[m][n][reg];
              else atomicAdd(o, acc[m][n][reg]);
            }
          }
      }
    }
  }
}

template <int KS, int MT, int CSL, int CB, int NSPLIT, int WN, int RW, int W, int H, bool BF16>
int launch_wgrad_band_t(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                        const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                        hipStream_t st, float* partial = nullptr, int64_t partial_bytes = 0, int* nparts = nullptr) {
  using G = WgradCfg<KS, MT, CSL, CB, NSPLIT, WN, RW, W, H, BF16>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  if (d->Cb > CB * NSPLIT || d->Cs > CSL) return 0;
  auto kern = conv_wgrad_band_kernel<KS, MT, CSL, CB, NSPLIT, WN, RW, W, H, BF16>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_wgrad_band");
  if (rc) return rc;
  const int units = d->B * G::BANDS;
  int per_cu = (int)min((size_t)2, (size_t)kMaxLds / bytes);
#ifdef PGV_PHASE_TIMING
  if (getenv("PGV_WGRAD_PER_CU")) per_cu = atoi(getenv("PGV_WGRAD_PER_CU"));
#endif
  const int grid = min(units, 256 * per_cu / NSPLIT) * NSPLIT;
  if (partial) {   // one slot of the workspace per group of NSPLIT workgroups (they write disjoint channel ranges of it)
    if (units == 0 || d->Cb != CB * NSPLIT || (int64_t)(grid / NSPLIT) * d->Cs * d->Cb * G::KK * (int64_t)sizeof(float) > partial_bytes)
      return 0;
    *nparts = grid / NSPLIT;
  } else if (!(d->flags & PGV_PREZEROED) &&
             hipMemsetAsync(gw, 0, sizeof(float) * (size_t)d->Cs * d->Cb * G::KK, st) != hipSuccess) {
    pgv_set_error("conv_wgrad_band: memset failed");
    return PGV_E_LAUNCH;
  }
  if (units == 0) return 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, st, d->B, d->Cb, d->Cs, big, big_scale, big_shift, small_in,
                     small_scale, small_shift, gw, partial);
  PGV_CHECK_LAUNCH("conv_wgrad_band");
  return 1;
}

template <int KS, int MT, int CSL, int CB, int NSPLIT, int WN, int RW, int W, int H>
int launch_wgrad_band(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                      const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                      hipStream_t st) {
  if (d->flags & PGV_COMPUTE_BF16)
    return launch_wgrad_band_t<KS, MT, CSL, CB, NSPLIT, WN, RW, W, H, true>(d, big, big_scale, big_shift, small_in,
                                                                          small_scale, small_shift, gw, st);
  return launch_wgrad_band_t<KS, MT, CSL, CB, NSPLIT, WN, RW, W, H, false>(d, big, big_scale, big_shift, small_in,
                                                                         small_scale, small_shift, gw, st);
}

// ---------------------------------------------------------------------------------------------------------------
// UP (ConvTranspose2d forward / Conv2d input-gradient), k = 4, stride 2, pad 2, by sub-pixel phases:
//   out[cb][2u+ph][2v+pw] = sum_{cs,th,tw} w[cs][cb][ph+2th][pw+2tw] * X[cs][u+1-th][v+1-tw]
// i.e. D[(cb,ph,pw)][(u,v)] with M = 4*Cb rows (MT tiles), one MFMA per input channel (k = th*2+tw), the same input
// gather for all four phases.  The grid of (u,v) is Hg x Wg = ceil(H/2) x ceil(W/2) for an H x W output; a unit is R
// grid rows of one sample (2R output rows).  Small tile: R+1 rows at stride WsP >= Ws+1 (zero pad: the v+1 read of the
// last grid column when W is odd).  The 2R x W output band of EM*4 channels at a time goes through LDS and leaves as
// 16-byte stores of contiguous NCHW segments.
// ---------------------------------------------------------------------------------------------------------------
template <int MT, int NT, int CK, int NCH, bool WRES, int EM, int R, int W, int H>
struct UpCfg {
  static constexpr int Ws = W / 2 + 1, Hs = H / 2 + 1;        // input (small) size
  static constexpr int Wg = (W + 1) / 2, Hg = (H + 1) / 2;    // sub-pixel grid
  static constexpr int ROWS = R + 1;
  static constexpr int WsP = (Ws + 1 + 3) / 4 * 4;
  static constexpr int PLANE = ROWS * WsP;
  static constexpr int MSP = MT * 16 + 1;
  static constexpr int KC = CK * 4;                            // k rows per chunk
  static constexpr int OPS = (2 * R * W + 3) / 4 * 4;          // out tile channel stride
  static constexpr int TILE = (CK * PLANE > EM * 4 * OPS ? CK * PLANE : EM * 4 * OPS);
  static constexpr int WT = (WRES ? NCH : 1) * KC * MSP;
  static constexpr int NWQ = KC * MT * 16 / 4 / 256;           // 16-byte weight loads per lane and chunk
  static constexpr size_t LDS_FLOATS = TILE + WT + 4 * MT * 4 * 2 + 2 * NCH * CK;
  static constexpr int BANDS = (Hg + R - 1) / R;
  static_assert(R * Wg <= 64 * NT, "band does not fit the wave tiles");
  static_assert(MT % EM == 0, "epilogue passes");
  static_assert((KC * MT * 16) % 1024 == 0, "weight chunk must split into whole 16-byte loads per lane");
};

template <int MT, int NT, int CK, int NCH, bool WRES, int EM, int R, int W, int H, bool FUSE, bool BF16>
__global__ __launch_bounds__(256, 2) void conv_up_band_kernel(int B, int Cb, int Cs, const float* __restrict__ small_in,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, int act, float slope,
                                                            float* __restrict__ out, double* __restrict__ stats,
                                                            pgv_bwd_fuse fuse) {
  using G = UpCfg<MT, NT, CK, NCH, WRES, EM, R, W, H>;
  constexpr int Ws = G::Ws, Hs = G::Hs, Wg = G::Wg, Hg = G::Hg, WsP = G::WsP, PLANE = G::PLANE, MSP = G::MSP;
  constexpr int KC = G::KC, OPS = G::OPS, BANDS = G::BANDS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* in_tile = lds;
  float* w_tile = in_tile + G::TILE;
  float* st_tile = w_tile + G::WT;        // [4 waves][MT*4][2]
  float* aff = st_tile + 4 * MT * 4 * 2;  // [2][Cs]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int units = B * BANDS;
  const int nchunk = (Cs + CK - 1) / CK;

  // per-lane B base: grid pixel (ur, v) of tile t, tap k = lane>>4 = th*2+tw:  (ur+1-th)*WsP + v+1-tw
  int offB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wave * NT + t) * 16 + (lane & 15);
    const int pv = p < R * Wg ? p : 0;
    const int ur = pv / Wg, v = pv - ur * Wg;
    const int k = lane >> 4;
    offB[t] = (ur + 1 - (k >> 1)) * WsP + v + 1 - (k & 1);
  }
  const int offA = (lane >> 4) * MSP + (lane & 15);

  typename PickPrefetch<CK, G::ROWS, Ws, WsP, Hs, 1>::type pf;
  pf.init(tid);
  for (int i = tid; i < 4 * MT * 4 * 2; i += 256) st_tile[i] = 0.f;
  stage_affine(aff, in_scale, in_shift, Cs, tid);  // visible after the first barrier of the item loop
  const pgv_act_params actp = pgv_act_setup(act, slope);
  const pgv_actd_params actd = pgv_actd_setup(FUSE ? fuse.act : 0, FUSE ? fuse.slope : 0.f);

  // weights: 16-byte loads of one kernel row (cs, cb, kh, kw0..3), scattered to [k = (c, th, tw)][m = (cb, ph, pw)]
  constexpr int NWQ = G::NWQ;
  f32x4 wv[WRES ? 1 : NWQ];
  auto load_weights = [&](int cs0, f32x4 (&dst)[NWQ]) {
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
      const int idx = tid + j * 256;  // (c, cb, kh)
      const int kh = idx & 3, cb = (idx >> 2) % (MT * 4), c = idx / (MT * 16);
      const bool ok = cs0 + c < Cs && cb < Cb;
      const float* g = w + (((int64_t)(ok ? cs0 + c : 0) * Cb + (ok ? cb : 0)) * 4 + kh) * 4;
      const f32x4 t = *reinterpret_cast<const f32x4*>(g);
      dst[j] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_weights = [&](float* wt, const f32x4 (&src)[NWQ]) {
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
      const int idx = tid + j * 256;
      const int kh = idx & 3, cb = (idx >> 2) % (MT * 4), c = idx / (MT * 16);
      const int ph = kh & 1, th = kh >> 1;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int pw = kw & 1, tw = kw >> 1;
        wt[(c * 4 + th * 2 + tw) * MSP + cb * 4 + ph * 2 + pw] = src[j][kw];
      }
    }
  };
  float bias_r[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int cb = m * 4 + (lane >> 4);
    bias_r[m] = (bias && cb < Cb) ? bias[cb] : 0.f;
  }
  auto issue_item = [&](int u, int ch) {
    const int b = u / BANDS, band = u - b * BANDS;
    pf.issue(small_in + ((int64_t)b * Cs + ch * CK) * (Hs * Ws), band * R, Cs - ch * CK);
    if constexpr (!WRES) load_weights(ch * CK, wv);
  };

  constexpr bool kRegStats = MT <= 4;
  float st_s[kRegStats ? MT : 1], st_q[kRegStats ? MT : 1];
#pragma unroll
  for (int m = 0; m < (kRegStats ? MT : 1); ++m) st_s[m] = st_q[m] = 0.f;
  int u = pgv_xcd_block();
  if (u >= units) return;
  BAND_T0();
  issue_item(u, 0);
  if constexpr (WRES) {
    for (int c = 0; c < nchunk; ++c) {
      f32x4 tmp[NWQ];
      load_weights(c * CK, tmp);
      store_weights(w_tile + c * KC * MSP, tmp);
    }
  }
  f32x4 acc[MT][NT];
  int ch = 0;
#pragma unroll 1
  while (true) {
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();  // previous item's MFMA reads / epilogue copy of the tile region are complete
    BAND_ACC(0);
    pf.commit(in_tile, in_scale ? aff : nullptr, Cs, ch * CK, Cs - ch * CK, tid);
    BAND_ACC(1);
    if constexpr (!WRES) store_weights(w_tile, wv);
    const float* wt = WRES ? w_tile + ch * KC * MSP : w_tile;
    BAND_ACC(2);
    int nu = u, nch = ch + 1;
    if (nch == nchunk) {
      nch = 0;
      nu = u + gridDim.x;
    }
    if (nu < units) issue_item(nu, nch);
    BAND_ACC(3);
    __syncthreads();
    BAND_ACC(4);
    // ragged last band (e.g. 17 grid rows = 8 + 8 + 1): a wave whose pixel tiles all lie beyond the band's rows has
    // nothing to multiply - it leaves the matrix pipe of its SIMD to the co-resident workgroup
    const bool wave_active = wave * NT * 16 < min(R, Hg - (u % BANDS) * R) * Wg;
    if (!wave_active) {
    } else if constexpr (!BF16) {
      constexpr int S = CK;  // one k-group (th, tw) per input channel
      static_assert(S % 2 == 0, "step count must be even");
      float a0[MT], a1[MT], b0[NT], b1[NT];
      auto load_step = [&](int st, float (&av)[MT], float (&bv)[NT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = wt[st * 4 * MSP + m * 16 + offA];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = in_tile[st * PLANE + offB[t]];
      };
      auto compute_step = [&](const float (&av)[MT], const float (&bv)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], acc[m][t], 0, 0, 0);
      };
      load_step(0, a0, b0);
#pragma unroll
      for (int st = 0; st < S; st += 2) {
        load_step(st + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < S) load_step(st + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // bf16 compute: one v_mfma_f32_16x16x32_bf16 per (8 input channels, m, t) - two 4-channel steps paired in one operand
      // (conv_tile.h): the lane group keeps its tap (th, tw) and supplies it for channels 4s..4s+3 of each step (the operands
      // of four fp32 steps), rounded to bf16 while packing
      constexpr int S = CK / 4;
      static_assert(CK % 8 == 0, "channel chunk must hold an even number of 4-channel steps");
      u32x4 a0[MT], b0[NT];
      auto load_step = [&](auto half_c, int st, u32x4 (&av)[MT], u32x4 (&bv)[NT]) {
        constexpr int HALF = decltype(half_c)::value;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float* ap = wt + st * 16 * MSP + m * 16 + offA;
          set_half<HALF>(av[m], pack_bf16x4(ap[0], ap[4 * MSP], ap[8 * MSP], ap[12 * MSP]));
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float* bp = in_tile + st * 4 * PLANE + offB[t];
          set_half<HALF>(bv[t], pack_bf16x4(bp[0], bp[PLANE], bp[2 * PLANE], bp[3 * PLANE]));
        }
      };
      // two 16-deep steps per v_mfma_f32_16x16x32_bf16 (conv_tile.h); pairs are double-buffered where the registers allow
      // it: the operands of pair p + 1 are read and packed while pair p multiplies.  The kernels with large tiles (33x45
      // planes) would spill: they read a pair, multiply it, read the next - the co-resident workgroup covers the latency.
      auto compute_pair = [&](const u32x4 (&a8)[MT], const u32x4 (&b8)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][t] = mfma_bf16_k32(a8[m], b8[t], acc[m][t]);
      };
      auto load_pair = [&](int st, u32x4 (&a8)[MT], u32x4 (&b8)[NT]) {
        load_step(std::integral_constant<int, 0>{}, st, a8, b8);
        if (st + 1 < S) {
          load_step(std::integral_constant<int, 1>{}, st + 1, a8, b8);
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m) a8[m][2] = a8[m][3] = 0u;
#pragma unroll
          for (int t = 0; t < NT; ++t) b8[t][2] = b8[t][3] = 0u;
        }
      };
      constexpr bool DB = MT + NT <= 6;
      if constexpr (DB) {
        u32x4 a1[MT], b1[NT];
        load_pair(0, a0, b0);
#pragma unroll
        for (int st = 0; st < S; st += 4) {
          if (st + 2 < S) load_pair(st + 2, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          compute_pair(a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 4 < S) load_pair(st + 4, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 2 < S) compute_pair(a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int st = 0; st < S; st += 2) {
          load_pair(st, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          compute_pair(a0, b0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    BAND_ACC(5);
    BAND_ITEM();
    if (ch == nchunk - 1) {
      // ---- epilogue: lane owns channel cb = m*4 + (lane>>4) at grid pixel (ur, v); regs = (ph, pw)
      const int b = u / BANDS, band = u - b * BANDS;
      const int u0 = band * R;
      const int rows_g = min(R, Hg - u0);
      const int Pb = rows_g * Wg;
      const int rows_o = min(2 * rows_g, H - 2 * u0);  // output rows of this band
      float* out_tile = in_tile;
#pragma unroll
      for (int m0 = 0; m0 < MT; m0 += EM) {
        constexpr int BN_IT = (OPS / 4 * EM * 4 + 255) / 256;
        f4u bn_av[FUSE ? BN_IT : 1];
        if constexpr (FUSE) {  // saved activation of this channel group: in flight while the band goes through LDS
          const int nchf = min(Cb - m0 * 4, EM * 4);
          const int64_t offf = (((int64_t)b * Cb + m0 * 4) * H + 2 * u0) * W;
          bnred_fetch<EM * 4, BN_IT>(fuse.a + offf, (int64_t)H * W, nchf, rows_o * W, tid, bn_av);
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < EM; ++mm) {
          const int m = m0 + mm;
          float* ot = out_tile + (mm * 4 + (lane >> 4)) * OPS;
          const float bv = bias_r[m];
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int p = (wave * NT + t) * 16 + (lane & 15);
            const int ur = p / Wg, v = p - ur * Wg;
            const bool pin = p < Pb;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              const float v0 = pgv_act_apply(acc[m][t][ph * 2 + 0] + bv, actp);
              const float v1 = pgv_act_apply(acc[m][t][ph * 2 + 1] + bv, actp);
              const int orow = 2 * ur + ph;
              const bool rin = pin && orow < rows_o;
              float* o = ot + orow * W + 2 * v;
              if (W % 2 == 0) {
                if (p < R * Wg) *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                const float f0 = rin ? v0 : 0.f, f1 = rin ? v1 : 0.f;
                s += f0 + f1;
                q = fmaf(f0, f0, fmaf(f1, f1, q));
              } else {
                const bool c1 = 2 * v + 1 < W;
                if (p < R * Wg) {
                  o[0] = v0;
                  if (c1) o[1] = v1;
                }
                const float f0 = rin ? v0 : 0.f, f1 = (rin && c1) ? v1 : 0.f;
                s += f0 + f1;
                q = fmaf(f0, f0, fmaf(f1, f1, q));
              }
            }
          }
          if (stats) {
            if constexpr (kRegStats) {  // per-lane partial sums live in registers over all units of the workgroup
              st_s[m] += s;
              st_q[m] += q;
            } else {
              const float ss = group16_sum(s), qq = group16_sum(q);
              if ((lane & 15) == 0) {
                st_tile[(wave * MT * 4 + m * 4 + (lane >> 4)) * 2 + 0] += ss;  // slot owned by this lane
                st_tile[(wave * MT * 4 + m * 4 + (lane >> 4)) * 2 + 1] += qq;
              }
            }
          }
        }
        __syncthreads();
        const int nchn = min(Cb - m0 * 4, EM * 4);
        if (nchn > 0) {
          const int64_t off = (((int64_t)b * Cb + m0 * 4) * H + 2 * u0) * W;
          if constexpr (FUSE)  // BatchNorm + activation backward of the lower block on the way out (pgv_bwd_fuse)
            store_rows_bwd<EM * 4, BN_IT>(out_tile, OPS, out + off, fuse.a + off, (int64_t)H * W, nchn, rows_o * W, tid,
                                          fuse.coef + m0 * 4, Cb, actd, st_tile + m0 * 4, bn_av);
          else
            store_rows_contig(out_tile, OPS, out + off, (int64_t)H * W, nchn, rows_o * W, tid);
        }
      }
      BAND_ACC(6);
    }
    u = nu;
    ch = nch;
    if (u >= units) break;
  }
  if (stats) {  // one float64 atomic per channel per workgroup (see conv_down_band_kernel)
    if constexpr (kRegStats) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float ss = group16_sum(st_s[m]), qq = group16_sum(st_q[m]);
        if ((lane & 15) == 0) {
          st_tile[(wave * MT * 4 + m * 4 + (lane >> 4)) * 2 + 0] = ss;
          st_tile[(wave * MT * 4 + m * 4 + (lane >> 4)) * 2 + 1] = qq;
        }
      }
    }
    __syncthreads();
    if (tid < MT * 4 && tid < Cb) {
      double ss = 0.0, qq = 0.0;
#pragma unroll
      for (int wv2 = 0; wv2 < 4; ++wv2) {
        ss += (double)st_tile[(wv2 * MT * 4 + tid) * 2 + 0];
        qq += (double)st_tile[(wv2 * MT * 4 + tid) * 2 + 1];
      }
      atomicAdd(&stats[tid], ss);
      atomicAdd(&stats[Cb + tid], qq);
    }
  }
  if constexpr (FUSE) {
    __syncthreads();
    if (fuse.gbias && tid < MT * 4 && tid < Cb)
      atomicAdd(&fuse.gbias[(fuse.gbias_copies ? (blockIdx.x & (PGV_CLS_COPIES - 1)) * Cb : 0) + tid], st_tile[tid]);
  }
  BAND_FLUSH();
}

template <int MT, int NT, int CK, int NCH, bool WRES, int EM, int R, int W, int H, bool CANFUSE>
int launch_up_band(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                   const float* w, const float* bias, int act, float slope, float* out, double* stats,
                   const pgv_bwd_fuse* fuse, hipStream_t st) {
  using G = UpCfg<MT, NT, CK, NCH, WRES, EM, R, W, H>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  if (d->Cs > NCH * CK || d->Cb > MT * 4) return 0;
  if (fuse && !CANFUSE) return 0;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  auto kern = conv_up_band_kernel<MT, NT, CK, NCH, WRES, EM, R, W, H, false, false>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_up_band");
  if (rc) return rc;
  auto kernf = conv_up_band_kernel<MT, NT, CK, NCH, WRES, EM, R, W, H, CANFUSE, false>;
  if (CANFUSE) {
    static bool attr_done_f = false;
    if ((rc = raise_lds_limit(kernf, &attr_done_f, "conv_up_band"))) return rc;
  }
  auto kernb = conv_up_band_kernel<MT, NT, CK, NCH, WRES, EM, R, W, H, false, true>;
  static bool attr_done_b = false;
  if ((rc = raise_lds_limit(kernb, &attr_done_b, "conv_up_band"))) return rc;
  auto kernfb = conv_up_band_kernel<MT, NT, CK, NCH, WRES, EM, R, W, H, CANFUSE, true>;
  if (CANFUSE) {
    static bool attr_done_fb = false;
    if ((rc = raise_lds_limit(kernfb, &attr_done_fb, "conv_up_band"))) return rc;
  }
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_band: memset failed");
    return PGV_E_LAUNCH;
  }
  const int units = d->B * G::BANDS;
  int per_cu = (int)min((size_t)2, (size_t)kMaxLds / bytes);
  const int grid = min(units, 256 * per_cu);
  const pgv_bwd_fuse fz = {nullptr, nullptr, nullptr, 0, 0.f, nullptr};
  hipLaunchKernelGGL(bf16 ? (fuse ? kernfb : kernb) : (fuse ? kernf : kern), dim3(grid), dim3(256), bytes, st, d->B,
                     d->Cb, d->Cs, small_in, in_scale, in_shift, w, bias, act, slope, out, stats, fuse ? *fuse : fz);
  PGV_CHECK_LAUNCH("conv_up_band");
  return 1;
}

}  // namespace

int pgv_conv_down_band(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                       const float* w, const float* bias, int act, float slope, float* small_out, double* stats,
                       const pgv_bwd_fuse* fuse, hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4 || d->Cb > 64) return 0;
  if (d->Hb == 129 && d->Wb == 174 && d->Cs <= 16)
    return launch_down_band<1, 6, 8, 1, true, 4, 174, 129, true>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, st);
  if (d->Hb == 65 && d->Wb == 88 && d->Cs <= 32)
    return launch_down_band<2, 3, 8, 2, true, 4, 88, 65, true>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, st);
  if (d->Hb == 33 && d->Wb == 45 && d->Cs <= 64)
    return launch_down_band<4, 4, 8, 4, false, 9, 45, 33, false>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, st);
  return 0;
}

int pgv_conv_wgrad_band(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                        const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                        hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != d->kw) return 0;
#define PGV_WGB(...) \
  return launch_wgrad_band<__VA_ARGS__>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, st)
  if (d->kh == 5 && d->Hb == 257 && d->Wb == 347) PGV_WGB(5, 1, 8, 1, 1, 1, 1, 347, 257);
  if (d->kh != 4) return 0;
  if (d->Hb == 129 && d->Wb == 174) PGV_WGB(4, 1, 16, 8, 1, 1, 1, 174, 129);
  if (d->Hb == 65 && d->Wb == 88) PGV_WGB(4, 2, 32, 16, 1, 2, 1, 88, 65);
  if (d->Hb == 33 && d->Wb == 45) PGV_WGB(4, 4, 64, 16, 2, 4, 3, 45, 33);
#undef PGV_WGB
  return 0;
}

// The same kernels with their result left as per-workgroup partial gradients in ``partial`` (*nparts slots in the layout
// of gw) for the reduce launch of conv_v2_wgrad.hip.  bf16 operand mode, the three k4 layers at the reference sizes and
// channel counts; 0 = not covered (nothing launched).
int pgv_conv_wgrad_band_partial(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                                const float* small_in, const float* small_scale, const float* small_shift, float* partial,
                                int64_t partial_bytes, int* nparts, hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4 || !(d->flags & PGV_COMPUTE_BF16) || !partial) return 0;
#define PGV_WGP(...)                                                                                                  \
  return launch_wgrad_band_t<__VA_ARGS__, true>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, nullptr, \
                                                st, partial, partial_bytes, nparts)
  if (d->Hb == 129 && d->Wb == 174 && d->Cb == 8 && d->Cs == 16) PGV_WGP(4, 1, 16, 8, 1, 1, 1, 174, 129);
  if (d->Hb == 65 && d->Wb == 88 && d->Cb == 16 && d->Cs == 32) PGV_WGP(4, 2, 32, 16, 1, 2, 1, 88, 65);
  if (d->Hb == 33 && d->Wb == 45 && d->Cb == 32 && d->Cs == 64) PGV_WGP(4, 4, 64, 16, 2, 4, 3, 45, 33);
#undef PGV_WGP
  return 0;
}

int pgv_conv_up_band(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* big_out, double* stats,
                     const pgv_bwd_fuse* fuse, hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4) return 0;
  if (d->Hb == 129 && d->Wb == 174)
    return launch_up_band<2, 7, 16, 1, true, 2, 5, 174, 129, true>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, st);  // 65 grid rows = 13 x 5
  if (d->Hb == 65 && d->Wb == 88)
    return launch_up_band<4, 3, 16, 2, true, 4, 4, 88, 65, true>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, st);
  if (d->Hb == 33 && d->Wb == 45)
    return launch_up_band<8, 3, 16, 4, false, 4, 8, 45, 33, false>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, st);  // epilogue in 2 passes of 16 channels
  return 0;
}
