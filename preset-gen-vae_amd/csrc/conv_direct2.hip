// Second-generation direct (vector-ALU) kernels for the 1 <-> 8 channel 5x5 stride-2 layers at the reference size:
// enc1 = Conv2d(1,8,5,2,2) (model/encoder.py:241), the output layer ConvTranspose2d(8,1,5,2,2) (model/decoder.py:218) and
// the input gradient of the latter - the [B,1,257,347] <-> [B,8,129,174] pair.
//
// These layers move 1.07 MB per sample for 4.5 MMAC: bound by the CU's memory path (~10 B/clk/CU, i.e. ~46 us for the
// 275 MB of a batch of 256).  The first-generation kernels (conv_direct.hip) gave one output pixel (or one 2x2 output
// block) to a lane: 25 (9 per channel) ds_read_b32 per lane and DWORD stores - 4 scattered dwords per lane in the
// transposed case - which made them store-issue- and LDS-issue-bound (113 / 75 us).  Here a lane owns FOUR consecutive
// output pixels (grid positions): the input patch comes in as ds_read_b128 / b64 from a zero-padded LDS band (no column
// masks), every result leaves as a 16-byte store, and the band halo shrinks from 40 % to 9-18 % (11 rows per unit).
// Workgroups are persistent (two per CU: one stages while the other multiplies).
#include <stdlib.h>
#include "conv_tile.h"
#include "band_prefetch.h"

namespace {

constexpr int K5 = 5, KK5 = 25;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- UP: out[b,0,2u+ph,2v+pw] = act(bias + sum_{cs,th,tw} x'[b,cs,u+1-th,v+1-tw] * w[cs,0,ph+2th,pw+2tw]) ----------
// th, tw in {0,1,2}; kh = ph + 2 th <= 4.  A lane owns grid positions v0 .. v0+3 of one grid row: 2 output rows x 8
// output columns.
template <int CS, int H, int W, int R>
struct UpC1Cfg {
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, Hg = (H + 1) / 2, Wg = (W + 1) / 2;
  static constexpr int WsP = (Ws + 2 + 3) / 4 * 4, ROWS = R + 2, PLANE = ROWS * WsP;
  static constexpr int QW = (Wg + 3) / 4, BANDS = (Hg + R - 1) / R;
  static constexpr int FRONT = 4;
  static constexpr int WSTR = 28;  // LDS weight row per channel: 25 taps padded to 7 x 16 bytes
  static constexpr size_t LDS_FLOATS = FRONT + (size_t)CS * PLANE + 16 + 2 * CS + 8 + CS * WSTR;
};

template <int CS, int H, int W, int R>
__global__ __launch_bounds__(256, (R * ((W + 1) / 2 + 3) / 4 <= 256) ? 3 : 2) void up_c1_v2_kernel(int B, const float* __restrict__ small_in,
                                                        const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int act, float slope,
                                                        float* __restrict__ out, int bf16, pgv_bn_src bn) {
  using G = UpC1Cfg<CS, H, W, R>;
  constexpr int Hs = G::Hs, Ws = G::Ws, Hg = G::Hg, WsP = G::WsP, PLANE = G::PLANE, QW = G::QW, BANDS = G::BANDS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds + G::FRONT;
  float* aff = tile + CS * PLANE + 16;
  // the 200 weights are uniform: read from LDS (broadcast) one channel at a time into registers - as 200 scalar
  // registers they do not fit (the compiler spills them to VGPR lanes: one v_readlane per FMA)
  float* wl = aff + 2 * CS + ((4 - ((2 * CS) & 3)) & 3) + 4;  // 16-byte aligned
  wl = lds + ((wl - lds) + 3) / 4 * 4;
  const int tid = threadIdx.x;
  if (tid < G::FRONT) lds[tid] = 0.f;
  if (tid < 16) tile[CS * PLANE + tid] = 0.f;
  if (in_scale && tid < CS) {
    float sc, sh;
    // (pgv_conv_up_bn: the producer's BatchNorm is finalized here, from its statistics, instead of by a launch of its own)
    if (bn.stats)
      pgv_bn_finalize_dev(bn, CS, tid, blockIdx.x == 0, sc, sh);
    else
      sc = in_scale[tid], sh = in_shift[tid];
    aff[tid] = sc;
    aff[CS + tid] = sh;
  }
  for (int i = tid; i < CS * G::WSTR; i += 256) {
    const int cs = i / G::WSTR, k = i - cs * G::WSTR;
    wl[i] = k < KK5 ? pgv_opnd(w[cs * KK5 + k], bf16 != 0) : 0.f;   // (PGV_COMPUTE_BF16: operands rounded, fp32 FMAs)
  }
  const pgv_act_params actp = pgv_act_setup(act, slope);
  const float bv = bias ? bias[0] : 0.f;
  const int units = B * BANDS;
  // register prefetch (band_prefetch.h): ALL 16-byte loads of a unit are in flight at once, and the loads of unit n+1
  // are issued before unit n is multiplied (the first-generation kernels staged 4 loads per lane at a time: five
  // dependent HBM round trips per unit - that, not arithmetic or bandwidth, made them take 110 us)
  typename PickPrefetch<CS, G::ROWS, Ws, WsP, Hs>::type pf;
  pf.init(tid);
  auto issue_unit = [&](int un) {
    const int b = un / BANDS, band = un - b * BANDS;
    pf.issue(small_in + (int64_t)b * CS * Hs * Ws, band * R - 1, CS);
  };
  const int bid = pgv_xcd_block();
  if (bid < units) issue_unit(bid);
  for (int un = bid; un < units; un += gridDim.x) {
    const int b = un / BANDS, band = un - b * BANDS;
    const int u0 = band * R, Rb = min(R, Hg - u0);
    __syncthreads();  // the previous unit's reads are complete (first pass: the tables are visible)
    pf.commit(tile, in_scale ? aff : nullptr, CS, 0, CS, tid, bf16 != 0);
    if (un + (int)gridDim.x < units) issue_unit(un + gridDim.x);
    __syncthreads();
    float* ob = out + (int64_t)b * H * W;
    // NPASS quads per lane, processed together: the weights of a channel are read from LDS once per unit and used for
    // all of them (inside a per-quad loop the compiler hoists all 8 x 28 weight registers out and spills)
    constexpr int NPASS = (R * QW + 255) / 256;
    int rq[NPASS], vq0[NPASS];
    bool okq[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int q = tid + 256 * ps;
      okq[ps] = q < Rb * QW;
      const int qq = okq[ps] ? q : 0;
      rq[ps] = qq / QW;
      vq0[ps] = 4 * (qq - rq[ps] * QW);
    }
    // accumulators as (pw = 0, pw = 1) pairs: one v_pk_fma_f32 per input value and kernel-column pair
    f32x2 acc[NPASS][2][4];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[ps][0][i] = acc[ps][1][i] = f32x2{bv, bv};
#pragma unroll 1
    for (int cs = 0; cs < CS; ++cs) {
      float wc[G::WSTR];
#pragma unroll
      for (int i = 0; i < G::WSTR / 4; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(wl + cs * G::WSTR + 4 * i);
        wc[4 * i] = t.x, wc[4 * i + 1] = t.y, wc[4 * i + 2] = t.z, wc[4 * i + 3] = t.w;
      }
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
        for (int th = 0; th < 3; ++th) {
          const float* row = tile + cs * PLANE + (rq[ps] + 2 - th) * WsP + vq0[ps];  // local row of input row u+1-th
          const float2 xm = *reinterpret_cast<const float2*>(row - 2);
          const f32x4 xc = *reinterpret_cast<const f32x4*>(row);
          const float2 xp = *reinterpret_cast<const float2*>(row + 4);
          const float x[6] = {xm.y, xc.x, xc.y, xc.z, xc.w, xp.x};  // columns v0-1 .. v0+4
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
              const float xin = x[j + 2 - tw];  // column v0+j+1-tw
              const f32x2 xx = {xin, xin};
              // kernel columns kw = 2 tw (pw = 0) and 2 tw + 1 (pw = 1; does not exist for tw = 2)
              const f32x2 w0 = {wc[(2 * th) * K5 + 2 * tw], tw < 2 ? wc[(2 * th) * K5 + 2 * tw + 1] : 0.f};
              acc[ps][0][j] = __builtin_elementwise_fma(xx, w0, acc[ps][0][j]);
              if (th < 2) {
                const f32x2 w1 = {wc[(2 * th + 1) * K5 + 2 * tw], tw < 2 ? wc[(2 * th + 1) * K5 + 2 * tw + 1] : 0.f};
                acc[ps][1][j] = __builtin_elementwise_fma(xx, w1, acc[ps][1][j]);
              }
            }
        }
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if (!okq[ps]) continue;
      const int oh = 2 * (u0 + rq[ps]), ow = 2 * vq0[ps];
      const int n = W - ow;  // valid output columns from ow on
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        if (oh + ph < H) {
          float* o = ob + (int64_t)(oh + ph) * W + ow;
          float y[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) y[i] = pgv_act_apply_nan(acc[ps][ph][i >> 1][i & 1], actp);
          if (n >= 8) {
            f4u a, c;
            a.x = y[0], a.y = y[1], a.z = y[2], a.w = y[3];
            c.x = y[4], c.y = y[5], c.z = y[6], c.w = y[7];
            *reinterpret_cast<f4u*>(o) = a;
            *reinterpret_cast<f4u*>(o + 4) = c;
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
              if (i < n) o[i] = y[i];
          }
        }
      }
    }
  }
}

// ---- DOWN: out[b,cs,oh,ow] = act(bias[cs] + sum_{kh,kw} x'[b,0,2oh-2+kh,2ow-2+kw] * w[cs,0,kh,kw]) ------------------
// A lane owns output pixels ow0 .. ow0+3 of one output row, all CS channels.  FUSE: BatchNorm-backward projections of
// the written tensor against the saved activation `a` (pgv_bwd_fuse), one float64 atomic per channel per workgroup.
template <int CS, int H, int W, int R>
struct DownC1Cfg {
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1;
  static constexpr int WP = (W + 2 + 3) / 4 * 4, ROWS = 2 * (R - 1) + K5;
  static constexpr int QW = (Ws + 3) / 4, BANDS = (Hs + R - 1) / R;
  static constexpr int FRONT = 4;
  static constexpr size_t LDS_FLOATS = FRONT + (size_t)ROWS * WP + 16 + 4 + 4 * 4 * CS + 8 + K5 * CS * 8;
};

template <int CS, int H, int W, int R, bool FUSE>
__global__ __launch_bounds__(256, (R * ((W / 2 + 1) + 3) / 4 <= 256) ? 4 : 2) void down_c1_v2_kernel(int B, const float* __restrict__ big,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          int act, float slope, float* __restrict__ out,
                                                          pgv_bwd_fuse fuse, int bf16) {
  using G = DownC1Cfg<CS, H, W, R>;
  constexpr int Hs = G::Hs, Ws = G::Ws, WP = G::WP, QW = G::QW, BANDS = G::BANDS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds + G::FRONT;
  float* aff = tile + G::ROWS * WP + 16;  // [2] (+2 pad)
  float* red = aff + 4;                   // [4 waves][4*CS]
  float* wl = lds + ((red + 4 * 4 * CS - lds) + 3) / 4 * 4;  // [kh][kw][cs], 16-byte aligned
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < G::FRONT) lds[tid] = 0.f;
  if (tid < 16) tile[G::ROWS * WP + tid] = 0.f;
  for (int i = tid; i < KK5 * CS; i += 256) {
    const int k = i / CS, cs = i - k * CS;
    wl[i] = pgv_opnd(w[cs * KK5 + k], bf16 != 0);   // (PGV_COMPUTE_BF16: operands rounded, fp32 FMAs)
  }
  if (in_scale && tid == 0) {
    aff[0] = in_scale[0];
    aff[1] = in_shift[0];
  }
  const pgv_act_params actp = pgv_act_setup(act, slope);
  // FUSE (pgv_bwd_fuse): the lower block's BatchNorm + activation backward, g_y = act'(a) * (ka*g + kb*a + kc)
  const pgv_actd_params actd = pgv_actd_setup(FUSE ? fuse.act : 0, FUSE ? fuse.slope : 0.f);
  // s1: sums of g_y over this lane's even / odd output columns (its row parity is a constant of the kernel, see below);
  // by (row parity, column parity) class they are pgv_bwd_fuse.cls, their total is the bias gradient
  float bias_r[CS], ka_r[CS], kb_r[CS], kc_r[CS], s1[CS][2];
#pragma unroll
  for (int cs = 0; cs < CS; ++cs) {
    bias_r[cs] = bias ? bias[cs] : 0.f;
    ka_r[cs] = FUSE ? fuse.coef[cs] : 0.f;
    kb_r[cs] = FUSE ? fuse.coef[CS + cs] : 0.f;
    kc_r[cs] = FUSE ? fuse.coef[2 * CS + cs] : 0.f;
    s1[cs][0] = s1[cs][1] = 0.f;
  }
  const int units = B * BANDS;
  typename PickPrefetch<1, G::ROWS, W, WP, H>::type pf;  // (see up_c1_v2_kernel)
  pf.init(tid);
  auto issue_unit = [&](int un) {
    const int b = un / BANDS, band = un - b * BANDS;
    pf.issue(big + (int64_t)b * H * W, 2 * band * R - 2, 1);
  };
  const int bid = pgv_xcd_block();
  if (bid < units) issue_unit(bid);
  for (int un = bid; un < units; un += gridDim.x) {
    const int b = un / BANDS, band = un - b * BANDS;
    const int oh0 = band * R, Rb = min(R, Hs - oh0);
    __syncthreads();
    pf.commit(tile, in_scale ? aff : nullptr, 1, 0, 1, tid, bf16 != 0);
    if (un + (int)gridDim.x < units) issue_unit(un + gridDim.x);
    __syncthreads();
    constexpr int NPASS = (R * QW + 255) / 256;
    int rq[NPASS], vqq[NPASS];
    bool okq[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int q = tid + 256 * ps;
      okq[ps] = q < Rb * QW;
      const int qq = okq[ps] ? q : 0;
      rq[ps] = qq / QW;
      vqq[ps] = qq - rq[ps] * QW;
    }
    // FUSE: the saved activation of the quad's channels is fetched ahead of the multiply phase - issued in the
    // epilogue (as it first was) every unit exposed one memory latency to the whole workgroup: 150 us against 59 us for
    // the plain kernel, slower than the separate reduce pass the fusion replaces.  The fused kernel walks the CS channels
    // in halves (CH at a time) over the same LDS tile: accumulators and prefetched activations of ONE half are live at a
    // time (32 registers fewer: a fourth workgroup per CU); the second half's activations are requested when the first
    // half's have been consumed, and are covered by the second half's multiply phase and the other workgroups.
    const int64_t cstride = (int64_t)Hs * Ws;
    constexpr int CH = FUSE ? CS / 2 : CS;   // channels per pass over the tile
    f4u avv[FUSE ? NPASS : 1][FUSE ? CH : 1];
    auto fetch_a = [&](int c0) {
      if constexpr (FUSE) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int ow0 = 4 * vqq[ps];
          const int64_t off = (int64_t)b * CS * cstride + (int64_t)(oh0 + rq[ps]) * Ws + ow0;
          if (okq[ps] && Ws - ow0 >= 4) {
#pragma unroll
            for (int cs = 0; cs < CH; ++cs) avv[ps][cs] = *reinterpret_cast<const f4u*>(fuse.a + off + (c0 + cs) * cstride);
          }
        }
      }
    };
    fetch_a(0);
#pragma unroll
    for (int c0 = 0; c0 < CS; c0 += CH) {
    // accumulators as channel pairs: one v_pk_fma_f32 per input value, tap and channel pair
    f32x2 acc[NPASS][4][CH / 2];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c2 = 0; c2 < CH / 2; ++c2) acc[ps][j][c2] = f32x2{bias_r[c0 + 2 * c2], bias_r[c0 + 2 * c2 + 1]};
#pragma unroll 1
    for (int kh = 0; kh < K5; ++kh) {
      float x[NPASS][11];
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const float* row = tile + (2 * rq[ps] + kh) * WP + 8 * vqq[ps];  // columns 8vq-2 .. 8vq+8 are needed
        const f32x4 xa = *reinterpret_cast<const f32x4*>(row - 4);
        const f32x4 xb = *reinterpret_cast<const f32x4*>(row);
        const f32x4 xc = *reinterpret_cast<const f32x4*>(row + 4);
        x[ps][0] = xa.z, x[ps][1] = xa.w, x[ps][2] = xb.x, x[ps][3] = xb.y, x[ps][4] = xb.z, x[ps][5] = xb.w;
        x[ps][6] = xc.x, x[ps][7] = xc.y, x[ps][8] = xc.z, x[ps][9] = xc.w, x[ps][10] = row[8];
      }
#pragma unroll
      for (int kw = 0; kw < K5; ++kw) {
        // wl[kh][kw][cs]: the weights of one tap, uniform address (broadcast reads), channel pairs adjacent
        f32x2 wp[CH / 2];
#pragma unroll
        for (int c4 = 0; c4 < CH / 4; ++c4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(wl + (kh * K5 + kw) * CS + c0 + 4 * c4);
          wp[2 * c4] = f32x2{t.x, t.y};
          wp[2 * c4 + 1] = f32x2{t.z, t.w};
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xin = x[ps][2 * j + kw];
            const f32x2 xx = {xin, xin};
#pragma unroll
            for (int c2 = 0; c2 < CH / 2; ++c2) acc[ps][j][c2] = __builtin_elementwise_fma(xx, wp[c2], acc[ps][j][c2]);
          }
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if (!okq[ps]) continue;
      const int ow0 = 4 * vqq[ps];
      const int n = Ws - ow0;  // valid pixels from ow0 on
      const int64_t off = (int64_t)b * CS * cstride + (int64_t)(oh0 + rq[ps]) * Ws + ow0;
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const int cs = c0 + ch;
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = pgv_act_apply(acc[ps][j][ch >> 1][ch & 1], actp);
        float* o = out + off + cs * cstride;
        if (n >= 4) {
          if constexpr (FUSE) {
            const f4u av = avv[ps][ch];
            y[0] = pgv_bwd_apply(y[0], av.x, ka_r[cs], kb_r[cs], kc_r[cs], actd);
            y[1] = pgv_bwd_apply(y[1], av.y, ka_r[cs], kb_r[cs], kc_r[cs], actd);
            y[2] = pgv_bwd_apply(y[2], av.z, ka_r[cs], kb_r[cs], kc_r[cs], actd);
            y[3] = pgv_bwd_apply(y[3], av.w, ka_r[cs], kb_r[cs], kc_r[cs], actd);
            s1[cs][0] += y[0] + y[2], s1[cs][1] += y[1] + y[3];   // ow0 is a multiple of 4: even / odd columns
          }
          f4u t;
          t.x = y[0], t.y = y[1], t.z = y[2], t.w = y[3];
          *reinterpret_cast<f4u*>(o) = t;
        } else {
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (j < n) {
              float yy = y[j];
              if constexpr (FUSE) {
                yy = pgv_bwd_apply(yy, fuse.a[off + cs * cstride + j], ka_r[cs], kb_r[cs], kc_r[cs], actd);
                s1[cs][j & 1] += yy;
              }
              o[j] = yy;
            }
        }
      }
    }
    if (c0 + CH < CS) fetch_a(c0 + CH);   // (the first half's activations have been consumed)
    }
  }
  if constexpr (FUSE) {  // class sums / bias gradient of the lower block: one float atomic per value per workgroup
    __syncthreads();
    // Row parity of this lane's output row: a unit is R rows of a sample, units are walked bid, bid + grid, ...; with an
    // even number of bands per sample and an even grid (or one unit per workgroup) the parity of the band index is the
    // parity of bid, and the lane's row inside the band (rq = tid / QW) never changes - one quad per lane (NPASS = 1).
    static_assert(R * QW <= 256 && BANDS % 2 == 0, "constant row parity per lane");
    const bool rodd = ((R & 1 ? bid : 0) + tid / QW) & 1;
    float v32[4 * CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs)
#pragma unroll
      for (int k = 0; k < 4; ++k) v32[4 * cs + k] = ((k >> 1) != 0) == rodd ? s1[cs][k & 1] : 0.f;
    // (DPP row reductions + one barrier: 4*CS wave reductions through ds_bpermute took ~10 us at the end of every workgroup)
    float t = pgv_block_sums<4 * CS>(v32, red);
    if (tid < 4 * CS) {
      if (fuse.cls) atomicAdd(&fuse.cls[(blockIdx.x & (PGV_CLS_COPIES - 1)) * 4 * CS + tid], t);   // (partial copy by XCD)
      // bias gradient = the four classes of a channel added up (lanes 4cs .. 4cs+3)
      t += dpp_mov<0xB1>(t);
      t += dpp_mov<0x4E>(t);
      if (fuse.gbias && (tid & 3) == 0)
        atomicAdd(&fuse.gbias[(fuse.gbias_copies ? (blockIdx.x & (PGV_CLS_COPIES - 1)) * CS : 0) + (tid >> 2)], t);
    }
  }
}

template <typename K>
int raise_lds(K kern, const char* who) {
  static const void* done[16];
  static int n_done = 0;
  for (int i = 0; i < n_done; ++i)
    if (done[i] == (const void*)kern) return PGV_OK;
  const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
  if (e != hipSuccess) {
    pgv_set_error("%s: cannot raise the dynamic LDS limit: %s", who, hipGetErrorString(e));
    return PGV_E_LAUNCH;
  }
  if (n_done < 16) done[n_done++] = (const void*)kern;
  return PGV_OK;
}

}  // namespace

// The reference size only (8 channels, 257x347 <-> 129x174), fp32 or bf16 operand mode (operands rounded where they are
// committed to LDS, fp32 FMAs: products of bf16 values are exact in fp32); everything else stays with conv_direct.hip.
template <int R>
static int launch_up_c1(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                        const float* w, const float* bias, int act, float slope, float* out, const pgv_bn_src* bn,
                        hipStream_t st) {
  using G = UpC1Cfg<8, 257, 347, R>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds / 2, "two workgroups per CU");
  auto kern = up_c1_v2_kernel<8, 257, 347, R>;
  if (int rc = raise_lds(kern, "conv_up_direct2")) return rc;
  const int units = d->B * G::BANDS;
  const int per_cu = (int)min((size_t)4, (size_t)kMaxLds / bytes);
  hipLaunchKernelGGL(kern, dim3(min(units, 256 * per_cu)), dim3(256), bytes, st, d->B, small_in, in_scale, in_shift, w,
                     bias, act, slope, out, (d->flags & PGV_COMPUTE_BF16) ? 1 : 0, bn ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("conv_up_direct2");
  return 1;
}

int pgv_conv_up_direct2(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                        const float* w, const float* bias, int act, float slope, float* out, double* stats,
                        hipStream_t st, const pgv_bn_src* bn) {
  if (d->kh != 5 || d->kw != 5 || d->stride != 2 || d->pad != 2 || d->Cb != 1 || d->Cs != 8 || stats) return 0;
  if (d->Hb != 257 || d->Wb != 347 || d->B <= 0) return 0;
  if (bn && !in_scale) return 0;
#ifndef PGV_NO_C1_RING
  // fp32: the ring kernel (conv_c1_ring.hip: rows by LDS-DMA, affine folded into the weights); bf16 operand mode rounds the
  // operand AFTER the affine and stays here
  if (int rc = pgv_conv_up_ring(d, small_in, in_scale, in_shift, w, bias, act, slope, out, stats, st, bn)) return rc;
#endif
  // 11 grid rows per unit: since the units of neighbouring bands share an XCD's L2 (pgv_xcd_block) the larger unit no
  // longer pays for its halo and its longer multiply phase hides more of the next unit's loads (88 -> 83 us)
  return launch_up_c1<11>(d, small_in, in_scale, in_shift, w, bias, act, slope, out, bn, st);
}

template <int R>
static int launch_down_c1(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                          const float* w, const float* bias, int act, float slope, float* out, const pgv_bwd_fuse* fuse,
                          hipStream_t st) {
  using G = DownC1Cfg<8, 257, 347, R>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds / 2, "two workgroups per CU");
  const int units = d->B * G::BANDS;
  const int per_cu = (int)min((size_t)4, (size_t)kMaxLds / bytes);
  const int grid = min(units, 256 * per_cu);
  const pgv_bwd_fuse fz = {nullptr, nullptr, nullptr, 0, 0.f, nullptr};
  if (fuse) {
    auto kern = down_c1_v2_kernel<8, 257, 347, R, true>;
    if (int rc = raise_lds(kern, "conv_down_direct2")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, st, d->B, big, in_scale, in_shift, w, bias, act, slope, out,
                       *fuse, (d->flags & PGV_COMPUTE_BF16) ? 1 : 0);
    PGV_CHECK_LAUNCH("conv_down_direct2");
    return fuse->cls ? 3 : 1;   // (3: handled, class sums included)
  } else {
    auto kern = down_c1_v2_kernel<8, 257, 347, R, false>;
    if (int rc = raise_lds(kern, "conv_down_direct2")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, st, d->B, big, in_scale, in_shift, w, bias, act, slope, out,
                       fz, (d->flags & PGV_COMPUTE_BF16) ? 1 : 0);
  }
  PGV_CHECK_LAUNCH("conv_down_direct2");
  return 1;
}

int pgv_conv_down_direct2(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                          const float* w, const float* bias, int act, float slope, float* out, double* stats,
                          const pgv_bwd_fuse* fuse, hipStream_t st) {
  if (d->kh != 5 || d->kw != 5 || d->stride != 2 || d->pad != 2 || d->Cb != 1 || d->Cs != 8 || stats) return 0;
  if (d->Hb != 257 || d->Wb != 347 || d->B <= 0) return 0;
  return launch_down_c1<5>(d, big, in_scale, in_shift, w, bias, act, slope, out, fuse, st);
}
