// Implicit-GEMM convolutions on the gfx950 f32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fma chain, same
// peak as the fp32 VALU but one instruction retires 1024 MACs from two VGPR operands).
//
// Stride-2 / pad-2 layers of the reference tables (model/encoder.py:241-255, model/decoder.py:205-218; k = 4 or 5).
//
//  * DOWN  (Conv2d forward, ConvTranspose2d input-gradient):  D[cs][pixel] = sum_k W[cs][k] * X[k][pixel],
//      k = (cb, kh, kw).  One MFMA consumes the 4 kw taps of one (cb, kh): the B operand of lane (pixel j, kw) is read
//      straight out of the RAW input tile in LDS at  2*row*Wt + 2*col + kw  (no im2col is ever materialised), the A
//      operand is the weight tile transposed to [k][cs].
//  * UP    (ConvTranspose2d forward, Conv2d input-gradient) by sub-pixel phases: output pixel (2u+ph, 2v+pw) only
//      sees taps kh = ph+2*th, kw = pw+2*tw, reading input (u+1-th, v+1-tw) — the SAME input gather for all four
//      phases.  So  D[(cb,ph,pw)][(u,v)] = sum_k W'[(cb,ph,pw)][k] * X[k][(u,v)],  k = (cs, th, tw): M = 4*Cb rows, one
//      MFMA per (cs, th-group); each lane ends up with the 2x2 output block of its (u,v) for one channel and stores
//      it as two float2 rows.
//  * WGRAD gw[cs][cb][tap] = sum_pixels small[cs][pixel] * big[cb][pixel, tap]: M = cs, N = (cb, 16 taps), K = 4
//      consecutive output pixels per MFMA; persistent workgroups keep the accumulators in registers over many
//      (sample, band) units and flush once with float atomics.
//
// A workgroup (4 waves) owns a full-width band of output rows of one sample, so every global row it touches is a
// contiguous NCHW segment; the band's input rows (+halo, zero padding, and the producer's BatchNorm affine folded in)
// are staged into LDS with batched (8 loads in flight per lane) coalesced reads, weights are staged per channel
// chunk.  Epilogue: bias + LeakyReLU/Hardtanh, store, and per-channel sum / sum-of-squares partials (wave shuffle ->
// LDS -> one float64 atomic per channel per workgroup) for the BatchNorm that follows.
#include "conv_tile.h"

// Phase timing for kernel tuning (scratch/phase_timing.py builds a separate debug library with -DPGV_PHASE_TIMING;
// the product library never defines it).
#ifdef PGV_PHASE_TIMING
__device__ unsigned long long* pgv_tlog = nullptr;  // [workgroup][wave][8] wall_clock64() stamps (100 MHz)
extern "C" int pgv_dbg_set_tlog(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(pgv_tlog), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define PGV_TICK(i)                                                                                       \
  do {                                                                                                    \
    if ((threadIdx.x & 63) == 0 && pgv_tlog)                                                              \
      pgv_tlog[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8 + (i)] =     \
          wall_clock64();                                                                                 \
  } while (0)
#define PGV_TICK_HWID()                                                                                   \
  do {                                                                                                    \
    if ((threadIdx.x & 63) == 0 && pgv_tlog)                                                              \
      pgv_tlog[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8 + 7] =       \
          ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |                       \
          (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);                                            \
  } while (0)
#else
#define PGV_TICK(i)
#define PGV_TICK_HWID()
#endif

#ifndef PGV_STAGE_U
#define PGV_STAGE_U 4  // 16-byte loads in flight per lane while staging a band
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------
// DOWN
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int MT, int NT, int CK>
__global__ __launch_bounds__(256, 2) void conv_down_mfma_kernel(pgv_conv_desc d, const float* __restrict__ big,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int act, float slope,
                                                             float* __restrict__ out, double* __restrict__ stats,
                                                             int R, int plane) {
  constexpr int KWS = (KS + 3) / 4;  // MFMA k-groups along kw
  constexpr int KWP = KWS * 4;       // kw padded to a multiple of 4 (zero weights beyond KS)
  constexpr int CSP = MT * 16 + 1;   // padded [k][cs] weight row: odd stride => conflict-free transposing writes
  constexpr int KC = CK * KS * KWP;  // k rows per channel chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rows_in = 2 * (R - 1) + KS;
  const int Wb = d.Wb;
  float* w_tile = lds + 8 + CK * plane;
  float* st_tile = w_tile + KC * CSP;      // [4 waves][MT*16][2]
  float* aff = st_tile + 4 * MT * 16 * 2;  // [2][Cb]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int oh0 = blockIdx.x * R;
  const int rows_out = min(R, d.Hs - oh0);
  const int Pb = rows_out * d.Ws;
  const int ih0 = oh0 * 2 - d.pad;
  const float* src = big + (int64_t)b * d.Cb * d.Hb * Wb;
  // input tile: row stride = Wb (rows are copied as contiguous NCHW segments); 4 floats of front slack for the
  // col = -2 reads, plus a shift that makes the first copied row 16-byte aligned in LDS
  const int lead = (max(ih0, 0) - ih0) * Wb;
  float* in_tile = lds + 4 + ((4 - (lead & 3)) & 3);

  int offB[NT];
  bool okB[NT][KWS];
  const float inv_ws = 1.0f / (float)d.Ws;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wave * NT + t) * 16 + (lane & 15);
    const int pv = p < Pb ? p : 0;
    const int r = fast_div(pv, inv_ws), c = pv - r * d.Ws;
    const int col = 2 * c - d.pad + (lane >> 4);
    offB[t] = 2 * r * Wb + col;
#pragma unroll
    for (int kws = 0; kws < KWS; ++kws) okB[t][kws] = (unsigned)(col + 4 * kws) < (unsigned)Wb;
  }
  const int offA = (lane >> 4) * CSP + (lane & 15);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef PGV_STAGGER
  {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (lin >= 256 && lin < 512)
      for (int i = 0; i < PGV_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  PGV_TICK(0);
  PGV_TICK_HWID();
  stage_affine(aff, in_scale, in_shift, d.Cb, tid);
  if (in_scale) __syncthreads();
  for (int cb0 = 0; cb0 < d.Cb; cb0 += CK) {
    if (cb0) __syncthreads();
#if !defined(PGV_EXP) || (PGV_EXP != 3)
    stage_rows_contig<PGV_STAGE_U>(in_tile, plane, src, d.Cb, d.Hb, Wb, cb0, CK, rows_in, ih0, in_scale ? aff : nullptr,
                         aff + d.Cb, tid);
#endif
    PGV_TICK(1);
    for (int i0 = tid; i0 < KC * MT * 16; i0 += 256 * 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * 256;
        const int cs = idx / KC, k = idx - cs * KC;
        const int c = k / (KS * KWP), rem = k - c * (KS * KWP);
        const int kh = rem / KWP, kw = rem - kh * KWP;
        const bool ok = idx < KC * MT * 16 && cs < d.Cs && cb0 + c < d.Cb && kw < KS;
        v[u] = ok ? w[(((int64_t)cs * d.Cb + cb0 + c) * KS + kh) * KS + kw] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * 256;
        if (idx < KC * MT * 16) {
          const int cs = idx / KC, k = idx - cs * KC;
          w_tile[k * CSP + cs] = v[u];
        }
      }
    }
    __syncthreads();
    PGV_TICK(2);
    // MFMA loop, software-pipelined by hand: step s = (c, kh, kws) (kws fastest); the LDS reads of step s+1 are
    // issued into the other register set before the MFMAs of step s, so ds_read latency hides under matrix work.
    {
      constexpr int J = KS * KWS;  // steps per channel (even: 4 or 10)
      constexpr int S = CK * J;
      float a0[MT], a1[MT], b0[NT], b1[NT];
      auto load_step = [&](int st, float (&av)[MT], float (&bv)[NT]) {
        const int c = st / J, rem = st - c * J;
        const int kh = rem / KWS, kws = rem - kh * KWS;
        const float* ap = w_tile + (st * 4) * CSP + offA;
        const float* bp = in_tile + c * plane + kh * Wb + kws * 4;
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = ap[m * 16];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = bp[offB[t]];
      };
      auto compute_step = [&](int kws, const float (&av)[MT], float (&bv)[NT]) {
#if !defined(PGV_EXP) || (PGV_EXP != 4)
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = okB[t][kws] ? bv[t] : 0.f;
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], acc[m][t], 0, 0, 0);
      };
#if !defined(PGV_EXP) || (PGV_EXP != 1)
      // sched_barrier(0): nothing crosses.  Without them the compiler sinks the ds_reads of the next step down to
      // their first use (shorter live ranges) and every step stalls a full LDS latency with the matrix pipe idle.
      load_step(0, a0, b0);
#pragma unroll 2
      for (int st = 0; st < S; st += 2) {
        load_step(st + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < S) load_step(st + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(1 % KWS, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      acc[0][0][0] = in_tile[offB[0]] + w_tile[offA];
#endif
    }
  }

  PGV_TICK(3);
#if defined(PGV_EXP) && (PGV_EXP == 1 || PGV_EXP == 2)
  if (acc[0][0][0] == 12345.678f) out[0] = 1.f;
  return;
#endif
  // ---- epilogue.  D layout: col = lane&15 (pixel), row = (lane>>4)*4 + reg (channel inside the M tile).
  // bias + activation in registers, per-channel statistics by shuffles, then the band goes through LDS (the input
  // tile is dead by now) so that it leaves as 16-byte stores of contiguous NCHW segments.
  constexpr int PS = 64 * NT + 4;  // out_tile row stride: % 8 == 4 -> the four lane groups write disjoint banks
  constexpr int EM = MT > 2 ? 2 : MT;  // M tiles per pass through LDS (keeps the out tile within the input tile)
  float* out_tile = lds + 8;
#pragma unroll
  for (int m0 = 0; m0 < MT; m0 += EM) {
    __syncthreads();  // every wave is done with the input tile / the previous pass has been copied out
#pragma unroll
    for (int mm = 0; mm < EM; ++mm) {
      const int m = m0 + mm;
      float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int cl = m * 16 + (lane >> 4) * 4 + reg;
        const float bv = (bias && cl < d.Cs) ? bias[cl] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int p = (wave * NT + t) * 16 + (lane & 15);
          const float v = pgv_act(acc[m][t][reg] + bv, act, slope);
          out_tile[(cl - m0 * 16) * PS + p] = v;
          if (p < Pb) {
            s[reg] += v;
            q[reg] = fmaf(v, v, q[reg]);
          }
        }
      }
      if (stats) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float ss = group16_sum(s[reg]), qq = group16_sum(q[reg]);
          if ((lane & 15) == 0) {
            const int cl = m * 16 + (lane >> 4) * 4 + reg;
            st_tile[(wave * MT * 16 + cl) * 2 + 0] = ss;
            st_tile[(wave * MT * 16 + cl) * 2 + 1] = qq;
          }
        }
      }
    }
    if (m0 + EM >= MT) PGV_TICK(4);
    __syncthreads();
    const int nch = min(d.Cs - m0 * 16, EM * 16);
    if (nch > 0)
      store_rows_contig(out_tile, PS,
                        out + ((int64_t)b * d.Cs + m0 * 16) * d.Hs * d.Ws + (int64_t)oh0 * d.Ws,
                        (int64_t)d.Hs * d.Ws, nch, Pb, tid);
  }
  if (stats && tid < MT * 16 && tid < d.Cs) {
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      ss += (double)st_tile[(wv * MT * 16 + tid) * 2 + 0];
      qq += (double)st_tile[(wv * MT * 16 + tid) * 2 + 1];
    }
    atomicAdd(&stats[tid], ss);
    atomicAdd(&stats[d.Cs + tid], qq);
  }
  PGV_TICK(5);
}

template <int KS, int MT, int NT, int CK>
int launch_down(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift, const float* w,
                const float* bias, int act, float slope, float* out, double* stats, hipStream_t st) {
  constexpr int KWS = (KS + 3) / 4, KWP = KWS * 4, CSP = MT * 16 + 1, KC = CK * KS * KWP;
  int R = min(d->Hs, (64 * NT) / d->Ws);
  if (R < 1) return 0;
  size_t bytes = 0;
  int plane = 0;
  for (; R >= 1; --R) {
    plane = ((2 * (R - 1) + KS) * d->Wb + 8 + 3) / 4 * 4;
    // the epilogue re-uses the tile region for the [MT*16][64*NT+4] output band
    plane = max(plane, ((MT > 2 ? 2 : MT) * 16 * (64 * NT + 4) + CK - 1) / CK + 3) / 4 * 4;
    bytes = sizeof(float) * (8 + (size_t)CK * plane + (size_t)KC * CSP + 4 * MT * 16 * 2 + 2 * d->Cb);
    if (bytes <= (size_t)kLdsTarget || R == 1) break;
  }
  if (bytes > (size_t)kMaxLds) return 0;
#ifdef PGV_ONE_WG
  bytes = max(bytes, (size_t)90 * 1024);
#endif
  auto kern = conv_down_mfma_kernel<KS, MT, NT, CK>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_down_mfma");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_mfma: memset failed");
    return PGV_E_LAUNCH;
  }
  dim3 grid((unsigned)pgv_cdiv(d->Hs, R), (unsigned)d->B);
  hipLaunchKernelGGL(kern, grid, dim3(256), bytes, st, *d, big, in_scale, in_shift, w, bias, act, slope, out, stats, R,
                     plane);
  PGV_CHECK_LAUNCH("conv_down_mfma");
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// UP  (sub-pixel phases; see the header comment)
// ---------------------------------------------------------------------------------------------------------------
// KS = 4: T = 2 taps per axis, one MFMA k-group per input channel with k = th*2 + tw.
// KS = 5: T = 3 taps per axis, three k-groups per input channel (group = th) with k = tw padded to 4 (zero weight).
template <int KS, int MT, int NT, int CK>
__global__ __launch_bounds__(256, 2) void conv_up_mfma_kernel(pgv_conv_desc d, const float* __restrict__ small_in,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int act, float slope, float* __restrict__ out,
                                                           double* __restrict__ stats, int R, int Wg, int Hg,
                                                           int plane) {
  constexpr int T = (KS + 1) / 2;         // taps per axis
  constexpr int KG = (KS == 4) ? 1 : T;   // MFMA k-groups per input channel
  constexpr int TWR = (KS == 4) ? 2 : 4;  // tw values a k-group reads along the row
  constexpr int MSP = MT * 16 + 1;        // padded [k][m] weight row
  constexpr int KC = CK * KG * 4;         // k rows per channel chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rows_in = R + T - 1;
  const int Ws = d.Ws;
  float* w_tile = lds + 8 + CK * plane;
  float* st_tile = w_tile + KC * MSP;     // [4 waves][MT*4][2]
  float* aff = st_tile + 4 * MT * 4 * 2;  // [2][Cs]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int u0 = blockIdx.x * R;
  const int rows_g = min(R, Hg - u0);
  const int Pb = rows_g * Wg;
  const int ih0 = u0 + 2 - T;  // input row of local row 0
  const float* src = small_in + (int64_t)b * d.Cs * d.Hs * Ws;
  // input tile with row stride = Ws (contiguous NCHW row segments), front slack for col < 0 reads (masked)
  const int lead = (max(ih0, 0) - ih0) * Ws;
  float* in_tile = lds + 4 + ((4 - (lead & 3)) & 3);

  int offB[NT];
  bool okB[NT];
  const float inv_wg = 1.0f / (float)Wg;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = (wave * NT + t) * 16 + (lane & 15);
    const int pv = p < Pb ? p : 0;
    const int ur = fast_div(pv, inv_wg), v = pv - ur * Wg;
    const int k = lane >> 4;
    const int col = (KS == 4) ? v + 1 - (k & 1) : v + 1 - k;
    const int row = (KS == 4) ? ur + 1 - (k >> 1) : ur + T - 1;
    offB[t] = row * Ws + col;
    okB[t] = (unsigned)col < (unsigned)Ws;
  }
  const int offA = (lane >> 4) * MSP + (lane & 15);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int M = 4 * d.Cb;
  stage_affine(aff, in_scale, in_shift, d.Cs, tid);
  if (in_scale) __syncthreads();
  for (int cs0 = 0; cs0 < d.Cs; cs0 += CK) {
    if (cs0) __syncthreads();
    stage_rows_contig<PGV_STAGE_U>(in_tile, plane, src, d.Cs, d.Hs, Ws, cs0, CK, rows_in, ih0, in_scale ? aff : nullptr,
                         aff + d.Cs, tid);
    // weights: w_tile[(c*KG + g)*4 + kk][m], m = cb*4 + ph*2 + pw, value w[cs][cb][ph+2th][pw+2tw]
    for (int i0 = tid; i0 < KC * MT * 16; i0 += 256 * 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * 256;
        const int krow = idx / (MT * 16), m = idx - krow * (MT * 16);
        const int c = krow / (KG * 4), rem = krow - c * (KG * 4);
        const int g = rem >> 2, kk = rem & 3;
        const int th = (KS == 4) ? (kk >> 1) : g, tw = (KS == 4) ? (kk & 1) : kk;
        const int cb = m >> 2, ph = (m >> 1) & 1, pw = m & 1;
        const int kh = ph + 2 * th, kw = pw + 2 * tw;
        const bool ok = idx < KC * MT * 16 && m < M && cs0 + c < d.Cs && kh < KS && kw < KS;
        v[u] = ok ? w[(((int64_t)(cs0 + c) * d.Cb + cb) * KS + kh) * KS + kw] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * 256;
        if (idx < KC * MT * 16) {
          const int krow = idx / (MT * 16), m = idx - krow * (MT * 16);
          w_tile[krow * MSP + m] = v[u];
        }
      }
    }
    __syncthreads();
    {
      constexpr int S = CK * KG;  // steps (c, g), g fastest; even for every instantiated (CK, KG)
      static_assert(S % 2 == 0, "step count must be even");
      float a0[MT], a1[MT], b0[NT], b1[NT];
      auto load_step = [&](int st, float (&av)[MT], float (&bv)[NT]) {
        const int c = st / KG, g = st - c * KG;
        const float* ap = w_tile + (st * 4) * MSP + offA;
        const float* bp = in_tile + c * plane - ((KS == 4) ? 0 : g * Ws);
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = ap[m * 16];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = bp[offB[t]];
      };
      auto compute_step = [&](const float (&av)[MT], float (&bv)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = okB[t] ? bv[t] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], acc[m][t], 0, 0, 0);
      };
      load_step(0, a0, b0);
#pragma unroll 2
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
    }
  }

  // ---- epilogue: lane owns channel cb = mt*4 + (lane>>4) at grid pixel (u,v); regs = (ph,pw)
  const bool vec2 = (d.Wb & 1) == 0;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int cb = m * 4 + (lane >> 4);
    float s = 0.f, q = 0.f;
    if (cb < d.Cb) {
      const float bv = bias ? bias[cb] : 0.f;
      float* obase = out + ((int64_t)b * d.Cb + cb) * d.Hb * d.Wb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p = (wave * NT + t) * 16 + (lane & 15);
        if (p < Pb) {
          const int ur = fast_div(p, inv_wg), v = p - ur * Wg;
          const int ow = 2 * v;
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
            const int oh = 2 * (u0 + ur) + ph;
            if (oh < d.Hb) {
              const float v0 = pgv_act(acc[m][t][ph * 2 + 0] + bv, act, slope);
              const float v1 = pgv_act(acc[m][t][ph * 2 + 1] + bv, act, slope);
              float* o = obase + (int64_t)oh * d.Wb + ow;
              if (ow + 1 < d.Wb) {
                if (vec2)
                  *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                else {
                  o[0] = v0;
                  o[1] = v1;
                }
                s += v0 + v1;
                q = fmaf(v0, v0, fmaf(v1, v1, q));
              } else if (ow < d.Wb) {
                o[0] = v0;
                s += v0;
                q = fmaf(v0, v0, q);
              }
            }
          }
        }
      }
    }
    if (stats) {
      const float ss = group16_sum(s), qq = group16_sum(q);
      if ((lane & 15) == 0) {
        st_tile[(wave * MT * 4 + m * 4 + (lane >> 4)) * 2 + 0] = ss;
        st_tile[(wave * MT * 4 + m * 4 + (lane >> 4)) * 2 + 1] = qq;
      }
    }
  }
  if (stats) {
    __syncthreads();
    if (tid < MT * 4 && tid < d.Cb) {
      double ss = 0.0, qq = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        ss += (double)st_tile[(wv * MT * 4 + tid) * 2 + 0];
        qq += (double)st_tile[(wv * MT * 4 + tid) * 2 + 1];
      }
      atomicAdd(&stats[tid], ss);
      atomicAdd(&stats[d.Cb + tid], qq);
    }
  }
}

template <int KS, int MT, int NT, int CK>
int launch_up(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
              const float* w, const float* bias, int act, float slope, float* out, double* stats, hipStream_t st) {
  constexpr int T = (KS + 1) / 2, KG = (KS == 4) ? 1 : T, TWR = (KS == 4) ? 2 : 4, MSP = MT * 16 + 1,
                KC = CK * KG * 4;
  const int Hg = (d->Hb + 1) / 2, Wg = (d->Wb + 1) / 2;
  int R = min(Hg, (64 * NT) / Wg);
  if (R < 1) return 0;
  size_t bytes = 0;
  int plane = 0;
  (void)TWR;
  for (; R >= 1; --R) {
    plane = ((R + T - 1) * d->Ws + 8 + 3) / 4 * 4;
    bytes = sizeof(float) * (8 + (size_t)CK * plane + (size_t)KC * MSP + 4 * MT * 4 * 2 + 2 * d->Cs);
    if (bytes <= (size_t)kLdsTarget || R == 1) break;
  }
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = conv_up_mfma_kernel<KS, MT, NT, CK>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_up_mfma");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_mfma: memset failed");
    return PGV_E_LAUNCH;
  }
  dim3 grid((unsigned)pgv_cdiv(Hg, R), (unsigned)d->B);
  hipLaunchKernelGGL(kern, grid, dim3(256), bytes, st, *d, small_in, in_scale, in_shift, w, bias, act, slope, out,
                     stats, R, Wg, Hg, plane);
  PGV_CHECK_LAUNCH("conv_up_mfma");
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// WGRAD:  gw[cs][cb][kh][kw] = sum_{b,oh,ow} small[b,cs,oh,ow] * big[b,cb,2oh-2+kh,2ow-2+kw]
// GEMM with M = cs, N = (cb, tap) — one 16-wide N tile per (cb, 16 taps) — and K = output pixels, 4 consecutive ow
// per MFMA.  A[cs][pixel] comes from the small tile, B[pixel][tap] again straight from the raw big tile:
// lane (tap j, pixel k) reads  cb*plane + (2r+kh)*Wt + 2*(ow0+k) + kw.
// Workgroups are persistent over (sample, band) units: accumulators stay in registers across units and are flushed
// once at the end with float atomics (Cs*Cb*k*k values per wave), so atomic traffic is grid-size x weight-size, not
// unit-count x weight-size.  Waves split the N tiles WN ways and the pixel steps 4/WN ways.
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int MT, int NB, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_mfma_kernel(pgv_conv_desc d, const float* __restrict__ big,
                                                              const float* __restrict__ big_scale,
                                                              const float* __restrict__ big_shift,
                                                              const float* __restrict__ small_in,
                                                              const float* __restrict__ small_scale,
                                                              const float* __restrict__ small_shift,
                                                              float* __restrict__ gw, int R, int plane, int WsP, int SP,
                                                              int bands, int units) {
  constexpr int KK = KS * KS;
  constexpr int NTAP_T = (KK + 15) / 16;  // N tiles per big channel
  constexpr int WK = 4 / WN;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rows_in = 2 * (R - 1) + KS;
  const int Wb = d.Wb, Ws = d.Ws;
  float* small_tile = lds + 8 + d.Cb * plane;  // [Cs][SP]: R rows of Ws floats, contiguous (SP % 4 == 0)
  float* aff_b = small_tile + d.Cs * SP;       // [2][Cb]
  float* aff_s = aff_b + 2 * d.Cb;         // [2][Cs]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gn = wave % WN, wk = wave / WN;
  const int n_tiles = d.Cb * NTAP_T;

  int offB[NB], colB[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    int nt = gn * NB + n;
    if (nt >= n_tiles) nt = 0;
    const int cb = nt / NTAP_T;
    const int tau = (nt - cb * NTAP_T) * 16 + (lane & 15);
    const int kh = tau < KK ? tau / KS : 0, kw = tau < KK ? tau - (tau / KS) * KS : 0;
    colB[n] = kw - d.pad + 2 * (lane >> 4);  // + 2*ow0 = image column of this lane's B element
    offB[n] = cb * plane + kh * Wb + colB[n];
  }
  int offA[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) offA[m] = min(m * 16 + (lane & 15), d.Cs - 1) * SP + (lane >> 4);

  f32x4 acc[MT][NB];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_affine(aff_b, big_scale, big_shift, d.Cb, tid);
  stage_affine(aff_s, small_scale, small_shift, d.Cs, tid);
  const int steps_per_row = WsP / 4;
  const int ow_hi = (Wb + d.pad - KS) / 2 - 3;  // ow0 <= ow_hi: all 4 pixels x all taps of the step are inside the row
  __syncthreads();
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int b = u / bands, band = u - b * bands;
    const int oh0 = band * R;
    const int rows_out = min(R, d.Hs - oh0);
    const int ih0 = oh0 * 2 - d.pad;
    const int lead = (max(ih0, 0) - ih0) * Wb;
    float* big_tile = lds + 4 + ((4 - (lead & 3)) & 3);  // [Cb][plane], row stride Wb
    if (u != (int)blockIdx.x) __syncthreads();
    if (u == (int)blockIdx.x) PGV_TICK(0);
    stage_rows_contig<PGV_STAGE_U>(big_tile, plane, big + (int64_t)b * d.Cb * d.Hb * Wb, d.Cb, d.Hb, Wb, 0, d.Cb, rows_in, ih0,
                         big_scale ? aff_b : nullptr, aff_b + d.Cb, tid);
    stage_rows_contig<PGV_STAGE_U>(small_tile, SP, small_in + (int64_t)b * d.Cs * d.Hs * Ws, d.Cs, d.Hs, Ws, 0, d.Cs, R, oh0,
                         small_scale ? aff_s : nullptr, aff_s + d.Cs, tid);
    if (u == (int)blockIdx.x) PGV_TICK(1);
    __syncthreads();
    if (u == (int)blockIdx.x) PGV_TICK(2);
    const int S = rows_out * steps_per_row;
    // pixel steps of this wave: st = wk, wk+WK, ... ; LDS reads of the next step are issued before the MFMAs of the
    // current one (two register sets, ping-pong)
    auto load_step = [&](int st, float (&av)[MT], float (&bv)[NB]) {
      const int r = st / steps_per_row, ow0 = (st - r * steps_per_row) * 4;
      const float* ap = small_tile + r * Ws + ow0;
      const float* bp = big_tile + 2 * r * Wb + 2 * ow0;
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = ap[offA[m]];
#pragma unroll
      for (int n = 0; n < NB; ++n) bv[n] = bp[offB[n]];
    };
    auto compute_step = [&](int st, float (&av)[MT], float (&bv)[NB]) {
      const int r = st / steps_per_row, ow0 = (st - r * steps_per_row) * 4;
      if (!(ow0 >= 1 && ow0 <= ow_hi && ow0 + 4 <= Ws)) {
        // row ends: pixels beyond Ws contribute nothing, columns outside the image are the zero padding
        const bool a_ok = ow0 + (lane >> 4) < Ws;
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = a_ok ? av[m] : 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) bv[n] = ((unsigned)(2 * ow0 + colB[n]) < (unsigned)Wb) ? bv[n] : 0.f;
      }
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
    };
    {
      float a0[MT], a1[MT], b0[NB], b1[NB];
      int st = wk;
      if (st < S) load_step(st, a0, b0);
      for (; st < S; st += 2 * WK) {
        const bool has1 = st + WK < S;
        if (has1) load_step(st + WK, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        compute_step(st, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (has1) {
          if (st + 2 * WK < S) load_step(st + 2 * WK, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          compute_step(st + WK, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (u == (int)blockIdx.x) PGV_TICK(3);
  }
  PGV_TICK(4);
  // ---- flush: D col = lane&15 = tap within the N tile, row = (lane>>4)*4 + reg = cs within the M tile
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int nt = gn * NB + n;
    if (nt < n_tiles) {
      const int cb = nt / NTAP_T;
      const int tau = (nt - cb * NTAP_T) * 16 + (lane & 15);
      if (tau < KK) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int cs = m * 16 + (lane >> 4) * 4 + reg;
            if (cs < d.Cs) atomicAdd(&gw[((int64_t)cs * d.Cb + cb) * KK + tau], acc[m][n][reg]);
          }
        }
      }
    }
  }
  PGV_TICK(5);
}

template <int KS, int MT, int NB, int WN>
int launch_wgrad(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                 const float* small_in, const float* small_scale, const float* small_shift, float* gw, hipStream_t st) {
  constexpr int KK = KS * KS;
  const int WsP = (d->Ws + 3) / 4 * 4;
  int R = min(d->Hs, 4);
  size_t bytes = 0;
  int SP = 0, plane = 0;
  for (; R >= 1; --R) {
    SP = (R * d->Ws + 8 + 3) / 4 * 4;
    SP += (36 - (SP % 32)) % 32;  // SP % 32 == 4: 16-byte aligned channel rows, at most 2-way A-read conflicts
    plane = ((2 * (R - 1) + KS) * d->Wb + 16 + 3) / 4 * 4;
    bytes = sizeof(float) * (8 + (size_t)d->Cb * plane + (size_t)d->Cs * SP + 2 * (size_t)(d->Cb + d->Cs));
    if (bytes <= (size_t)kLdsTarget || R == 1) break;
  }
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = conv_wgrad_mfma_kernel<KS, MT, NB, WN>;
  static bool attr_done = false;
  int rc = raise_lds_limit(kern, &attr_done, "conv_wgrad_mfma");
  if (rc) return rc;
  if (!(d->flags & PGV_PREZEROED) && hipMemsetAsync(gw, 0, sizeof(float) * (size_t)d->Cs * d->Cb * KK, st) != hipSuccess) {
    pgv_set_error("conv_wgrad_mfma: memset failed");
    return PGV_E_LAUNCH;
  }
  const int bands = (int)pgv_cdiv(d->Hs, R);
  const int units = bands * d->B;
  if (units == 0) return 1;
  const int per_cu = (int)max((size_t)1, min((size_t)2, (size_t)kMaxLds / bytes));
  const int grid = min(units, 256 * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, st, *d, big, big_scale, big_shift, small_in, small_scale,
                     small_shift, gw, R, plane, WsP, SP, bands, units);
  PGV_CHECK_LAUNCH("conv_wgrad_mfma");
  return 1;
}

}  // namespace

int pgv_conv_down_tuned(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                        const float* w, const float* bias, int act, float slope, float* small_out, double* stats,
                        hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != d->kw) return 0;
  if (d->kh == 4) {
    if (d->Cs <= 16)
      return launch_down<4, 1, 6, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, st);
    if (d->Cs <= 32)
      return launch_down<4, 2, 3, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, st);
    if (d->Cs <= 64)
      return launch_down<4, 4, 3, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, st);
    return 0;
  }
  if (d->kh == 5) {
    if (d->Cs <= 16)
      return launch_down<5, 1, 12, 1>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, st);
    return 0;
  }
  return 0;
}

int pgv_conv_up_tuned(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                      const float* w, const float* bias, int act, float slope, float* big_out, double* stats,
                      hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != d->kw) return 0;
  if (d->kh == 4) {
    if (d->Cb <= 8)
      return launch_up<4, 2, 6, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, st);
    if (d->Cb <= 16)
      return launch_up<4, 4, 3, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, st);
    if (d->Cb <= 32)
      return launch_up<4, 8, 3, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, st);
    return 0;
  }
  if (d->kh == 5) {
    if (d->Cb <= 4)
      return launch_up<5, 1, 6, 8>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, st);
    return 0;
  }
  return 0;
}

int64_t pgv_conv_wgrad_tuned_workspace(const pgv_conv_desc*) { return 0; }

int pgv_conv_wgrad_tuned(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                         const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                         void* /*workspace*/, int64_t /*workspace_bytes*/, hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != d->kw) return 0;
#define PGV_WG(KS, MT, NB, WN) \
  return launch_wgrad<KS, MT, NB, WN>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, st)
  if (d->kh == 4) {
    // N tiles = Cb; a wave group covers NB of them, WN groups cover NB*WN >= Cb
    if (d->Cs <= 16 && d->Cb <= 8) PGV_WG(4, 1, 8, 1);
    if (d->Cs <= 32 && d->Cb <= 16) PGV_WG(4, 2, 8, 2);
    if (d->Cs <= 64 && d->Cb <= 32) PGV_WG(4, 4, 8, 4);
    return 0;
  }
  if (d->kh == 5) {
    if (d->Cs <= 16 && d->Cb <= 2) PGV_WG(5, 1, 4, 1);  // 2 tap tiles per big channel
    return 0;
  }
#undef PGV_WG
  return 0;
}
