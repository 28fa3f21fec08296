// Raw-plane implicit-GEMM kernels for the DEEP k4 s2 p2 layers of speccnn8l1_bn (enc5..enc7, dec2..dec4;
// model/encoder.py:249-255, model/decoder.py:205-210): planes of 17x23, 9x12 and 5x7 pixels against 64..512 channels,
// i.e. K = 1024..4096 - dense contractions whose operands are the weights (up to 8 MB per layer, streamed through LDS
// in K-slabs) and a few whole input planes per workgroup.
//
// Unlike the gather-GEMM of conv_gemm.hip nothing is expanded (im2col) or gathered element-wise from global memory:
//   * both operands arrive as contiguous 16-byte global loads (a weight row slab W[cs][cb0..cb0+CK][16] is CK*64
//     contiguous bytes, the CK planes of a sample are CK*H*W contiguous floats), one K-slab ahead of the MFMA loop in
//     registers, committed to the other half of a double-buffered LDS stage (one barrier per slab);
//   * planes sit in LDS zero-padded ([ch][sample][HP][WP]), so the B fragment of output pixel n and tap (kh,kw) is
//     LDS[base(n) + ch*stride + kh*WP + kw]: a per-lane base computed once plus compile-time immediates, and the
//     zero padding of the convolution needs no masks;
//   * the MFMA k index is the kernel tap: lane group j = lane>>4 holds kernel row kh = j, so one ds_read_b128 of the
//     weight row gives the A operands of the four k-steps (kw = 0..3) of a channel.
// PGV_COMPUTE_BF16: the same tiles, operands packed to bf16 while they are read (v_cvt_pk_bf16_f32), one
// v_mfma_f32_16x16x32_bf16 per (channel pair, tile) instead of eight fp32 steps (two 16-deep steps per operand, conv_tile.h).
#include "conv_tile.h"
#include "conv_deep_common.h"

#ifdef PGV_DEEP_STAMPS
__device__ unsigned long long g_deep_stamps[8 * 64 * 8];
extern "C" int pgv_debug_read_stamps(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_deep_stamps), sizeof(unsigned long long) * n);
}
#define DEEP_STAMP(s, k)                                                                                  \
  do {                                                                                                    \
    if (blockIdx.x == PGV_DEEP_STAMPS && lane == 0 && (s) < 64) g_deep_stamps[((threadIdx.x >> 6) * 64 + (s)) * 8 + (k)] = clock64(); \
  } while (0)
#else
#define DEEP_STAMP(s, k)
#endif

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
// DOWN: out[b,cs,oh,ow] = act(bias[cs] + sum_{cb,kh,kw} w[cs,cb,kh,kw] * x'[b,cb,2oh-2+kh,2ow-2+kw])
// GEMM: M = cs (64 per workgroup, 16 per wave), N = the NS*Hs*Ws output pixels of NS samples, K = (cb, 16 taps).
template <int H, int W, int NS, int CK, int NTHR = 512>
struct DeepDown {
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, HW = H * W;
  // rows / cols -2 .. 2*Hs-1 / 2*Ws-1.  Row and plane strides are padded so that the 16 lanes of a B-fragment read
  // (ds_read2_b64: banks (a/4) mod 32, 16 contiguous lanes per LDS cycle; lane = output pixel n -> address
  // si*PLANE + (2 oh + kh)*WP + 2 ow) fall on distinct banks: with WP = Ws (mod 16) and PLANE/2 = P (mod 16) the bank pair
  // of pixel n is n mod 16.  Unpadded (WP = 2 Ws + 2) the reads of the 5x7 / 9x12 / 17x23 layers took 2.0 / 2.6 / 1.9
  // LDS cycles per group instead of 1 (SQ_LDS_BANK_CONFLICT = 0.41-0.57 of SQ_LDS_IDX_ACTIVE); 9x12 (odd Ws) keeps 1.6.
  static constexpr int HP = 2 * Hs + 2;
  static constexpr int WP = (H == 5 && W == 7) ? 12 : (H == 9 && W == 12) ? 26 : (H == 17 && W == 23) ? 28 : 2 * Ws + 2;
  static constexpr int PLANE = (H == 5 && W == 7) ? 104 : (H == 9 && W == 12) ? 330 : HP * WP;
  static_assert(WP >= 2 * Ws + 2 && PLANE >= HP * WP, "padded plane");
  static constexpr int N = NS * P, NT = (N + 15) / 16;
  static constexpr int AS = CK * 16 + 4;                 // weight row stride: 16-byte aligned, banks spread by 4
  static constexpr int A_FLOATS = 64 * AS;
  static constexpr int CH_STRIDE = NS * PLANE;
  static constexpr int B_FLOATS = CK * CH_STRIDE;
  static constexpr int STAGE = (A_FLOATS + B_FLOATS + 3) / 4 * 4;
  static constexpr int QA = 64 * CK * 4 / NTHR;           // float4 weight loads per thread per slab
  static constexpr int QB_ITEMS = NS * CK * HW / 4;      // float4 plane loads per slab (whole workgroup)
  static constexpr int QB = (QB_ITEMS + NTHR - 1) / NTHR;
  static_assert(CK % 4 == 0 && (CK * HW) % 4 == 0 && (64 * CK * 4) % NTHR == 0, "16-byte plane runs");
  static_assert(PLANE % 2 == 0 && WP % 2 == 0, "8-byte aligned tap rows");
};

template <int H, int W, int NS, int CK, bool BF16>
__global__ __launch_bounds__(512) void deep_down_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                        const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int act, float slope, float* __restrict__ out,
                                                        double* __restrict__ stats, int groups, int stat_stride,
                                                        pgv_bn_src in_bn) {
  using G = DeepDown<H, W, NS, CK>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* aff = lds + 2 * G::STAGE;  // [2*CB]
  // 8 waves = 2 (M: 32 output channels each, two 16-row MFMA tiles) x 4 (K: the channels of every slab dealt over four
  // wave groups kg, partial tiles added up through LDS before the epilogue).  While one wave of a SIMD streams MFMAs, the
  // SIMD's other waves get an issue slot only every ~70 clocks (conv_v2_common.h): what bounds these kernels is the
  // number of non-MFMA instructions (fragment reads, their address arithmetic, waits) per MFMA.  Two M tiles per wave use
  // every B fragment for 8 MFMAs instead of 4; four K groups keep 4 waves per SIMD with ONE channel per wave and slab.
  constexpr int NTHR = 512, CPW = CK / 4;
  static_assert(CK % 4 == 0, "channels of a slab over four wave groups");
  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, mw = wave8 & 1, kg = wave8 >> 1;
  const int m = lane & 15, j = lane >> 4;
  int mb, grp;
  deep_block(CS / 64, groups, mb, grp);
  const int cs0 = mb * 64, b0 = grp * NS;

  // zero both stages' planes once (the data cells are rewritten every slab, the padding never)
  for (int i = tid; i < G::B_FLOATS; i += NTHR) {
    lds[G::A_FLOATS + i] = 0.f;
    lds[G::STAGE + G::A_FLOATS + i] = 0.f;
  }
  for (int i = tid; i < CB; i += NTHR) {   // identity when the input carries no folded BatchNorm: the commit is branch-free
    float sc = 1.f, sh = 0.f;
    // (pgv_conv_down_bn: the producer's BatchNorm is finalized here, from its statistics, instead of by a launch of its own)
    if (in_bn.stats)
      pgv_bn_finalize_dev(in_bn, CB, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[CB + i] = sh;
  }

  // ---- loader coordinates (identical for every slab)
  int a_src[G::QA], a_dst[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = tid + NTHR * i, row = q / (CK * 4), f = q - row * (CK * 4);
    a_src[i] = (cs0 + row) * CB * 16 + 4 * f;
    a_dst[i] = row * G::AS + 4 * f;
  }
  int b_src[G::QB], b_dst[G::QB][4], b_ch[G::QB][4];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + NTHR * i, G::QB_ITEMS - 1);
    b_ok[i] = tid + NTHR * i < G::QB_ITEMS;
    const int si = q / (CK * G::HW / 4), qq = q - si * (CK * G::HW / 4);
    const int bs = min(b0 + si, B - 1);  // partial last group: duplicate the last sample (masked at the store)
    b_src[i] = bs * CB * G::HW + 4 * qq;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int el = 4 * qq + e, ch = el / G::HW, rem = el - ch * G::HW, r = rem / W, c = rem - r * W;
      b_ch[i][e] = ch;
      b_dst[i][e] = ch * G::CH_STRIDE + si * G::PLANE + (r + 2) * G::WP + c + 2;
    }
  }
  // ---- fragment coordinates
  const int a_frag = (mw * 32 + m) * G::AS + j * 4;   // second M tile: + 16 rows
  int bn[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t) {
    const int n = min(t * 16 + m, G::N - 1);
    const int si = n / G::P, pix = n - si * G::P, oh = pix / G::Ws, ow = pix - oh * G::Ws;
    bn[t] = si * G::PLANE + (2 * oh + j) * G::WP + 2 * ow;
  }
  f32x4 acc[2][G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t) acc[0][t] = acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Loads run TWO slabs ahead of the MFMA loop in two register sets, and the commit of slab s+1 sits in the MIDDLE of slab
  // s's MFMA block: its loads are then 1.5 slabs old (no wait), its LDS writes issue under the running matrix pipe, and
  // the only non-MFMA time of a slab is its barrier.  (With one set, issued at the top of the slab and committed after the
  // block, the commit phase waited for the loads and the two co-resident workgroups of a CU - which move in lockstep -
  // left the pipe idle together: MFMA time and the load / commit skeleton of the kernel ADDED UP, 46 + 53 of 102 us on
  // the 17x23 layer.)
  struct RegSet {
    f32x4 a[G::QA], b[G::QB];
  };
  RegSet r0, r1;
  // (the loads are inline asm with manual s_waitcnt: the compiler's counter model merges the two in-flight sets at the
  // loop header and waits for BOTH at every commit - vmcnt(4..0) where vmcnt(9..5) is meant - which halves the distance)
  constexpr int NLD = G::QA + G::QB;
  auto issue = [&](int slab, RegSet& r) {
    const int cb0 = slab * CK;
#pragma unroll
    for (int i = 0; i < G::QA; ++i)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.a[i]) : "v"((a_src[i] + cb0 * 16) * 4), "s"(w) : "memory");
#pragma unroll
    for (int i = 0; i < G::QB; ++i)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.b[i]) : "v"((b_src[i] + cb0 * G::HW) * 4), "s"(big) : "memory");
  };
  // `younger`: the other set has been requested after this one and may stay in flight
  auto wait_set = [&](RegSet& r, bool younger) {
    if (younger)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < G::QA; ++i) asm volatile("" : "+v"(r.a[i]));   // the values exist from here on
#pragma unroll
    for (int i = 0; i < G::QB; ++i) asm volatile("" : "+v"(r.b[i]));
  };
  // The folded-BatchNorm affine of the slab to commit is fetched from the LDS table at the START of the slab step, all
  // 8 QB values at once: fetched inside the commit (one dependent LDS round trip per element, in a branch each) the
  // commit of the plane data took 1500 (5x7) / 3200 (17x23) clocks of a 4900 / 9800-clock slab, with the other waves of
  // the workgroup waiting at the barrier and the co-resident workgroup in the same phase.
  float bsc[G::QB][4], bsh[G::QB][4];
  auto fetch_aff = [&](int slab) {
    const int cb0 = slab * CK;
#pragma unroll
    for (int i = 0; i < G::QB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bsc[i][e] = aff[cb0 + b_ch[i][e]];
        bsh[i][e] = aff[CB + cb0 + b_ch[i][e]];
      }
  };
  auto commit = [&](float* st, const RegSet& r) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) *reinterpret_cast<f32x4*>(st + a_dst[i]) = r.a[i];
    float* bt = st + G::A_FLOATS;
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(r.b[i][e], bsc[i][e], bsh[i][e]);
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bt[b_dst[i][e]] = v[e];
      }
    }
  };

  const int nslab = CB / CK;
  // one slab: MFMA block over stage (s & 1); half way, slab s+1 (in `rn`) goes to the other stage and slab s+3 is
  // requested into the registers it frees
  auto slab_step = [&](int s, RegSet& rn) {
    const float* st = lds + (s & 1) * G::STAGE;
    const float* ap = st + a_frag;
    const float* bp = st + G::A_FLOATS;
    DEEP_STAMP(s, 0);
    fetch_aff(min(s + 1, nslab - 1));
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the fetch to the commit)
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      const int ch = kg * CPW + cc;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + ch * 16);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + 16 * G::AS + ch * 16);
      f32x4 bf[G::NT];
#pragma unroll
      for (int t = 0; t < G::NT; ++t) {
        const float* p = bp + bn[t] + ch * G::CH_STRIDE;
        const f32x2 lo = *reinterpret_cast<const f32x2*>(p), hi = *reinterpret_cast<const f32x2*>(p + 2);
        bf[t] = f32x4{lo[0], lo[1], hi[0], hi[1]};
      }
      auto mfmas = [&](const f32x4& a, f32x4 (&ac)[G::NT]) {
        if constexpr (BF16) {
          // (one channel per wave and slab: the K = 32 instruction runs with its upper half zero - the same 16 cycles the
          // legacy K = 16 form took; pairing channels needs CK = 8 slabs, twice the stage)
          const s16x4 av = pack_bf16x4(a[0], a[1], a[2], a[3]);
#pragma unroll
          for (int t = 0; t < G::NT; ++t)
            ac[t] = mfma_bf16_k32(av, zero_bf16x4(), pack_bf16x4(bf[t][0], bf[t][1], bf[t][2], bf[t][3]), zero_bf16x4(), ac[t]);
        } else {
#pragma unroll
          for (int kw = 0; kw < 4; ++kw)
#pragma unroll
            for (int t = 0; t < G::NT; ++t) ac[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kw], bf[t][kw], ac[t], 0, 0, 0);
        }
      };
      mfmas(a0, acc[0]);
      if (cc == CPW - 1 && s + 1 < nslab) {   // the next slab goes to the other stage between the two M tiles
        DEEP_STAMP(s, 1);
        wait_set(rn, s + 2 < nslab);
        DEEP_STAMP(s, 2);
        commit(lds + ((s + 1) & 1) * G::STAGE, rn);
        if (s + 3 < nslab) issue(s + 3, rn);
        DEEP_STAMP(s, 3);
      }
      mfmas(a1, acc[1]);
    }
    DEEP_STAMP(s, 4);
    __syncthreads();
    DEEP_STAMP(s, 5);
  };
  issue(0, r0);
  if (nslab > 1) issue(1, r1);
  __syncthreads();  // planes zeroed, affine staged
  fetch_aff(0);
  wait_set(r0, nslab > 1);
  commit(lds, r0);
  if (nslab > 2) issue(2, r0);
  __syncthreads();
  for (int s = 0; s < nslab; s += 2) {
    slab_step(s, r1);
    if (s + 1 < nslab) slab_step(s + 1, r0);
  }

  // ---- the K groups' partial tiles -> group 0, one group per round in a fixed order (deterministic sums; the last
  // slab's barrier has passed: the stages are free)
  {
    f32x4* red = reinterpret_cast<f32x4*>(lds);
#pragma unroll 1
    for (int r = 1; r < 4; ++r) {
      if (kg == r) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int t = 0; t < G::NT; ++t) red[((mt * G::NT + t) * 2 + mw) * 64 + lane] = acc[mt][t];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int t = 0; t < G::NT; ++t) acc[mt][t] += red[((mt * G::NT + t) * 2 + mw) * 64 + lane];
      }
      __syncthreads();
    }
  }
  // ---- epilogue: acc[mt][t][i] = channel cs0 + mw*32 + mt*16 + 4j + i, pixel n = t*16 + m.  The tile is assembled as
  // [sample][channel][P] in the free stages and copied out by all 8 waves: a sample's 64 planes are one contiguous run of
  // the output (written from the accumulators - 64-byte pieces P floats apart - the output moved at ~0.5 TB/s: one slab +
  // prologue + epilogue cost 39 / 25 / 17 us for 14 / 7 / 3 MB of output).
  const pgv_act_params ap = pgv_act_setup(act, slope);
  static_assert(NS * 64 * G::P <= 2 * G::STAGE, "output tile fits the stages");
  if (stats) stats += (size_t)(blockIdx.x & (PGV_CLS_COPIES - 1)) * stat_stride;   // PGV_STATS_COPIES: this XCD's partial copy
  if (kg == 0) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float bv[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      const int cl = mw * 32 + mt * 16 + 4 * j, c0 = cs0 + cl;
#pragma unroll
      for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[c0 + i] : 0.f;
#pragma unroll
      for (int t = 0; t < G::NT; ++t) {
        const int n = t * 16 + m;
        const int si = n / G::P, pix = n - si * G::P;
        const bool ok = n < G::N && b0 + si < B;
        float* o = lds + (si * 64 + cl) * G::P + pix;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = pgv_act_apply(acc[mt][t][i] + bv[i], ap);
          if (ok) {
            o[i * G::P] = v;
            s1[i] += v;
            s2[i] += v * v;
          }
        }
      }
      if (stats) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a1 = group16_sum(s1[i]), a2 = group16_sum(s2[i]);
          if (m == 0) {
            atomicAdd(&stats[c0 + i], (double)a1);
            atomicAdd(&stats[CS + c0 + i], (double)a2);
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int si = 0; si < NS; ++si) {
    if (b0 + si < B) {
      float* dst = out + ((int64_t)(b0 + si) * CS + cs0) * G::P;
      const float* src = lds + si * 64 * G::P;
      for (int i = tid; i < 64 * G::P; i += NTHR) dst[i] = src[i];
    }
  }
}

template <int H, int W, int NS, int CK>
int launch_deep_down(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* out, double* stats,
                     hipStream_t st, const pgv_bn_src* bn = nullptr) {
  using G = DeepDown<H, W, NS, CK>;
  if (d->Cs % 64 || d->Cb % CK) return 0;
  // (the loader's inline-asm loads carry 32-bit byte offsets from the tensor bases)
  if ((int64_t)d->B * d->Cb * G::HW * 4 >= (int64_t)1 << 31 || (int64_t)d->Cs * d->Cb * 64 >= (int64_t)1 << 31) return 0;
  const size_t bytes = sizeof(float) * (2 * G::STAGE + 2 * (size_t)d->Cb + 8);
  if (bytes > (size_t)kMaxLds) return 0;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  auto kern = bf16 ? deep_down_kernel<H, W, NS, CK, true> : deep_down_kernel<H, W, NS, CK, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[bf16], "conv_down_deep");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_deep: memset failed");
    return PGV_E_LAUNCH;
  }
  const int groups = (d->B + NS - 1) / NS;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * (d->Cs / 64))), dim3(512), bytes, st, d->B, d->Cb, d->Cs, big,
                     in_scale, in_shift, w, bias, act, slope, out, stats, groups,
                     (d->flags & PGV_STATS_COPIES) ? 2 * d->Cs : 0, bn ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("conv_down_deep");
  return 1;
}


// ---------------------------------------------------------------------------------------------------------------
// UP: out[b,cb,ih,iw] = act(bias[cb] + sum_{cs,kh,kw} w[cs,cb,kh,kw] * s'[b,cs,oh,ow]),  ih = 2oh-2+kh, iw = 2ow-2+kw.
// Output pixel (ih,iw) = (2u+ph, 2v+pw) only meets the taps kh = ph+2th, kw = pw+2tw (th,tw in {0,1}) at
// oh = u+1-th, ow = v+1-tw: four 2x2-tap convolutions, one per output phase.  GEMM: M = cb (64 per workgroup),
// K = (cs, 4 taps), N = the output pixels of NS samples, listed phase by phase (each 16-pixel tile belongs to one
// phase, so its weight taps are compile-time).  Weights are permuted to [cs][cb][phase][tap] while they are committed.
template <int H, int W, int NS, int CK>
struct DeepUp {
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, HW = H * W;
  static constexpr int SWP = Ws + 1, SPLANE = ((Hs + 1) * SWP + 1) / 2 * 2;  // zero row below / column right
  static constexpr int hu(int p) { return (p >> 1) ? H / 2 : (H + 1) / 2; }
  static constexpr int wu(int p) { return (p & 1) ? W / 2 : (W + 1) / 2; }
  static constexpr int cnt(int p) { return NS * hu(p) * wu(p); }
  static constexpr int ntp(int p) { return (cnt(p) + 15) / 16; }
  static constexpr int tile0(int p) { return p == 0 ? 0 : tile0(p - 1) + ntp(p - 1); }
  static constexpr int NT = tile0(4);
  static constexpr int AS = 20;                          // floats per (cs, cb) weight row: 16 + 4
  static constexpr int ACS = 64 * AS;                    // per small channel
  static constexpr int A_FLOATS = CK * ACS;
  static constexpr int CH_STRIDE = NS * SPLANE;
  static constexpr int B_FLOATS = CK * CH_STRIDE;
  static constexpr int STAGE = (A_FLOATS + B_FLOATS + 3) / 4 * 4;
  static constexpr int NTHR = 512;                      // 8 waves: 4 (M) x 2 (K groups)
  static constexpr int QA = CK * 64 * 4 / NTHR;
  static constexpr int QB_ITEMS = NS * CK * P / 4;
  static constexpr int QB = (QB_ITEMS + NTHR - 1) / NTHR;
  static_assert(CK % 8 == 0 && (CK * P) % 4 == 0 && (CK * 64 * 4) % NTHR == 0, "16-byte plane runs, two K groups");
};

template <int H, int W, int NS, int CK, bool BF16>
__global__ __launch_bounds__(512) void deep_up_kernel(int B, int CB, int CS, const float* __restrict__ small_in,
                                                      const float* __restrict__ in_scale,
                                                      const float* __restrict__ in_shift,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      int act, float slope, float* __restrict__ out,
                                                      double* __restrict__ stats, int groups, int stat_stride,
                                                      pgv_bn_src in_bn) {
  using G = DeepUp<H, W, NS, CK>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* aff = lds + 2 * G::STAGE;  // [2*CS]
  // 8 waves = 4 (M: 16 output channels each) x 2 (K: the small channels of every slab split between two wave groups,
  // partial tiles added through LDS before the epilogue).  These launches are 256 workgroups (64 output channels x NS
  // samples): as 4-wave workgroups every SIMD of the chip held ONE wave, and everything that wave did besides MFMAs
  // (fragment reads, commit, barrier) was time without an MFMA.
  constexpr int NTHR = G::NTHR;
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, kg = tid >> 8;
  const int m = lane & 15, j = lane >> 4;
  int mb, grp;
  deep_block(CB / 64, groups, mb, grp);
  const int cb0 = mb * 64, b0 = grp * NS;

  for (int i = tid; i < G::B_FLOATS; i += NTHR) {
    lds[G::A_FLOATS + i] = 0.f;
    lds[G::STAGE + G::A_FLOATS + i] = 0.f;
  }
  for (int i = tid; i < CS; i += NTHR) {   // identity when the input carries no folded BatchNorm: branch-free commit
    float sc = 1.f, sh = 0.f;
    // (pgv_conv_up_bn: the producer's BatchNorm is finalized here, from its statistics, instead of by a launch of its own)
    if (in_bn.stats)
      pgv_bn_finalize_dev(in_bn, CS, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[CS + i] = sh;
  }

  // ---- loaders: weights W[cs][cb0+row][16] (one float4 = the four kw of a kernel row kh), planes s[b][cs][P]
  int a_src[G::QA], a_dst[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = tid + NTHR * i, c = q / 256, r = q - c * 256, row = r >> 2, kh = r & 3;
    a_src[i] = (c * CB + cb0 + row) * 16 + 4 * kh;
    // (kh, kw) -> phase (kh&1)*2 + (kw&1), tap (kh>>1)*2 + (kw>>1): element kw of this float4 goes to
    // a_dst + (kw&1)*4 + (kw>>1)
    a_dst[i] = c * G::ACS + row * G::AS + (kh & 1) * 8 + (kh >> 1) * 2;
  }
  int b_src[G::QB], b_dst[G::QB][4], b_ch[G::QB][4];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + NTHR * i, G::QB_ITEMS - 1);
    b_ok[i] = tid + NTHR * i < G::QB_ITEMS;
    const int si = q / (CK * G::P / 4), qq = q - si * (CK * G::P / 4);
    const int bs = min(b0 + si, B - 1);
    b_src[i] = bs * CS * G::P + 4 * qq;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int el = 4 * qq + e, ch = el / G::P, rem = el - ch * G::P, r = rem / G::Ws, c = rem - r * G::Ws;
      b_ch[i][e] = ch;
      b_dst[i][e] = ch * G::CH_STRIDE + si * G::SPLANE + r * G::SWP + c;
    }
  }
  // ---- fragment coordinates.  Lane group j is small channel 4g + j in both precisions: the lane's four taps are four
  // consecutive k values of ONE bf16 MFMA, or the operands of FOUR fp32 MFMAs (one per tap, k = the 4 channels) fetched
  // by one 16-byte weight read and two 8-byte plane reads.  (fp32 used to take the tap as k: one 4-byte read of each
  // operand per MFMA; with two waves per SIMD the non-MFMA instructions per MFMA are what bounds the kernel.)
  const int a_frag = (wave * 16 + m) * G::AS + j * G::ACS;
  int bn[G::NT];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int tt = 0; tt < G::ntp(p); ++tt) {
      const int n = min(tt * 16 + m, G::cnt(p) - 1);
      const int per = G::hu(p) * G::wu(p);
      const int si = n / per, rem = n - si * per, u = rem / G::wu(p), v = rem - u * G::wu(p);
      const int base = si * G::SPLANE + (u + 1) * G::SWP + v + 1;
      bn[G::tile0(p) + tt] = base + j * G::CH_STRIDE;
    }
  f32x4 acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 ra[G::QA], rb[G::QB];
  auto issue = [&](int slab) {
    const int cs0 = slab * CK;
#pragma unroll
    for (int i = 0; i < G::QA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(w + a_src[i] + cs0 * CB * 16);
#pragma unroll
    for (int i = 0; i < G::QB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(small_in + b_src[i] + cs0 * G::P);
  };
  // (the affine of the slab to commit is fetched from the LDS table at the START of the slab, see deep_down)
  float bsc[G::QB][4], bsh[G::QB][4];
  auto fetch_aff = [&](int slab) {
    const int cs0 = slab * CK;
#pragma unroll
    for (int i = 0; i < G::QB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bsc[i][e] = aff[cs0 + b_ch[i][e]];
        bsh[i][e] = aff[CS + cs0 + b_ch[i][e]];
      }
  };
  auto commit = [&](int slab, float* st) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) {
      float* a = st + a_dst[i];
      a[0] = ra[i][0];
      a[4] = ra[i][1];
      a[1] = ra[i][2];
      a[5] = ra[i][3];
    }
    float* bt = st + G::A_FLOATS;
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(rb[i][e], bsc[i][e], bsh[i][e]);
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bt[b_dst[i][e]] = v[e];
      }
    }
  };

#ifdef PGV_DEEP_UP_ONE_SLAB   // timing aid (scratch/): prologue + one slab + epilogue
  const int nslab = 1;
#else
  const int nslab = CS / CK;
#endif
  issue(0);
  __syncthreads();
  fetch_aff(0);
  commit(0, lds);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const float* st = lds + (s & 1) * G::STAGE;
    if (s + 1 < nslab) issue(s + 1);
    fetch_aff(min(s + 1, nslab - 1));
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the fetch to the commit)
    const float* ap = st + a_frag;
    const float* bp = st + G::A_FLOATS;
    if constexpr (BF16) {
      // v_mfma_f32_16x16x32_bf16: two 4-channel groups per instruction where a wave has two (zero upper half otherwise)
      constexpr int NG = CK / 8;
#pragma unroll
      for (int gg = 0; gg < NG; gg += 2) {
        const int g = kg * NG + gg;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 4 * g * G::ACS + 4 * p);
          const s16x4 av = pack_bf16x4(a[0], a[1], a[2], a[3]);
          s16x4 av2 = zero_bf16x4();
          if constexpr (NG > 1) {
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(ap + 4 * (g + 1) * G::ACS + 4 * p);
            av2 = pack_bf16x4(a2[0], a2[1], a2[2], a2[3]);
          }
#pragma unroll
          for (int tt = 0; tt < G::ntp(p); ++tt) {
            const int t = G::tile0(p) + tt;
            const float* q = bp + bn[t] + 4 * g * G::CH_STRIDE;
            s16x4 bv2 = zero_bf16x4();
            if constexpr (NG > 1) {
              const float* q2 = q + 4 * G::CH_STRIDE;
              bv2 = pack_bf16x4(q2[0], q2[-1], q2[-G::SWP], q2[-G::SWP - 1]);
            }
            acc[t] = mfma_bf16_k32(av, av2, pack_bf16x4(q[0], q[-1], q[-G::SWP], q[-G::SWP - 1]), bv2, acc[t]);
          }
        }
      }
      static_assert(NG == 1 || NG % 2 == 0, "channel groups of a slab in pairs");
    } else {
#pragma unroll
      for (int gg = 0; gg < CK / 8; ++gg) {
        const int g = kg * (CK / 8) + gg;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 4 * g * G::ACS + 4 * p);
          float bq[G::ntp(0)][4];   // (phase 0 has the most tiles)
#pragma unroll
          for (int tt = 0; tt < G::ntp(p); ++tt) {
            const float* q = bp + bn[G::tile0(p) + tt] + 4 * g * G::CH_STRIDE;
            bq[tt][0] = q[0], bq[tt][1] = q[-1], bq[tt][2] = q[-G::SWP], bq[tt][3] = q[-G::SWP - 1];
          }
#pragma unroll
          for (int tap = 0; tap < 4; ++tap)   // (tiles inside: consecutive MFMAs on different accumulators)
#pragma unroll
            for (int tt = 0; tt < G::ntp(p); ++tt)
              acc[G::tile0(p) + tt] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap], bq[tt][tap], acc[G::tile0(p) + tt], 0, 0, 0);
        }
      }
    }
    if (s + 1 < nslab) commit(s + 1, lds + ((s + 1) & 1) * G::STAGE);
    __syncthreads();
  }

  // ---- the two K groups' partial tiles -> waves 0-3, half of the tiles per round (the last slab's barrier has passed:
  // the stages are free; 17x23 has 26 tiles = 106 KB of partials against 90 KB of stages)
  {
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    constexpr int HALF = (G::NT + 1) / 2;
    static_assert((size_t)HALF * 4 * 64 * 16 <= sizeof(float) * 2 * (size_t)G::STAGE, "partial tiles fit the stages");
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (kg == 1) {
#pragma unroll
        for (int t = r * HALF; t < (r == 0 ? HALF : G::NT); ++t) red[((t - r * HALF) * 4 + wave) * 64 + lane] = acc[t];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int t = r * HALF; t < (r == 0 ? HALF : G::NT); ++t) acc[t] += red[((t - r * HALF) * 4 + wave) * 64 + lane];
      }
      __syncthreads();
    }
  }
  // ---- epilogue: acc[t][i] = channel cb0 + wave*16 + 4j + i, pixel n of phase p.  The output leaves through LDS: a
  // lane's pixels of one phase are every second float of a row, and written straight from the accumulators (4-byte
  // stores, 8 bytes apart, the other half of every line coming from another phase's tile much later) the output moved
  // at 0.55 TB/s - 46 of the 17x23 layer's 103 us (one slab + prologue + epilogue = 54 / 30 / 20 us for 25.6 / 14 / 9 MB
  // of output).  Per round, half of the 64 channels are assembled as [sample][channel][H*W] in the free stages and
  // copied out by all 8 waves as whole lines (a sample's 32 planes are one contiguous run of the output).
  const pgv_act_params ap = pgv_act_setup(act, slope);
  float bv[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int c0 = cb0 + wave * 16 + 4 * j;
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[c0 + i] : 0.f;
  static_assert(sizeof(float) * NS * 32 * G::HW <= sizeof(float) * 2 * (size_t)G::STAGE, "output half fits the stages");
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (kg == 0 && (wave >> 1) == h) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int tt = 0; tt < G::ntp(p); ++tt) {
          const int t = G::tile0(p) + tt;
          const int n = tt * 16 + m;
          const int per = G::hu(p) * G::wu(p);
          const int si = n / per, rem = n - si * per, u = rem / G::wu(p), v = rem - u * G::wu(p);
          const bool ok = n < G::cnt(p) && b0 + si < B;
          float* o = lds + (si * 32 + (wave & 1) * 16 + 4 * j) * G::HW + (2 * u + (p >> 1)) * W + 2 * v + (p & 1);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float val = pgv_act_apply(acc[t][i] + bv[i], ap);
            if (ok) {
              o[i * G::HW] = val;
              s1[i] += val;
              s2[i] += val * val;
            }
          }
        }
    }
    __syncthreads();
#pragma unroll
    for (int si = 0; si < NS; ++si) {
      if (b0 + si < B) {
        float* dst = out + ((int64_t)(b0 + si) * CB + cb0 + 32 * h) * G::HW;
        const float* src = lds + si * 32 * G::HW;
        for (int i = tid; i < 32 * G::HW; i += NTHR) dst[i] = src[i];
      }
    }
    __syncthreads();
  }
  if (kg == 1) return;
  if (stats) {
    stats += (size_t)(blockIdx.x & (PGV_CLS_COPIES - 1)) * stat_stride;   // PGV_STATS_COPIES: this XCD's partial copy
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a1 = group16_sum(s1[i]), a2 = group16_sum(s2[i]);
      if (m == 0) {
        atomicAdd(&stats[c0 + i], (double)a1);
        atomicAdd(&stats[CB + c0 + i], (double)a2);
      }
    }
  }
}

template <int H, int W, int NS, int CK>
int launch_deep_up(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                   const float* w, const float* bias, int act, float slope, float* out, double* stats,
                   hipStream_t st, const pgv_bn_src* bn = nullptr) {
  using G = DeepUp<H, W, NS, CK>;
  if (d->Cb % 64 || d->Cs % CK) return 0;
  const size_t bytes = sizeof(float) * (2 * G::STAGE + 2 * (size_t)d->Cs + 8);
  if (bytes > (size_t)kMaxLds) return 0;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  auto kern = bf16 ? deep_up_kernel<H, W, NS, CK, true> : deep_up_kernel<H, W, NS, CK, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[bf16], "conv_up_deep");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_deep: memset failed");
    return PGV_E_LAUNCH;
  }
  const int groups = (d->B + NS - 1) / NS;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * (d->Cb / 64))), dim3(G::NTHR), bytes, st, d->B, d->Cb, d->Cs, small_in,
                     in_scale, in_shift, w, bias, act, slope, out, stats, groups,
                     (d->flags & PGV_STATS_COPIES) ? 2 * d->Cb : 0, bn ? *bn : pgv_no_bn());
  PGV_CHECK_LAUNCH("conv_up_deep");
  return 1;
}


// ---------------------------------------------------------------------------------------------------------------
// WGRAD: gw[cs,cb,kh,kw] = sum_{b,oh,ow} s'[b,cs,oh,ow] * x'[b,cb,2oh-2+kh,2ow-2+kw]
// GEMM: M = cs (64 per workgroup), N = (cb, 16 taps) = one 16-column tile per big channel (CBT per workgroup),
// K = output pixels, 4 consecutive ow per MFMA (rows of the small plane padded to a multiple of 4 with zeros).
// A[cs][pixel] from the small planes, B[pixel][tap] straight from the zero-padded raw big planes: lane (tap, j) reads
// plane[(2oh+kh)*WP + 2(ow0+j) + kw].  A workgroup sweeps its range of samples SB at a time (double-buffered stages,
// register prefetch), keeps the CBT accumulator tiles in registers and flushes once with float atomics (split-K).
template <int H, int W, int SB, int CBT, bool BF16>
struct DeepWgrad {
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, HW = H * W;
  static constexpr int WsP = (Ws + 3) / 4 * 4, SPL = Hs * WsP, STEPS = SPL / 4;
  // bf16: one MFMA covers 4 steps (16 pixels): rows of the A tile padded (with zeros) to whole groups
  static constexpr int GROUPS = (STEPS + 3) / 4;
  static constexpr int SROW = BF16 ? 16 * GROUPS + 4  // 16-byte reads, rows 20 / 52 floats apart (mod 64): no conflicts
                                   : ((SPL % 64 == 12 || SPL % 64 == 44) ? SPL : SPL + 4);  // conflict-free 4-byte reads
  static constexpr int HP = 2 * Hs + 2, WP = 2 * WsP + 2, PLANE = HP * WP;
  static constexpr int A_FLOATS = SB * 64 * SROW;
  static constexpr int B_FLOATS = SB * CBT * PLANE;
  static constexpr int STAGE = (A_FLOATS + B_FLOATS + 3) / 4 * 4;
  static constexpr int QA_ITEMS = SB * 64 * P / 4, QA = (QA_ITEMS + 255) / 256;
  static constexpr int QB_ITEMS = SB * CBT * HW / 4, QB = (QB_ITEMS + 255) / 256;
  static_assert(CBT % 4 == 0, "16-byte plane runs");
};

template <int H, int W, int SB, int CBT, bool BF16>
__global__ __launch_bounds__(256) void deep_wgrad_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                         const float* __restrict__ big_scale,
                                                         const float* __restrict__ big_shift,
                                                         const float* __restrict__ small_in,
                                                         const float* __restrict__ small_scale,
                                                         const float* __restrict__ small_shift,
                                                         float* __restrict__ gw, int nblk, int per_split) {
  using G = DeepWgrad<H, W, SB, CBT, BF16>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, j = lane >> 4;
  const int ks = blockIdx.x / nblk, blk = blockIdx.x - ks * nblk;
  const int ncb = CB / CBT, mb = blk / ncb, nb = blk - mb * ncb;
  const int cs0 = mb * 64, cb0 = nb * CBT;
  const int bbeg = ks * per_split, bend = min(B, bbeg + per_split);

  for (int i = tid; i < 2 * G::STAGE; i += 256) lds[i] = 0.f;  // padding cells (never rewritten) must be zero

  // ---- loaders: s[b][cs0 .. cs0+64][P] and x[b][cb0 .. cb0+CBT][HW], both contiguous per sample
  int a_src[G::QA], a_dst[G::QA][4], a_ch[G::QA][4];
  bool a_ok[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = min(tid + 256 * i, G::QA_ITEMS - 1);
    a_ok[i] = tid + 256 * i < G::QA_ITEMS;
    const int si = q / (64 * G::P / 4), qq = q - si * (64 * G::P / 4);
    a_src[i] = si * CS * G::P + 4 * qq;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int el = 4 * qq + e, c = el / G::P, pix = el - c * G::P, oh = pix / G::Ws, ow = pix - oh * G::Ws;
      a_ch[i][e] = c;
      a_dst[i][e] = (si * 64 + c) * G::SROW + oh * G::WsP + ow;
    }
  }
  int b_src[G::QB], b_dst[G::QB][4], b_ch[G::QB][4];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 256 * i, G::QB_ITEMS - 1);
    b_ok[i] = tid + 256 * i < G::QB_ITEMS;
    const int si = q / (CBT * G::HW / 4), qq = q - si * (CBT * G::HW / 4);
    b_src[i] = si * CB * G::HW + 4 * qq;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int el = 4 * qq + e, c = el / G::HW, rem = el - c * G::HW, r = rem / W, cc = rem - r * W;
      b_ch[i][e] = c;
      b_dst[i][e] = (si * CBT + c) * G::PLANE + (r + 2) * G::WP + cc + 2;
    }
  }
  const float* sbase = small_in + (int64_t)cs0 * G::P;
  const float* xbase = big + (int64_t)cb0 * G::HW;
  f32x4 ra[G::QA], rb[G::QB];
  // a stage holds samples b .. b+SB-1; samples at or beyond `bend` are loaded from the last valid one and committed
  // as zeros on the small side (their products vanish)
  auto issue = [&](int b) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) {
      const int si = a_src[i] / (CS * G::P);
      const int bs = min(b + si, bend - 1);
      ra[i] = *reinterpret_cast<const f32x4*>(sbase + (int64_t)(bs - si) * CS * G::P + a_src[i]);
    }
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int si = b_src[i] / (CB * G::HW);
      const int bs = min(b + si, bend - 1);
      rb[i] = *reinterpret_cast<const f32x4*>(xbase + (int64_t)(bs - si) * CB * G::HW + b_src[i]);
    }
  };
  // The folded-BatchNorm affines of a thread's elements do not change over the sample loop (its channels are fixed):
  // fetched once into registers (identity where an operand has none).  Fetched inside the commit - a global load per
  // element behind a branch - every slab paid QA*4 + QB*4 dependent memory round trips (see deep_down).
  float asc[G::QA][4], ash[G::QA][4], bsc[G::QB][4], bsh[G::QB][4];
#pragma unroll
  for (int i = 0; i < G::QA; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      asc[i][e] = small_scale ? small_scale[cs0 + a_ch[i][e]] : 1.f;
      ash[i][e] = small_scale ? small_shift[cs0 + a_ch[i][e]] : 0.f;
    }
#pragma unroll
  for (int i = 0; i < G::QB; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bsc[i][e] = big_scale ? big_scale[cb0 + b_ch[i][e]] : 1.f;
      bsh[i][e] = big_scale ? big_shift[cb0 + b_ch[i][e]] : 0.f;
    }
  auto commit = [&](int b, float* st) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) {
      const bool live = b + a_src[i] / (CS * G::P) < bend;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = live ? fmaf(ra[i][e], asc[i][e], ash[i][e]) : 0.f;
      if (a_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) st[a_dst[i][e]] = v[e];
      }
    }
    float* bt = st + G::A_FLOATS;
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(rb[i][e], bsc[i][e], bsh[i][e]);
      if (b_ok[i]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bt[b_dst[i][e]] = v[e];
      }
    }
  };

  // fp32: lane group j = pixel j of a 4-pixel step; bf16: lane group j = step 4g + j of a 16-pixel group, the
  // lane's four consecutive k values are that step's four pixels
  const int a_frag = (wave * 16 + m) * G::SROW + (BF16 ? 4 * j : j);
  const int b_frag = (m >> 2) * G::WP + (m & 3) + (BF16 ? 0 : 2 * j);  // tap (kh, kw) = (m>>2, m&3)
  int stepoff[BF16 ? G::GROUPS : 1];
  if constexpr (BF16) {
#pragma unroll
    for (int g = 0; g < G::GROUPS; ++g) {
      const int step = 4 * g + j, oh = (4 * step) / G::WsP, ow0 = 4 * step - oh * G::WsP;
      stepoff[g] = step < G::STEPS ? 2 * oh * G::WP + 2 * ow0 : 0;  // phantom steps: A is zero, any valid address
    }
  }
  f32x4 acc[CBT];
#pragma unroll
  for (int t = 0; t < CBT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (bbeg < bend) {
    issue(bbeg);
    __syncthreads();
    commit(bbeg, lds);
    __syncthreads();
    int stage = 0;
    for (int b = bbeg; b < bend; b += SB, stage ^= 1) {
      const float* st = lds + stage * G::STAGE;
      const bool more = b + SB < bend;
      if (more) issue(b + SB);
      const float* ap = st + a_frag;
      const float* bp = st + G::A_FLOATS + b_frag;
#pragma unroll
      for (int si = 0; si < SB; ++si) {
        if constexpr (BF16) {
          // v_mfma_f32_16x16x32_bf16: two 16-pixel groups of the sample per instruction (an odd last group: zero upper half)
#pragma unroll
          for (int g = 0; g < G::GROUPS; g += 2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + si * 64 * G::SROW + 16 * g);
            const s16x4 av = pack_bf16x4(a[0], a[1], a[2], a[3]);
            s16x4 av2 = zero_bf16x4();
            if (g + 1 < G::GROUPS) {
              const f32x4 a2 = *reinterpret_cast<const f32x4*>(ap + si * 64 * G::SROW + 16 * (g + 1));
              av2 = pack_bf16x4(a2[0], a2[1], a2[2], a2[3]);
            }
#pragma unroll
            for (int t = 0; t < CBT; ++t) {
              const float* q = bp + (si * CBT + t) * G::PLANE + stepoff[g];
              s16x4 bv2 = zero_bf16x4();
              if (g + 1 < G::GROUPS) {
                const float* q2 = bp + (si * CBT + t) * G::PLANE + stepoff[g + 1 < G::GROUPS ? g + 1 : g];
                bv2 = pack_bf16x4(q2[0], q2[2], q2[4], q2[6]);
              }
              acc[t] = mfma_bf16_k32(av, av2, pack_bf16x4(q[0], q[2], q[4], q[6]), bv2, acc[t]);
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < G::STEPS; ++i) {
            const int oh = (4 * i) / G::WsP, ow0 = (4 * i) - oh * G::WsP;
            const float a = ap[si * 64 * G::SROW + 4 * i];
#pragma unroll
            for (int t = 0; t < CBT; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[(si * CBT + t) * G::PLANE + 2 * oh * G::WP + 2 * ow0],
                                                            acc[t], 0, 0, 0);
          }
        }
      }
      if (more) commit(b + SB, lds + (stage ^ 1) * G::STAGE);
      __syncthreads();
    }
  }
  // ---- flush: acc[t][i] = gw[cs0 + wave*16 + 4j + i][cb0 + t][tap m]
#pragma unroll
  for (int t = 0; t < CBT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      atomicAdd(&gw[((int64_t)(cs0 + wave * 16 + 4 * j + i) * CB + cb0 + t) * 16 + m], acc[t][i]);
}

template <int H, int W, int SB, int CBT>
int launch_deep_wgrad(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                      const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                      hipStream_t st) {
  if (d->Cs % 64 || d->Cb % CBT) return 0;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  const size_t bytes = sizeof(float) * 2 * (size_t)(bf16 ? DeepWgrad<H, W, SB, CBT, true>::STAGE
                                                         : DeepWgrad<H, W, SB, CBT, false>::STAGE);
  if (bytes > (size_t)kMaxLds) return 0;
  auto kern = bf16 ? deep_wgrad_kernel<H, W, SB, CBT, true> : deep_wgrad_kernel<H, W, SB, CBT, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[bf16], "conv_wgrad_deep");
  if (rc) return rc;
  if (!(d->flags & PGV_PREZEROED) &&
      hipMemsetAsync(gw, 0, sizeof(float) * (size_t)d->Cs * d->Cb * 16, st) != hipSuccess) {
    pgv_set_error("conv_wgrad_deep: memset failed");
    return PGV_E_LAUNCH;
  }
  const int nblk = (d->Cs / 64) * (d->Cb / CBT);
  // split the minibatch until ~512 workgroups, whole stages per split
  int splits = (int)max((int64_t)1, min((int64_t)pgv_cdiv(512, nblk), pgv_cdiv(d->B, SB)));
  const int per_split = (int)(pgv_cdiv(pgv_cdiv(d->B, splits), SB) * SB);
  splits = (int)pgv_cdiv(d->B, per_split);
  hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * splits)), dim3(256), bytes, st, d->B, d->Cb, d->Cs, big, big_scale,
                     big_shift, small_in, small_scale, small_shift, gw, nblk, per_split);
  PGV_CHECK_LAUNCH("conv_wgrad_deep");
  return 1;
}


// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions of the features mixer / un-mixer (enc8 = Conv2d(512,2048,1), model/encoder.py:56-69, and
// dec1 = ConvTranspose2d(2048,512,1), model/decoder.py:72-75) on 3x4 planes: plain GEMMs over (sample, pixel).
//   DOWN  out[b,cs,p] = act(bias[cs] + sum_cb w[cs,cb] x'[b,cb,p])      A = w rows (K contiguous)        TRANSA = false
//   UP    out[b,cb,p] = act(bias[cb] + sum_cs w[cs,cb] s'[b,cs,p])      A = w^T: k-major rows of 64 cb   TRANSA = true
// M = 64 output channels per workgroup, N = NS samples x P pixels, K in slabs of CK input channels.  The MFMA k index
// is the input channel: lane group j holds channels 16g+4j .. +3 (fp32: one per k-step, bf16: the lane's 4 k values).
template <int P, int NS, int CK, bool TRANSA>
struct K1Fwd {
  static constexpr int N = NS * P, NT = (N + 15) / 16;
  static constexpr int AS = TRANSA ? 64 + 16 : CK + 4;   // conflict-free fragment reads (see kernel)
  static constexpr int A_FLOATS = TRANSA ? CK * AS : 64 * AS;
  static constexpr int CH_STRIDE = NS * P;
  static constexpr int B_FLOATS = CK * CH_STRIDE;
  static constexpr int STAGE = (A_FLOATS + B_FLOATS + 3) / 4 * 4;
  static constexpr int QA = 64 * CK / 4 / 256;
  static constexpr int QB_ITEMS = NS * CK * P / 4, QB = (QB_ITEMS + 255) / 256;
  static_assert(P % 4 == 0 && CK % 16 == 0 && (64 * CK / 4) % 256 == 0, "tile shapes");
};

template <int P, int NS, int CK, bool TRANSA, bool BF16>
__global__ __launch_bounds__(256) void k1_fwd_kernel(int B, int CIN, int COUT, const float* __restrict__ in,
                                                     const float* __restrict__ in_scale,
                                                     const float* __restrict__ in_shift,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     int act, float slope, float* __restrict__ out,
                                                     double* __restrict__ stats, int groups, int stat_stride) {
  using G = K1Fwd<P, NS, CK, TRANSA>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, j = lane >> 4;
  int mb, grp;
  deep_block(COUT / 64, groups, mb, grp);
  const int co0 = mb * 64, b0 = grp * NS;

  // ---- loaders.  Weights: !TRANSA w[co0+row][ci0 + 4f ..] (row-major [COUT][CIN]);
  //                        TRANSA  w[ci0+r][co0 + 4f ..]    (row-major [CIN][COUT])
  int a_src[G::QA], a_dst[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = tid + 256 * i;
    if (TRANSA) {
      const int r = q / 16, f = q - r * 16;
      a_src[i] = r * COUT + co0 + 4 * f;
      a_dst[i] = r * G::AS + 4 * f;
    } else {
      const int row = q / (CK / 4), f = q - row * (CK / 4);
      a_src[i] = (co0 + row) * CIN + 4 * f;
      a_dst[i] = row * G::AS + 4 * f;
    }
  }
  int b_src[G::QB], b_dst[G::QB], b_ch[G::QB];
  bool b_ok[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 256 * i, G::QB_ITEMS - 1);
    b_ok[i] = tid + 256 * i < G::QB_ITEMS;
    const int si = q / (CK * P / 4), qq = q - si * (CK * P / 4);
    const int ch = (4 * qq) / P, pix = 4 * qq - ch * P;
    const int bs = min(b0 + si, B - 1);
    b_src[i] = bs * CIN * P + 4 * qq;
    b_ch[i] = ch;
    b_dst[i] = ch * G::CH_STRIDE + si * P + pix;
  }
  int bn[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t) {
    const int n = min(t * 16 + m, G::N - 1);
    bn[t] = n + 4 * j * G::CH_STRIDE;  // n = si*P + pix is the offset inside one channel's [NS][P] block
  }
  const int a_frag = TRANSA ? 4 * j * G::AS + wave * 16 + m : (wave * 16 + m) * G::AS + 4 * j;
  f32x4 acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 ra[G::QA], rb[G::QB];
  float rsc[G::QB], rsh[G::QB];
  auto issue = [&](int slab) {
    const int ci0 = slab * CK;
#pragma unroll
    for (int i = 0; i < G::QA; ++i)
      ra[i] = *reinterpret_cast<const f32x4*>(w + a_src[i] + (TRANSA ? (int64_t)ci0 * COUT : (int64_t)ci0));
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      rb[i] = *reinterpret_cast<const f32x4*>(in + b_src[i] + ci0 * P);
      rsc[i] = in_scale ? in_scale[ci0 + b_ch[i]] : 1.f;
      rsh[i] = in_scale ? in_shift[ci0 + b_ch[i]] : 0.f;
    }
  };
  auto commit = [&](int slab, float* st) {
    const int ci0 = slab * CK;
#pragma unroll
    for (int i = 0; i < G::QA; ++i) *reinterpret_cast<f32x4*>(st + a_dst[i]) = ra[i];
    float* bt = st + G::A_FLOATS;
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      if (b_ok[i]) {
        f32x4 v = rb[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], rsc[i], rsh[i]);   // (fetched with the slab's loads, a slab ahead)
        *reinterpret_cast<f32x4*>(bt + b_dst[i]) = v;
      }
    }
  };

  const int nslab = CIN / CK;
  issue(0);
  commit(0, lds);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const float* st = lds + (s & 1) * G::STAGE;
    if (s + 1 < nslab) issue(s + 1);
    const float* ap = st + a_frag;
    const float* bp = st + G::A_FLOATS;
#pragma unroll
    for (int g = 0; g < CK / 16; ++g) {
      f32x4 a;
      if (TRANSA) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = ap[(16 * g + e) * G::AS];
      } else {
        a = *reinterpret_cast<const f32x4*>(ap + 16 * g);
      }
      if constexpr (BF16) {
        // v_mfma_f32_16x16x32_bf16: the 16-channel groups g and g + 1 of the slab in one instruction (issued at even g)
        static_assert((CK / 16) % 2 == 0, "16-channel groups of a slab in pairs");
        if (g & 1) continue;
        f32x4 a2;
        if (TRANSA) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a2[e] = ap[(16 * (g + 1) + e) * G::AS];
        } else {
          a2 = *reinterpret_cast<const f32x4*>(ap + 16 * (g + 1));
        }
        const s16x4 av = pack_bf16x4(a[0], a[1], a[2], a[3]), av2 = pack_bf16x4(a2[0], a2[1], a2[2], a2[3]);
#pragma unroll
        for (int t = 0; t < G::NT; ++t) {
          const float* q = bp + bn[t] + 16 * g * G::CH_STRIDE;
          const float* q2 = q + 16 * G::CH_STRIDE;
          acc[t] = mfma_bf16_k32(av, av2, pack_bf16x4(q[0], q[G::CH_STRIDE], q[2 * G::CH_STRIDE], q[3 * G::CH_STRIDE]),
                                 pack_bf16x4(q2[0], q2[G::CH_STRIDE], q2[2 * G::CH_STRIDE], q2[3 * G::CH_STRIDE]), acc[t]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < G::NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bp[bn[t] + (16 * g + e) * G::CH_STRIDE], acc[t], 0, 0, 0);
      }
    }
    if (s + 1 < nslab) commit(s + 1, lds + ((s + 1) & 1) * G::STAGE);
    __syncthreads();
  }

  const pgv_act_params apar = pgv_act_setup(act, slope);
  float bv[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int c0 = co0 + wave * 16 + 4 * j;
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[c0 + i] : 0.f;
#pragma unroll
  for (int t = 0; t < G::NT; ++t) {
    const int n = t * 16 + m;
    const int si = n / P, pix = n - si * P;
    const bool ok = n < G::N && b0 + si < B;
    float* o = out + ((int64_t)(b0 + si) * COUT + c0) * P + pix;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = pgv_act_apply(acc[t][i] + bv[i], apar);
      if (ok) {
        o[i * P] = v;
        s1[i] += v;
        s2[i] += v * v;
      }
    }
  }
  if (stats) {
    stats += (size_t)(blockIdx.x & (PGV_CLS_COPIES - 1)) * stat_stride;   // PGV_STATS_COPIES: this XCD's partial copy
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a1 = group16_sum(s1[i]), a2 = group16_sum(s2[i]);
      if (m == 0) {
        atomicAdd(&stats[c0 + i], (double)a1);
        atomicAdd(&stats[COUT + c0 + i], (double)a2);
      }
    }
  }
}

// 8-wave form of the 1x1 DOWN kernel: 128 output channels x (NS*P) columns per 512-thread workgroup (waves 4 (M) x
// 2 (N)).  The 64-row form re-reads the input planes once per 64 output channels and is bound by the L2 -> LDS stream
// (enc8: 32 x 25 MB + 16 x 4 MB per launch ~ 10 TB/s); doubling the rows per workgroup halves the dominant term at the
// same two waves per SIMD.
template <int P, int NS, int CK, bool BF16>
__global__ __launch_bounds__(512) void k1_down128_kernel(int B, int CIN, int COUT, const float* __restrict__ in,
                                                         const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         int act, float slope, float* __restrict__ out,
                                                         double* __restrict__ stats, int groups, int stat_stride) {
  constexpr int N = NS * P, NT = (N + 15) / 16, NTW = (NT + 1) / 2;   // column tiles per wave (parity split)
  constexpr int AS = CK + 4, A_FLOATS = 128 * AS, CH_STRIDE = NS * P, B_FLOATS = CK * CH_STRIDE;
  constexpr int STAGE = (A_FLOATS + B_FLOATS + 3) / 4 * 4;
  constexpr int QA = 128 * CK / 4 / 512, QB_ITEMS = NS * CK * P / 4, QB = (QB_ITEMS + 511) / 512;
  static_assert((128 * CK / 4) % 512 == 0 && CK % 16 == 0 && P % 4 == 0, "tile shapes");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, j = lane >> 4, wm = wave & 3, wn = wave >> 2;
  int mb, grp;
  deep_block(COUT / 128, groups, mb, grp);
  const int co0 = mb * 128, b0 = grp * NS;

  int a_src[QA], a_dst[QA];
#pragma unroll
  for (int i = 0; i < QA; ++i) {
    const int q = tid + 512 * i, row = q / (CK / 4), f = q - row * (CK / 4);
    a_src[i] = (co0 + row) * CIN + 4 * f;
    a_dst[i] = row * AS + 4 * f;
  }
  int b_src[QB], b_dst[QB], b_ch[QB];
  bool b_ok[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i) {
    const int q = min(tid + 512 * i, QB_ITEMS - 1);
    b_ok[i] = tid + 512 * i < QB_ITEMS;
    const int si = q / (CK * P / 4), qq = q - si * (CK * P / 4);
    const int ch = (4 * qq) / P, pix = 4 * qq - ch * P;
    const int bs = min(b0 + si, B - 1);
    b_src[i] = bs * CIN * P + 4 * qq;
    b_ch[i] = ch;
    b_dst[i] = ch * CH_STRIDE + si * P + pix;
  }
  int bn[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) bn[t] = min((2 * t + wn) * 16 + m, N - 1) + 4 * j * CH_STRIDE;
  const int a_frag = (wm * 32 + m) * AS + 4 * j;
  f32x4 acc[2][NTW];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 ra[QA], rb[QB];
  float rsc[QB], rsh[QB];
  auto issue = [&](int slab) {
    const int ci0 = slab * CK;
#pragma unroll
    for (int i = 0; i < QA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(w + a_src[i] + ci0);
#pragma unroll
    for (int i = 0; i < QB; ++i) {
      rb[i] = *reinterpret_cast<const f32x4*>(in + b_src[i] + ci0 * P);
      rsc[i] = in_scale ? in_scale[ci0 + b_ch[i]] : 1.f;
      rsh[i] = in_scale ? in_shift[ci0 + b_ch[i]] : 0.f;
    }
  };
  auto commit = [&](int slab, float* st) {
    const int ci0 = slab * CK;
#pragma unroll
    for (int i = 0; i < QA; ++i) *reinterpret_cast<f32x4*>(st + a_dst[i]) = ra[i];
    float* bt = st + A_FLOATS;
#pragma unroll
    for (int i = 0; i < QB; ++i) {
      if (b_ok[i]) {
        f32x4 v = rb[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], rsc[i], rsh[i]);   // (fetched with the slab's loads, a slab ahead)
        *reinterpret_cast<f32x4*>(bt + b_dst[i]) = v;
      }
    }
  };
  const int nslab = CIN / CK;
  issue(0);
  commit(0, lds);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const float* st = lds + (s & 1) * STAGE;
    if (s + 1 < nslab) issue(s + 1);
    const float* ap = st + a_frag;
    const float* bp = st + A_FLOATS;
#pragma unroll
    for (int g = 0; g < CK / 16; ++g) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + 16 * g);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + 16 * AS + 16 * g);
      if constexpr (BF16) {
        // v_mfma_f32_16x16x32_bf16: the 16-channel groups g and g + 1 of the slab in one instruction (issued at even g)
        static_assert((CK / 16) % 2 == 0, "16-channel groups of a slab in pairs");
        if (g & 1) continue;
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(ap + 16 * (g + 1));
        const f32x4 a3 = *reinterpret_cast<const f32x4*>(ap + 16 * AS + 16 * (g + 1));
        const s16x4 av0 = pack_bf16x4(a0[0], a0[1], a0[2], a0[3]), av1 = pack_bf16x4(a1[0], a1[1], a1[2], a1[3]);
        const s16x4 av2 = pack_bf16x4(a2[0], a2[1], a2[2], a2[3]), av3 = pack_bf16x4(a3[0], a3[1], a3[2], a3[3]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const float* q = bp + bn[t] + 16 * g * CH_STRIDE;
          const float* q2 = q + 16 * CH_STRIDE;
          const s16x4 bv = pack_bf16x4(q[0], q[CH_STRIDE], q[2 * CH_STRIDE], q[3 * CH_STRIDE]);
          const s16x4 bv2 = pack_bf16x4(q2[0], q2[CH_STRIDE], q2[2 * CH_STRIDE], q2[3 * CH_STRIDE]);
          acc[0][t] = mfma_bf16_k32(av0, av2, bv, bv2, acc[0][t]);
          acc[1][t] = mfma_bf16_k32(av1, av3, bv, bv2, acc[1][t]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            const float bvv = bp[bn[t] + (16 * g + e) * CH_STRIDE];
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], bvv, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], bvv, acc[1][t], 0, 0, 0);
          }
      }
    }
    if (s + 1 < nslab) commit(s + 1, lds + ((s + 1) & 1) * STAGE);
    __syncthreads();
  }
  const pgv_act_params apar = pgv_act_setup(act, slope);
  if (stats) stats += (size_t)(blockIdx.x & (PGV_CLS_COPIES - 1)) * stat_stride;   // PGV_STATS_COPIES: this XCD's partial copy
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float bv[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int c0 = co0 + wm * 32 + r * 16 + 4 * j;
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = bias ? bias[c0 + i] : 0.f;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int n = (2 * t + wn) * 16 + m;
      const int si = n / P, pix = n - si * P;
      const bool ok = n < N && b0 + si < B;
      float* o = out + ((int64_t)(b0 + si) * COUT + c0) * P + pix;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = pgv_act_apply(acc[r][t][i] + bv[i], apar);
        if (ok) {
          o[i * P] = v;
          s1[i] += v;
          s2[i] += v * v;
        }
      }
    }
    if (stats) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a1 = group16_sum(s1[i]), a2 = group16_sum(s2[i]);
        if (m == 0) {
          atomicAdd(&stats[c0 + i], (double)a1);
          atomicAdd(&stats[COUT + c0 + i], (double)a2);
        }
      }
    }
  }
}

template <int P, int NS, int CK>
int launch_k1_down128(int B, int CIN, int COUT, int flags, const float* in, const float* in_scale, const float* in_shift,
                      const float* w, const float* bias, int act, float slope, float* out, double* stats,
                      hipStream_t st, const char* who) {
  if (COUT % 128 || CIN % CK) return 0;
  constexpr int STAGE = (128 * (CK + 4) + CK * NS * P + 3) / 4 * 4;
  const size_t bytes = sizeof(float) * 2 * (size_t)STAGE;
  const bool bf16 = (flags & PGV_COMPUTE_BF16) != 0;
  auto kern = bf16 ? k1_down128_kernel<P, NS, CK, true> : k1_down128_kernel<P, NS, CK, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[bf16], who);
  if (rc) return rc;
  if (stats && !(flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * COUT, st) != hipSuccess) {
    pgv_set_error("%s: memset failed", who);
    return PGV_E_LAUNCH;
  }
  const int groups = (B + NS - 1) / NS;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * (COUT / 128))), dim3(512), bytes, st, B, CIN, COUT, in, in_scale,
                     in_shift, w, bias, act, slope, out, stats, groups, (flags & PGV_STATS_COPIES) ? 2 * COUT : 0);
  PGV_CHECK_LAUNCH(who);
  return 1;
}

template <int P, int NS, int CK, bool TRANSA>
int launch_k1_fwd(int B, int CIN, int COUT, int flags, const float* in, const float* in_scale, const float* in_shift,
                  const float* w, const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                  const char* who) {
  using G = K1Fwd<P, NS, CK, TRANSA>;
  if (COUT % 64 || CIN % CK) return 0;
  const size_t bytes = sizeof(float) * 2 * (size_t)G::STAGE;
  const bool bf16 = (flags & PGV_COMPUTE_BF16) != 0;
  auto kern = bf16 ? k1_fwd_kernel<P, NS, CK, TRANSA, true> : k1_fwd_kernel<P, NS, CK, TRANSA, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[bf16], who);
  if (rc) return rc;
  if (stats && !(flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * COUT, st) != hipSuccess) {
    pgv_set_error("%s: memset failed", who);
    return PGV_E_LAUNCH;
  }
  const int groups = (B + NS - 1) / NS;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * (COUT / 64))), dim3(256), bytes, st, B, CIN, COUT, in, in_scale,
                     in_shift, w, bias, act, slope, out, stats, groups, (flags & PGV_STATS_COPIES) ? 2 * COUT : 0);
  PGV_CHECK_LAUNCH(who);
  return 1;
}

// WGRAD 1x1: gw[cs,cb] = sum_{b,p} s'[b,cs,p] * x'[b,cb,p].  M = cs (64 per workgroup), N = cb (CBT per workgroup),
// K = (sample, pixel) flattened: LDS rows [channel][SB*P] so that 4 (fp32) / 16 (bf16) consecutive k are contiguous.
template <int P, int SB, int CBT>
struct K1Wgrad {
  static constexpr int KS = SB * P, ROW = KS + 4, NT = CBT / 16;
  static constexpr int A_FLOATS = 64 * ROW, B_FLOATS = CBT * ROW, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int QA = SB * 64 * P / 4 / 256, QB = SB * CBT * P / 4 / 256;
  static_assert(KS % 16 == 0 && (SB * 64 * P / 4) % 256 == 0 && (SB * CBT * P / 4) % 256 == 0, "tile shapes");
};

template <int P, int SB, int CBT, bool BF16>
__global__ __launch_bounds__(256) void k1_wgrad_kernel(int B, int CB, int CS, const float* __restrict__ big,
                                                       const float* __restrict__ big_scale,
                                                       const float* __restrict__ big_shift,
                                                       const float* __restrict__ small_in,
                                                       const float* __restrict__ small_scale,
                                                       const float* __restrict__ small_shift,
                                                       float* __restrict__ gw, int nblk, int per_split) {
  using G = K1Wgrad<P, SB, CBT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, j = lane >> 4;
  const int ks = blockIdx.x / nblk, blk = blockIdx.x - ks * nblk;
  const int ncb = CB / CBT, mb = blk / ncb, nb = blk - mb * ncb;
  const int cs0 = mb * 64, cb0 = nb * CBT;
  const int bbeg = ks * per_split, bend = min(B, bbeg + per_split);

  int a_si[G::QA], a_off[G::QA], a_dst[G::QA], a_ch[G::QA];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    const int q = tid + 256 * i, si = q / (64 * P / 4), qq = q - si * (64 * P / 4);
    const int c = (4 * qq) / P, pix = 4 * qq - c * P;
    a_si[i] = si;
    a_off[i] = 4 * qq;
    a_ch[i] = c;
    a_dst[i] = c * G::ROW + si * P + pix;
  }
  int b_si[G::QB], b_off[G::QB], b_dst[G::QB], b_ch[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = tid + 256 * i, si = q / (CBT * P / 4), qq = q - si * (CBT * P / 4);
    const int c = (4 * qq) / P, pix = 4 * qq - c * P;
    b_si[i] = si;
    b_off[i] = 4 * qq;
    b_ch[i] = c;
    b_dst[i] = c * G::ROW + si * P + pix;
  }
  const float* sbase = small_in + (int64_t)cs0 * P;
  const float* xbase = big + (int64_t)cb0 * P;
  f32x4 ra[G::QA], rb[G::QB];
  auto issue = [&](int b) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i)
      ra[i] = *reinterpret_cast<const f32x4*>(sbase + (int64_t)min(b + a_si[i], bend - 1) * CS * P + a_off[i]);
#pragma unroll
    for (int i = 0; i < G::QB; ++i)
      rb[i] = *reinterpret_cast<const f32x4*>(xbase + (int64_t)min(b + b_si[i], bend - 1) * CB * P + b_off[i]);
  };
  // (a thread's channels are fixed over the sample loop: the affines are fetched once, identity where absent)
  float asc[G::QA], ash[G::QA], bsc[G::QB], bsh[G::QB];
#pragma unroll
  for (int i = 0; i < G::QA; ++i) {
    asc[i] = small_scale ? small_scale[cs0 + a_ch[i]] : 1.f;
    ash[i] = small_scale ? small_shift[cs0 + a_ch[i]] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    bsc[i] = big_scale ? big_scale[cb0 + b_ch[i]] : 1.f;
    bsh[i] = big_scale ? big_shift[cb0 + b_ch[i]] : 0.f;
  }
  auto commit = [&](int b, float* st) {
#pragma unroll
    for (int i = 0; i < G::QA; ++i) {
      f32x4 v = ra[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], asc[i], ash[i]);
      if (b + a_si[i] >= bend) v = f32x4{0.f, 0.f, 0.f, 0.f};  // samples beyond the range contribute nothing
      *reinterpret_cast<f32x4*>(st + a_dst[i]) = v;
    }
    float* bt = st + G::A_FLOATS;
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      f32x4 v = rb[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], bsc[i], bsh[i]);
      *reinterpret_cast<f32x4*>(bt + b_dst[i]) = v;
    }
  };

  const int a_frag = (wave * 16 + m) * G::ROW + (BF16 ? 4 * j : j);
  const int b_frag = m * G::ROW + (BF16 ? 4 * j : j);
  f32x4 acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (bbeg < bend) {
    issue(bbeg);
    commit(bbeg, lds);
    __syncthreads();
    int stage = 0;
    for (int b = bbeg; b < bend; b += SB, stage ^= 1) {
      const float* st = lds + stage * G::STAGE;
      const bool more = b + SB < bend;
      if (more) issue(b + SB);
      const float* ap = st + a_frag;
      const float* bp = st + G::A_FLOATS + b_frag;
      if constexpr (BF16) {
#pragma unroll
        for (int g = 0; g < G::KS / 16; g += 2) {   // v_mfma_f32_16x16x32_bf16: two 16-deep steps per instruction
          const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 16 * g);
          const s16x4 av = pack_bf16x4(a[0], a[1], a[2], a[3]);
          s16x4 av2 = zero_bf16x4();
          if (g + 1 < G::KS / 16) {
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(ap + 16 * (g + 1));
            av2 = pack_bf16x4(a2[0], a2[1], a2[2], a2[3]);
          }
#pragma unroll
          for (int t = 0; t < G::NT; ++t) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(bp + 16 * t * G::ROW + 16 * g);
            s16x4 bv2 = zero_bf16x4();
            if (g + 1 < G::KS / 16) {
              const f32x4 x2 = *reinterpret_cast<const f32x4*>(bp + 16 * t * G::ROW + 16 * (g + 1));
              bv2 = pack_bf16x4(x2[0], x2[1], x2[2], x2[3]);
            }
            acc[t] = mfma_bf16_k32(av, av2, pack_bf16x4(x[0], x[1], x[2], x[3]), bv2, acc[t]);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < G::KS / 4; ++i) {
          const float a = ap[4 * i];
#pragma unroll
          for (int t = 0; t < G::NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[16 * t * G::ROW + 4 * i], acc[t], 0, 0, 0);
        }
      }
      if (more) commit(b + SB, lds + (stage ^ 1) * G::STAGE);
      __syncthreads();
    }
  }
#pragma unroll
  for (int t = 0; t < G::NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      atomicAdd(&gw[(int64_t)(cs0 + wave * 16 + 4 * j + i) * CB + cb0 + 16 * t + m], acc[t][i]);
}

template <int P, int SB, int CBT>
int launch_k1_wgrad(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                    const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                    hipStream_t st) {
  using G = K1Wgrad<P, SB, CBT>;
  if (d->Cs % 64 || d->Cb % CBT) return 0;
  const size_t bytes = sizeof(float) * 2 * (size_t)G::STAGE;
  const bool bf16 = (d->flags & PGV_COMPUTE_BF16) != 0;
  auto kern = bf16 ? k1_wgrad_kernel<P, SB, CBT, true> : k1_wgrad_kernel<P, SB, CBT, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[bf16], "conv_wgrad_deep");
  if (rc) return rc;
  if (!(d->flags & PGV_PREZEROED) && hipMemsetAsync(gw, 0, sizeof(float) * (size_t)d->Cs * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_wgrad_deep: memset failed");
    return PGV_E_LAUNCH;
  }
  const int nblk = (d->Cs / 64) * (d->Cb / CBT);
  int splits = (int)max((int64_t)1, min((int64_t)pgv_cdiv(512, nblk), pgv_cdiv(d->B, SB)));
  const int per_split = (int)(pgv_cdiv(pgv_cdiv(d->B, splits), SB) * SB);
  splits = (int)pgv_cdiv(d->B, per_split);
  hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * splits)), dim3(256), bytes, st, d->B, d->Cb, d->Cs, big, big_scale,
                     big_shift, small_in, small_scale, small_shift, gw, nblk, per_split);
  PGV_CHECK_LAUNCH("conv_wgrad_deep");
  return 1;
}

bool shape_k4(const pgv_conv_desc* d) { return d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 2; }
bool shape_k1_3x4(const pgv_conv_desc* d) {
  return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->Hb == 3 && d->Wb == 4 && d->Cb >= 64;
}

}  // namespace

// ``bn`` (optional): the input's BatchNorm, finalized in the kernel's prologue (pgv_conv_down_bn / pgv_conv_up_bn); the
// 1x1 kernels fetch the affine per slab from global memory and do not take it: 0 is returned and the caller finalizes first.
int pgv_conv_down_deep(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                       const float* w, const float* bias, int act, float slope, float* out, double* stats,
                       hipStream_t st, const pgv_bn_src* bn) {
  if (int rc = pgv_conv_down_deep_bf16(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn)) return rc;
  if (int rc = pgv_conv_down_deep_split(d, big, in_scale, in_shift, bias, act, slope, out, stats, st, bn)) return rc;
  if (bn && shape_k1_3x4(d)) return 0;
  if (shape_k1_3x4(d) && d->Cs % 128 == 0)
    return launch_k1_down128<12, 16, 32>(d->B, d->Cb, d->Cs, d->flags, big, in_scale, in_shift, w, bias, act, slope, out,
                                         stats, st, "conv_down_deep");
  if (shape_k1_3x4(d))
    return launch_k1_fwd<12, 16, 32, false>(d->B, d->Cb, d->Cs, d->flags, big, in_scale, in_shift, w, bias, act, slope, out,
                                           stats, st, "conv_down_deep");
  if (!shape_k4(d) || d->Cb < 64) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_down<17, 23, 1, 4>(d, big, in_scale, in_shift, w, bias, act, slope, out, stats, st, bn);
  // (4 samples per workgroup: 140 pixels = 9 tiles, 3 % padding instead of 12.5 %, and half the weight traffic per sample;
  // 256 workgroups of 8 waves: 128 -> 104 us)
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_down<9, 12, 4, 4>(d, big, in_scale, in_shift, w, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_down<5, 7, 4, 4>(d, big, in_scale, in_shift, w, bias, act, slope, out, stats, st, bn);
  return 0;
}

int pgv_conv_up_deep(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* out, double* stats,
                     hipStream_t st, const pgv_bn_src* bn) {
  if (int rc = pgv_conv_up_deep_bf16(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn)) return rc;
  if (int rc = pgv_conv_up_deep_split(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, st, bn)) return rc;
  if (bn && shape_k1_3x4(d)) return 0;
  if (shape_k1_3x4(d))
    return launch_k1_fwd<12, 4, 64, true>(d->B, d->Cs, d->Cb, d->flags, small_in, in_scale, in_shift, w, bias, act, slope,
                                          out, stats, st, "conv_up_deep");
  if (!shape_k4(d) || d->Cb < 64) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_up<17, 23, 1, 8>(d, small_in, in_scale, in_shift, w, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_up<9, 12, 2, 8>(d, small_in, in_scale, in_shift, w, bias, act, slope, out, stats, st, bn);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_up<5, 7, 4, 8>(d, small_in, in_scale, in_shift, w, bias, act, slope, out, stats, st, bn);
  return 0;
}

int pgv_conv_wgrad_deep(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                        const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                        hipStream_t st) {
  if (d->B == 0) return 0;
  if (shape_k1_3x4(d))
    return launch_k1_wgrad<12, 4, 64>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, st);
  if (!shape_k4(d) || d->Cb < 64) return 0;
  if (d->Hb == 17 && d->Wb == 23) return launch_deep_wgrad<17, 23, 1, 4>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, st);
  if (d->Hb == 9 && d->Wb == 12) return launch_deep_wgrad<9, 12, 2, 8>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, st);
  if (d->Hb == 5 && d->Wb == 7) return launch_deep_wgrad<5, 7, 4, 8>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw, st);
  return 0;
}
