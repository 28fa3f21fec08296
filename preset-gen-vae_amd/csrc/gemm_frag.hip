// nn.Linear products (model/encoder.py:85, model/decoder.py:64: forward, input gradient, weight gradient) on the fp32
// matrix cores WITHOUT LDS staging and without workgroup barriers: every wave owns a macro tile of the output and loads its
// MFMA operand fragments straight from global memory, in the layout v_mfma_f32_16x16x4_f32 consumes them.
//
// Why: the six products of a step are either long-K with a tiny output (K = 25 024, split over K)
// or short-K (64 .. 256) with one huge extent (25 024).  The LDS-tiled kernel (gemm.hip) ran 4 - 9 K slabs per workgroup
// with all workgroups in lockstep through load -> barrier -> multiply -> store: 17 - 28 us per product against 5 - 10 us of
// fp32 matrix time, operands re-read 1.4 - 3.0 x.  Here a wave keeps three 16-deep K chunks of fragments in flight in
// registers (inline-asm loads, counted s_waitcnt), never waits for another wave, and the launch is a list of wave-sized jobs.
//
// Fragment loads.  MFMA operand element (index i = lane & 15, k = lane >> 4) of k-step e of a 16-deep chunk is given the K
// order k = 4 * (lane >> 4) + e inside the chunk (any bijection works as long as both operands use it), so that
//   * a K-CONTIGUOUS operand (x[m][k], W[n][k]) is ONE 16-byte load per 16-index tile and chunk: the lane's float4 holds its
//     element for the four k-steps;
//   * an INDEX-CONTIGUOUS operand (W[k][n], gy[b][o]: the contraction index is the row) is read IW = 4 (or 2) indices at a
//     time: load e of the chunk fetches row 4 * (lane >> 4) + e, indices IW * (lane & 15) .. + IW - 1, and serves k-step e of
//     IW interleaved tiles (tile c owns indices IW * i + c): four loads per chunk and group of IW tiles.
// Either way a load feeds four MFMAs and rows are read in contiguous pieces of 64 - 256 bytes.
//
// Output.  P = the N extent (contiguous in C), Q = the M extent.  D = P-fragment x Q-fragment puts 4 consecutive P indices
// of one Q index into a lane's accumulator, so stores are 16 bytes per lane (64 bytes per lane for interleaved P).  Split-K
// jobs add their partial tile with float atomics through a wave-private LDS transpose (256 contiguous bytes per wave
// instruction) into a C that holds zeros (PGV_PREZEROED) or the bias (gemm.hip's init_c_kernel).
#include "pgv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

template <int N>
struct VecOf;
template <>
struct VecOf<4> {
  typedef f32x4 type;
};
template <>
struct VecOf<2> {
  typedef f32x2 type;
};

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

__device__ __forceinline__ void frag_load(f32x4& dst, unsigned off, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void frag_load(f32x2& dst, unsigned off, const char* base) {
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// One operand side of a wave's macro tile: T tiles of 16 indices.
//   L = 1: K-contiguous rows (element (i, k) at base[i * ld + k]), loaded COALESCED - lane l fetches the 16 bytes of row l / 4,
//          k piece l % 4, so the four lanes of a quad read one 64-byte run - and moved to the MFMA's lane (i + 16 * piece) by
//          four ds_bpermute_b32 (the LDS crossbar, no LDS memory).  Measured against L = 0: the fragment-
//          shaped load puts 64 different rows-or-pieces into one instruction's consecutive lanes, which the texture
//          addresser handles one lane at a time - 1.4 TB/s for the encoder forward's operands with nothing but the loads.
//   L = 0: the same rows loaded fragment-shaped (lane (i, piece) fetches its own 16 bytes; kept for A/B timing).
//   L = 2 / 4: index-contiguous rows (element (i, k) at base[k * ld + i]) read L floats per lane.
// KD = K indices of a chunk: 16 (fp32: four k-steps of v_mfma_f32_16x16x4_f32) or 32 (PGV_COMPUTE_BF16: ONE
// v_mfma_f32_16x16x32_bf16 - the chunk is two 16-deep halves loaded exactly like two fp32 chunks, a lane's eight K indices are
// 4 kq + e and 16 + 4 kq + e, rounded to bfloat16 and packed when the matrix instruction reads them).
template <int L, int T, int KD = 16>
struct Side {
  static constexpr bool KC = L == 0 || L == 1;
  static constexpr int IW = KC ? 4 : L;
  static constexpr int NL0 = KC ? T : (T / IW) * 4;   // loads per 16-deep half
  static constexpr int NL = NL0 * (KD / 16);          // loads per chunk
  static_assert(KC || T % IW == 0, "interleaved groups");
  static_assert(KD == 16 || KD == 32, "chunk depth");
  typedef typename VecOf<IW>::type vec;
  unsigned off[NL];
  int perm;   // byte address of the source lane for ds_bpermute (L == 1)
  __device__ __forceinline__ void init(int lane, long long ld) {
    const int l16 = lane & 15, kq = lane >> 4;
    perm = 4 * (4 * l16 + kq);
#pragma unroll
    for (int jj = 0; jj < NL; ++jj) {
      const int j = jj % NL0;
      const long long half = jj / NL0 ? (KC ? 16 : 16 * ld) : 0;   // the second 16 K indices of a 32-deep chunk
      if (L == 0)
        off[jj] = (unsigned)((((long long)(16 * j + l16)) * ld + 4 * kq + half) * 4);
      else if (L == 1)
        off[jj] = (unsigned)((((long long)(16 * j + (lane >> 2))) * ld + 4 * (lane & 3) + half) * 4);
      else
        off[jj] = (unsigned)((((long long)(4 * kq + (j & 3))) * ld + 16 * IW * (j >> 2) + IW * l16 + half) * 4);
    }
  }
  // bytes from the operand's origin to (first index i0 of the macro tile, chunk kc)
  static __device__ __forceinline__ long long origin(long long i0, long long kc, long long ld) {
    return KC ? (i0 * ld + KD * kc) * 4 : (KD * kc * ld + i0) * 4;
  }
  // the chunk's registers as the MFMAs read them (L == 1: lane exchange; otherwise as loaded)
  __device__ __forceinline__ void arrange(vec (&r)[NL]) const {
    if constexpr (L == 1) {
#pragma unroll
      for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          r[j][e] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm, __float_as_int(r[j][e])));
    }
  }
  // value of tile t for k-step e (KD = 32: K slot e = 0 .. 7 of the lane) out of the chunk's registers
  static __device__ __forceinline__ float val(const vec (&r)[NL], int t, int e) {
    const int h = (e >> 2) * NL0;
    e &= 3;
    if (KC) return r[h + t][e];
    return r[h + (t / IW) * 4 + e][t % IW];
  }
  // the lane's eight K slots of tile t as packed bfloat16 (round to nearest even)
  static __device__ __forceinline__ u32x4 frag_bf16(const vec (&r)[NL], int t) {
    return u32x4{pack_bf16x2(val(r, t, 0), val(r, t, 1)), pack_bf16x2(val(r, t, 2), val(r, t, 3)),
                 pack_bf16x2(val(r, t, 4), val(r, t, 5)), pack_bf16x2(val(r, t, 6), val(r, t, 7))};
  }
  // index (inside the macro tile) of MFMA index i of tile t
  static __device__ __forceinline__ int index(int t, int i) {
    return KC ? 16 * t + i : 16 * IW * (t / IW) + IW * i + (t % IW);
  }
};

struct FragArgs {
  const float* P;
  const float* Q;
  float* C;
  const float* bias;   // on P (the N extent), or null
  long long ldp, ldq, ldc;
  int MPw, MQw, KS;    // wave jobs: P macro tiles x Q macro tiles x K splits
  int nchunks;         // K / 16
  int njobs;
  int atomic;          // 0: store (+ bias); 1: atomic add; 2: atomic add, the first K split adds the bias
  int wgred;           // split-K jobs: the four waves of a workgroup take four consecutive K splits of ONE tile and add them
                       // up through LDS (fixed order) before the atomics - a quarter of the atomic traffic (KS % 4 == 0)
  int dbg;             // tuning builds: 1 = no MFMAs, 2 = no stores
};

template <int PL, int QL, int TPW, int TQW, bool ATOMIC, int kStages, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void gemm_frag_kernel(FragArgs a) {
  typedef Side<PL, TPW, BF16 ? 32 : 16> SP;
  typedef Side<QL, TQW, BF16 ? 32 : 16> SQ;
  constexpr int NLP = SP::NL, NLQ = SQ::NL, NLT = NLP + NLQ;
  constexpr int PEXT = 16 * TPW, QEXT = 16 * TQW, TROW = PEXT + 4;
  static_assert((kStages - 1) * NLT <= 63, "vmcnt range");
  __shared__ __attribute__((aligned(16))) float red[4 * QEXT * TROW];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l16 = lane & 15, kq = lane >> 4;
  SP sp;
  SQ sq;
  sp.init(lane, a.ldp);
  sq.init(lane, a.ldq);
  // workgroups b, b + 8, ... share an XCD (round-robin placement): give an XCD a contiguous range of jobs, so that the jobs
  // that read the same operand slices (same K split, neighbouring tiles) share its L2.  Speed only.
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int per_split = a.MPw * a.MQw;
  typename SP::vec pf[kStages][NLP];
  typename SQ::vec qf[kStages][NLQ];
  const bool wgred = ATOMIC && a.wgred;
  const int jend = wgred ? a.njobs / 4 : a.njobs, jstep = wgred ? (int)gridDim.x : (int)gridDim.x * 4;
#pragma unroll 1
  for (int job = wgred ? bid : bid * 4 + wave; job < jend; job += jstep) {
    int ks = job / per_split;
    const int rem = job - ks * per_split;
    if (wgred) ks = 4 * ks + wave;   // (the same trip count for the four waves: the reduction below has workgroup barriers)
    const int pt = rem / a.MQw, qt = rem - pt * a.MQw;
    const int kc0 = a.nchunks * ks / a.KS, kc1 = a.nchunks * (ks + 1) / a.KS;   // (32-bit: nchunks * KS is small)
    const long long p0 = (long long)pt * PEXT, q0 = (long long)qt * QEXT;
    const char* pbase = reinterpret_cast<const char*>(a.P);
    const char* qbase = reinterpret_cast<const char*>(a.Q);
    auto issue = [&](auto sc, int kc) {
      constexpr int s = decltype(sc)::value;
      kc = min(kc, kc1 - 1);   // (the pipeline runs ahead unconditionally: the counted waits need a fixed number of loads)
      const char* pb = uniform_ptr(pbase + SP::origin(p0, kc, a.ldp));
      const char* qb = uniform_ptr(qbase + SQ::origin(q0, kc, a.ldq));
      sfor<0, NLP>([&](auto j) { frag_load(pf[s][decltype(j)::value], sp.off[decltype(j)::value], pb); });
      sfor<0, NLQ>([&](auto j) { frag_load(qf[s][decltype(j)::value], sq.off[decltype(j)::value], qb); });
    };
    f32x4 acc[TPW][TQW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int tq = 0; tq < TQW; ++tq) acc[tp][tq] = f32x4{0.f, 0.f, 0.f, 0.f};
    sfor<0, kStages - 1>([&](auto s) { issue(s, kc0 + decltype(s)::value); });
#pragma unroll 1
    for (int c = kc0; c < kc1; c += kStages) {
      sfor<0, kStages>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int cc = c + s;
        issue(std::integral_constant<int, (s + kStages - 1) % kStages>{}, cc + kStages - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kStages - 1) * NLT) : "memory");   // chunk cc has landed
        __builtin_amdgcn_sched_barrier(0);
        if (cc < kc1 && !(a.dbg & 1)) {
          sp.arrange(pf[s]);
          sq.arrange(qf[s]);
          if constexpr (BF16) {
            // one K = 32 instruction per tile pair and chunk: operands rounded to bfloat16 here, fp32 accumulation
            u32x4 bq[TQW];
#pragma unroll
            for (int tq = 0; tq < TQW; ++tq) bq[tq] = SQ::frag_bf16(qf[s], tq);
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp) {
              const u32x4 ap = SP::frag_bf16(pf[s], tp);
#pragma unroll
              for (int tq = 0; tq < TQW; ++tq)
                acc[tp][tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap), __builtin_bit_cast(bf16x8, bq[tq]),
                                                                       acc[tp][tq], 0, 0, 0);
            }
          } else {
          // round-robin over the accumulators: the f32 MFMA issues every 32 cycles but feeds a dependent one after 40
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tq = 0; tq < TQW; ++tq)
#pragma unroll
              for (int tp = 0; tp < TPW; ++tp)
                acc[tp][tq] = __builtin_amdgcn_mfma_f32_16x16x4f32(SP::val(pf[s], tp, e), SQ::val(qf[s], tq, e),
                                                                    acc[tp][tq], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-ahead loads: nothing may land after the registers are reused
    __builtin_amdgcn_sched_barrier(0);
    // acc[tp][tq][r]: P index SP::index(tp, 4 * kq + r), Q index SQ::index(tq, l16) of the macro tile
    if (a.dbg & 2) continue;
    // Both epilogues go through a wave-private LDS image of the macro tile [QEXT][PEXT (+4)], so that a wave instruction
    // covers whole rows: consecutive lanes = consecutive addresses of C (16 bytes per lane for stores, 4 for atomics).
    // Straight from the accumulators a store instruction put 16-byte pieces of 16 different rows into consecutive lanes:
    // the encoder's input gradient took 28.9 us with the stores and 20.7 without.
    {
      float* tile = red + wave * (QEXT * TROW);
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int tq = 0; tq < TQW; ++tq)
#pragma unroll
          for (int r = 0; r < 4; ++r) tile[SQ::index(tq, l16) * TROW + SP::index(tp, 4 * kq + r)] = acc[tp][tq][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if constexpr (ATOMIC) {
        const bool with_bias = a.atomic == 2 && ks == 0 && a.bias;
        constexpr int RPI = 64 / PEXT;   // C rows per wave instruction
        static_assert(PEXT <= 64 && 64 % PEXT == 0, "atomic epilogue: a wave instruction covers whole rows of the tile");
        const int pc = lane % PEXT, qr = lane / PEXT;
        if (wgred) {
          // the four images of this workgroup are four K splits of the same tile: every wave adds up a quarter of the rows
          // (splits in order 0 .. 3) and issues the atomics for it
          __syncthreads();
          const float bv = (a.atomic == 2 && ks < 4 && a.bias) ? a.bias[p0 + pc] : 0.f;
#pragma unroll 2
          for (int q = qr + RPI * wave; q < QEXT; q += 4 * RPI) {
            const float* t0 = red + q * TROW + pc;
            const float v = ((t0[0] + t0[QEXT * TROW]) + t0[2 * QEXT * TROW]) + t0[3 * QEXT * TROW];
            atomicAdd(a.C + (q0 + q) * a.ldc + p0 + pc, v + bv);
          }
          __syncthreads();
        } else {
        const float bv = with_bias ? a.bias[p0 + pc] : 0.f;
#pragma unroll 4
        for (int q = qr; q < QEXT; q += RPI) atomicAdd(a.C + (q0 + q) * a.ldc + p0 + pc, tile[q * TROW + pc] + bv);
        }
      } else {
        constexpr int LPR = PEXT / 4, RPI = 64 / LPR;   // lanes per row, rows per wave instruction
        static_assert(64 % LPR == 0 && QEXT % RPI == 0, "store epilogue");
        const int pc = 4 * (lane % LPR), qr = lane / LPR;
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) b = *reinterpret_cast<const f32x4*>(a.bias + p0 + pc);
#pragma unroll
        for (int q = qr; q < QEXT; q += RPI)
          *reinterpret_cast<f32x4*>(a.C + (q0 + q) * a.ldc + p0 + pc) = *reinterpret_cast<const f32x4*>(tile + q * TROW + pc) + b;
      }
      __builtin_amdgcn_wave_barrier();   // (the image is rewritten by this wave's next job)
    }
  }
}

int g_frag_variant = 0;   // tuning knob (pgv_dbg_set_gemm_variant): bits 0-1 stages (3 / 4 / 5), bits 2-3 grid cap (1024 / 768 /
                          // 512 / 256), bits 4-5 split-K job target (1024 / 2048 / 4096), bit 6 no MFMAs, bit 7 no stores,
                          // bit 9 fragment-shaped loads of K-contiguous operands, bit 10 everything to
                          // gemm.hip, bit 11 short-K forward products here too, bit 12 every covered bf16 shape here,
                          // bit 13 split-K jobs without the in-workgroup reduction

// SPLITK: the tiling also exists in its split-K (atomic epilogue) form
template <int PL, int QL, int TPW, int TQW, bool SPLITK, bool BF16 = false>
int launch_frag(FragArgs a, hipStream_t st) {
  const int wgs = (a.njobs + 3) / 4;
  const int caps[4] = {1024, 768, 512, 256};
  const int cap = caps[(g_frag_variant >> 2) & 3];
  int grid = wgs < cap ? wgs : cap;   // a few workgroups per CU: a wave's next job starts where the last one ended
  if (grid > 8) grid = (grid + 7) & ~7;
  a.dbg = (g_frag_variant >> 6) & 3;
  const int stages = 3 + (g_frag_variant & 3) % 3;
#define PGV_FRAG_GO(AT, ST)                                                                                       \
  hipLaunchKernelGGL((gemm_frag_kernel<PL, QL, TPW, TQW, AT, ST, BF16>), dim3(grid), dim3(256), 0, st, a)
  if constexpr (BF16) {   // (one pipeline depth: three 32-deep chunks in flight)
    if (a.atomic) {
      if constexpr (SPLITK) PGV_FRAG_GO(true, 3);
      else return 0;
    } else
      PGV_FRAG_GO(false, 3);
    (void)stages;
  } else if (a.atomic) {
    if constexpr (SPLITK) {
      if (stages == 3) PGV_FRAG_GO(true, 3);
      else if (stages == 4) PGV_FRAG_GO(true, 4);
      else PGV_FRAG_GO(true, 5);
    } else
      return 0;
  } else {
    if (stages == 3) PGV_FRAG_GO(false, 3);
    else if (stages == 4) PGV_FRAG_GO(false, 4);
    else PGV_FRAG_GO(false, 5);
  }
#undef PGV_FRAG_GO
  return 1;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pgv_dbg_set_gemm_variant(int v) {
  g_frag_variant = v;
  return 0;
}

// Returns 1 when the product was launched here, 0 when the shape is not covered (pgv_gemm then takes the LDS-tiled path),
// < 0 on error.  C[m][n] = sum_k A(m,k) B(k,n) (+ bias[n]); ``flags``: PGV_PREZEROED = C holds zeros (split-K adds into it).
// ``init_c``: the caller's routine that fills C with the bias (or zeros) when a split-K product finds C uninitialised.
int pgv_gemm_frag(int M, int N, int K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                  float* C, int64_t ldc, const float* bias_n, int flags, hipStream_t st,
                  int (*init_c)(float*, int, int, int64_t, const float*, hipStream_t)) {
  // PGV_COMPUTE_BF16 (round 6): the same jobs with 32-deep chunks and one bfloat16 matrix instruction per tile pair and chunk -
  // built, tested (tests/test_gpu_kernels.py: every covered shape), and NOT the default: bit 12 of the tuning variant sends the
  // covered bf16 shapes here.  Measured on the z = 512 products of BASELINE config 2 against gemm.hip's LDS tiles (us, same
  // box, operands cold): decoder forward [256 x 12 288, K = 512] 27.9 / 37.0, decoder input gradient [256 x 512, K = 12 288]
  // 25.8 / 38.6; encoder forward 52.3 / 45.3 (47.3 with 64 x 32 macro tiles), encoder input gradient 44.5 / 44.5, weight
  // gradients 63.4 / 34.1 and 27.9 / 20.6 - 32 x 32 .. 64 x 32 macro tiles per wave re-read the operands 8 - 32 x out of L2
  // (800 MB for the encoder forward), which a 128 x 128 workgroup tile does not.  INSIDE the step, where the activations are
  // still in L2 / MALL, the two winners are 43.5 / 39.9 us under the profiler (LDS tiles: 45.3 / 45.3) and the whole step of
  // config 2 is no faster: 2.852 - 2.879 ms with them against 2.841 - 2.861 without (bench.py --gemm-variant 0 / 1024 at the
  // time, alternating on one box); all six here: 2.97 - 2.99.
  const bool bf16 = (flags & PGV_COMPUTE_BF16) != 0;
  const int CD = bf16 ? 32 : 16;
  if (g_frag_variant & 1024) return 0;   // (A/B timing: the LDS-tiled kernels of gemm.hip)
  if (bf16 && !(g_frag_variant & 4096)) return 0;
  if (K % CD != 0 || K < 2 * CD || ldc % 4 != 0 || !al16(A) || !al16(B) || !al16(C) || (bias_n && !al16(bias_n))) return 0;
  const bool p_kc = sbk == 1, p_ic = sbn == 1 && !p_kc, q_kc = sak == 1, q_ic = sam == 1 && !q_kc;
  if (!(p_kc || p_ic) || !(q_kc || q_ic)) return 0;
  FragArgs a;
  a.dbg = 0;
  a.P = B, a.Q = A, a.C = C, a.bias = bias_n;
  a.ldp = p_kc ? sbn : sbk, a.ldq = q_kc ? sam : sak, a.ldc = ldc;
  if (a.ldp % 4 != 0 || a.ldq % 4 != 0) return 0;
  a.nchunks = K / CD;
  a.KS = 1, a.atomic = 0, a.wgred = 0;
  // byte offsets of a macro tile's rows must fit the 32-bit lane offsets
  auto fits = [](int64_t rows, int64_t ld) { return rows * ld * 4 < ((int64_t)1 << 31); };
  const int64_t work = (int64_t)M * N;   // output elements
  const bool long_k = K >= 4096 && work <= 512 * 1024;
  auto finish = [&](int pext, int qext, int job_target = 1024, bool wg_reduce = false) {
    a.MPw = N / pext, a.MQw = M / qext;
    if (long_k) {
      // split K until ~job_target wave jobs exist (one or two per SIMD), at least 8 chunks per job
      int ks = (int)pgv_cdiv(job_target << ((g_frag_variant >> 4) & 3), (int64_t)a.MPw * a.MQw);
      ks = (int)max((int64_t)1, min((int64_t)ks, (int64_t)a.nchunks / (bf16 ? 4 : 8)));
      // (only where a job's tile is 64 x 32: input-gradient-shaped products 22.3 -> 16.1 us (z = 64) and 43.4 -> 38.5 (z = 512)
      // on cold operands; the 32 x 32 jobs of the long-K forward product lost 2 - 7 us to the two barriers.  Bit 13 of the
      // tuning variant: every wave on its own)
      if (ks >= 4 && wg_reduce && !(g_frag_variant & 8192)) ks &= ~3, a.wgred = 1;
      a.KS = ks;
      if (ks > 1) a.atomic = (flags & PGV_PREZEROED) ? 2 : 1;
    }
    a.njobs = a.MPw * a.MQw * a.KS;
  };
  int rc = 0;
  if (p_kc && q_kc) {
    // forward products: x[m][k] . W[n][k]
    if (N % 32 != 0 || M % 32 != 0 || !fits(32, a.ldp) || !fits(32, a.ldq)) return 0;
    // (same-box A/B, us: encoder forward [256 x 128, K = 25 024] 22.8 here with two jobs per SIMD against 27.5 for the
    // LDS-tiled kernel; the short-K decoder forward [256 x 25 024, K = 64] 18.0 against 16.4: that one stays there)
    if (!long_k && !(bf16 && (g_frag_variant & 4096)) && !(g_frag_variant & 2048)) return 0;
    if (!long_k && N % 64 == 0 && (int64_t)(N / 64) * (M / 32) >= 1536) {
      finish(64, 32);
      rc = bf16 ? launch_frag<1, 1, 4, 2, false, true>(a, st)
                : ((g_frag_variant & 512) ? launch_frag<0, 0, 4, 2, false>(a, st) : launch_frag<1, 1, 4, 2, false>(a, st));
    } else {
      finish(32, 32, 2048);
      if (a.atomic == 1 && init_c(C, M, N, ldc, bias_n, st)) return PGV_E_LAUNCH;
      rc = bf16 ? launch_frag<1, 1, 2, 2, true, true>(a, st)
                : ((g_frag_variant & 512) ? launch_frag<0, 0, 2, 2, true>(a, st) : launch_frag<1, 1, 2, 2, true>(a, st));
    }
  } else if (p_ic && q_kc) {
    // input-gradient products: gy[m][k] . W[k][n]
    if (N % 64 != 0 || M % 32 != 0 || !fits(32, a.ldq)) return 0;
    finish(64, 32, 1024, true);
    if (a.atomic == 1 && init_c(C, M, N, ldc, bias_n, st)) return PGV_E_LAUNCH;
    rc = bf16 ? launch_frag<4, 1, 4, 2, true, true>(a, st)
              : ((g_frag_variant & 512) ? launch_frag<4, 0, 4, 2, true>(a, st) : launch_frag<4, 1, 4, 2, true>(a, st));
  } else if (p_ic && q_ic) {
    // weight-gradient products: gy[b][m]^T . x[b][n]
    if (N % 64 != 0 || long_k) return 0;
    // (32 Q indices per job, read 2 at a time: twice the jobs of a 64 x 64 tiling - 782 instead of 391 for the decoder's
    // Linear, whose 25 024 rows otherwise fill 38 % of the chip: 14.7 against 21.5 us)
    if (M % 32 != 0) return 0;
    finish(64, 32);
    rc = bf16 ? launch_frag<4, 2, 4, 2, false, true>(a, st) : launch_frag<4, 2, 4, 2, false>(a, st);
  } else
    return 0;
  return rc;
}
