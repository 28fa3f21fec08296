// Second-generation implicit-GEMM kernels for the stride-2 k=4 layers at the reference sizes (model/encoder.py:241-255,
// model/decoder.py:205-218): ONE 512-thread workgroup per CU with fixed wave roles - waves 0-3 (one per SIMD) multiply,
// waves 4-7 (their SIMD partners) stage - instead of two co-resident workgroups that take turns on the matrix pipe
// (conv_band.hip).  DESIGN.md section 3.4 has the measurements behind every choice.
//
//   * waves split M (output channels) first: every MFMA wave multiplies ALL pixel tiles of a unit against its own
//     channel slice, so a ragged pixel count costs < 1 tile in 25 instead of whole idle waves;
//   * the weights never touch LDS: a lane's weight operand of k-step (c, kh) is ONE dword of the weight tensor, loaded
//     straight from global memory (L2-resident) into a two-halves register ring, half an item ahead of its use;
//   * the input band is double-buffered in LDS by channel chunks: while chunk i is multiplied, the loader waves commit
//     chunk i+1 from registers (producer's BatchNorm affine applied, zero padding stored as zeros) and the global loads
//     of chunk i+2 are in flight in a second register set - one workgroup barrier per chunk;
//   * every k-step is one scheduling region in which the MFMAs are pinned 1 : 1 with the ds_reads two steps ahead
//     (sched_group_barrier): 34.5 clk per MFMA against 53.7 for the compiler's own order;
//   * D^T orientation (pixels = M rows, channels = N columns): a lane's accumulator is 4 consecutive pixels of ONE
//     channel, so the epilogue stores 16 bytes of NCHW per lane without a transpose, and BatchNorm statistics /
//     backward projections are two registers per tile column;
//   * the weight-gradient kernel (conv_wgrad_ws_kernel) uses a leaner stage (StageLean: buffer loads with the hardware
//     range check, 2-4 instructions per 16-byte slot): an instruction of a loader wave gets an issue slot only every
//     ~70 clocks while its SIMD partner streams MFMAs.
#include "conv_tile.h"
#include "band_prefetch.h"

#ifdef PGV_V2_TIMING
// per-wave accumulated phase durations in shader cycles (s_memtime): slot i = time before V2_ACC(i) since the previous
// stamp, summed over the items of a persistent workgroup; slot 7 = number of items (scratch/v2_timing.py)
__device__ unsigned long long* pgv_tlog_v2 = nullptr;
extern "C" int pgv_dbg_set_tlog_v2(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(pgv_tlog_v2), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define V2_T0() unsigned long long v2_tp = clock64(), v2_sum[7] = {0, 0, 0, 0, 0, 0, 0}, v2_n = 0
#define V2_ACC(i)                                 \
  do {                                            \
    const unsigned long long now = clock64();     \
    v2_sum[i] += now - v2_tp;                     \
    v2_tp = now;                                  \
  } while (0)
#define V2_ITEM() (++v2_n)
#define V2_FLUSH()                                                                               \
  do {                                                                                           \
    if ((threadIdx.x & 63) == 0 && pgv_tlog_v2) {                                                \
      unsigned long long* o = pgv_tlog_v2 + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;  \
      for (int i = 0; i < 7; ++i) o[i] = v2_sum[i];                                              \
      o[7] = v2_n;                                                                               \
    }                                                                                            \
  } while (0)
#else
#define V2_T0()
#define V2_ACC(i)
#define V2_ITEM()
#define V2_FLUSH()
#endif

#ifndef PGV_V2_PRIO_MFMA
#define PGV_V2_PRIO_MFMA 0
#define PGV_V2_PRIO_LOADER 2
#endif
#ifndef PGV_V2_LOADER_SLEEP
#define PGV_V2_LOADER_SLEEP 3  // x 64 clocks
#endif

namespace {

#define PGV_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 4x4 transpose inside every aligned lane quad: in: lane i holds x[r] = element (row r, column i); out: lane i holds
// x[k] = element (row i, column k).
__device__ __forceinline__ void quad_transpose(float (&x)[4], int lane) {
  const bool odd = lane & 1, hi = lane & 2;
  {  // distance 1: (x0,x1) and (x2,x3)
    const float s0 = odd ? x[0] : x[1], s1 = odd ? x[2] : x[3];
    const float r0 = dpp_mov<0xB1>(s0), r1 = dpp_mov<0xB1>(s1);
    if (odd) {
      x[0] = r0;
      x[2] = r1;
    } else {
      x[1] = r0;
      x[3] = r1;
    }
  }
  {  // distance 2: (x0,x2) and (x1,x3)
    const float s0 = hi ? x[0] : x[2], s1 = hi ? x[1] : x[3];
    const float r0 = dpp_mov<0x4E>(s0), r1 = dpp_mov<0x4E>(s1);
    if (hi) {
      x[0] = r0;
      x[1] = r1;
    } else {
      x[2] = r0;
      x[3] = r1;
    }
  }
}

// sum over the 4 lanes that share (lane & 15)
__device__ __forceinline__ float lanegroup_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// Staging of one channel chunk of a band: CK channels x ROWS rows at LDS row stride WP (image columns 0..W-1, then >= 2
// zero pad columns which double as the left padding of the next row), channels back to back.  The tile is a flat list of
// 16-byte chunks; lane tid owns chunks tid + 256*j (slot j).  Slots are issued (global -> register) and committed
// (register -> LDS, producer's BatchNorm affine on image data, exact zeros elsewhere) by the loader waves; the per-lane
// slot geometry is computed once, the data registers exist twice (two items in flight).
// ---------------------------------------------------------------------------------------------------------------
template <int CK, int ROWS, int W, int WP, int H, int MINPAD = 2>
struct StageV2 {
  static constexpr int QR = WP / 4, PC = ROWS * QR, ITEMS = CK * PC, NPF = (ITEMS + 255) / 256, NP = W % 4;
  static_assert(WP % 4 == 0 && WP >= W + MINPAD && NPF <= 32 && ROWS < 256 && CK <= 256 && QR < 4096, "stage geometry");
  // per-lane constants of the slots (one copy, shared by the register sets)
  struct Geo {
    unsigned meta[NPF];  // rr | ncol << 8 | c << 12   (ncol = 0: pad chunk or idle lane)
    int off0[NPF];       // byte offset of the slot's window inside the chunk's planes for a band that starts at row 0
    __device__ __forceinline__ void init(int tid) {
#pragma unroll
      for (int j = 0; j < NPF; ++j) {
        const int e = tid + 256 * j;
        const int ee = min(e, ITEMS - 1);
        const int rowi = ee / QR, q = ee - rowi * QR;
        const int c = rowi / ROWS, rr = rowi - c * ROWS;
        const int nc = e < ITEMS ? min(max(W - 4 * q, 0), 4) : 0;
        meta[j] = (unsigned)rr | ((unsigned)nc << 8) | ((unsigned)c << 12);
        // the partial chunk at the end of a row reads the LAST four floats of the row (rotated into place at commit)
        const int col = 4 * q - ((NP != 0 && nc > 0 && nc < 4) ? 4 - NP : 0);
        off0[j] = ((c * H + rr) * W + col) * 4;
      }
    }
  };
  // one item in flight
  struct Set {
    f32x4 v[NPF];
    unsigned live;  // bit j: slot j holds image data
  };
  // plane0 = first element of the first channel of the chunk in its sample; chunks without image data read offset 0
  template <int J>
  static __device__ __forceinline__ void issue_slot(const Geo& g, Set& s, const float* __restrict__ plane0, int ih0) {
    const int rr = g.meta[J] & 255;
    const bool ok = (unsigned)(ih0 + rr) < (unsigned)H && (g.meta[J] & 0xF00u) != 0;
    const unsigned off = ok ? (unsigned)(g.off0[J] + ih0 * (W * 4)) : 0u;
    s.live = ok ? (s.live | (1u << J)) : (s.live & ~(1u << J));
    // The load is inline asm ON PURPOSE: with two register sets in flight across the loop back-edge the compiler's
    // s_waitcnt bookkeeping drains BOTH sets at every commit (vmcnt(0) at the loop header), which collapses the
    // prefetch to less than one item.  The asm load is invisible to that bookkeeping; the loader waits by hand
    // (wait_set) - loads of a wave complete in issue order and this wave issues nothing else on the vector memory path.
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(s.v[J]) : "v"(off), "s"(plane0) : "memory");
  }
  // Block until the OLDER of the two sets in flight has landed (the NPF loads of the newer one may stay outstanding).
  static __device__ __forceinline__ void wait_set() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPF) : "memory");
    __builtin_amdgcn_sched_barrier(0);  // nothing that reads the registers may be scheduled above the wait
  }
  // aff = LDS table ([C] scales, [C] shifts) or null; c0 = first channel of the chunk
  // per-slot affine of the chunk that starts at channel c0, from the LDS table ([C] scales, [C] shifts): all reads
  // issued back to back (one LDS latency per chunk, not one per slot)
  static __device__ __forceinline__ void load_affine(const Geo& g, const float* __restrict__ aff, int C, int c0,
                                                     float (&sc)[NPF], float (&sh)[NPF]) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int cg = c0 + (int)((g.meta[j] >> 12) & 255);
      sc[j] = aff[cg];
      sh[j] = aff[C + cg];
    }
  }
  template <int J>
  static __device__ __forceinline__ void commit_slot(const Geo& g, const Set& s, float* __restrict__ tile, int tid,
                                                     bool has_aff, float scj, float shj) {
    if (256 * (J + 1) <= ITEMS || tid + 256 * J < ITEMS) {
      const f32x4 t = s.v[J];
      const bool on = (s.live >> J) & 1u;
      const float m = on ? (has_aff ? scj : 1.f) : 0.f, a = (on && has_aff) ? shj : 0.f;
      f32x4 x;
      if (NP == 0) {
        x.x = fmaf(t.x, m, a);
        x.y = fmaf(t.y, m, a);
        x.z = fmaf(t.z, m, a);
        x.w = fmaf(t.w, m, a);
      } else {
        const bool part = ((g.meta[J] >> 8) & 15) < 4;
        const float e0 = part ? t[(4 - NP) & 3] : t.x;
        const float e1 = part ? t[(5 - NP) & 3] : t.y;
        const float e2 = part ? t[(6 - NP) & 3] : t.z;
        const float m1 = (part && NP < 2) ? 0.f : m, a1 = (part && NP < 2) ? 0.f : a;
        const float m2 = (part && NP < 3) ? 0.f : m, a2 = (part && NP < 3) ? 0.f : a;
        const float m3 = part ? 0.f : m, a3 = part ? 0.f : a;
        x.x = fmaf(e0, m, a);
        x.y = fmaf(e1, m1, a1);
        x.z = fmaf(e2, m2, a2);
        x.w = fmaf(t.w, m3, a3);
      }
      *reinterpret_cast<f32x4*>(tile + 4 * tid + 1024 * J) = x;
    }
  }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel symbol (not a stream operation: also fine while a
// graph is being captured, but there is no point in repeating it on every launch)
inline int raise_lds_once(const void* kern, const char* who) {
  static const void* done[64];
  static int n_done = 0;
  for (int i = 0; i < n_done; ++i)
    if (done[i] == kern) return PGV_OK;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
  if (e != hipSuccess) {
    pgv_set_error("%s: cannot raise the dynamic LDS limit: %s", who, hipGetErrorString(e));
    return PGV_E_LAUNCH;
  }
  if (n_done < 64) done[n_done++] = kern;
  return PGV_OK;
}

// Workgroup barrier of the wave-specialised kernels: LDS traffic of this wave complete, then s_barrier.  Unlike
// __syncthreads() it carries no fence, so the compiler does not drain the global loads that are in flight across it
// (the loader's prefetch, the MFMA waves' weight loads); the "memory" clobber keeps LDS accesses on their side.
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// compile-time loop helper: f(integral_constant<int, I>) for I in [0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Loader stage of the wgrad kernel.  Measured on gfx950 (scratch/ubench/ws_share.hip): while one wave of a SIMD streams
// MFMAs back to back, every instruction of the SIMD's other wave (VALU or LDS, any s_setprio) gets an issue slot only
// about every 70 clocks.  The StageV2 loader (20 instructions per 16-byte slot) then needs 1.3x the time the MFMA waves
// need for the item.  This stage spends 2-4 instructions per slot instead:
//  * buffer_load_dwordx4 with the hardware range check: pad lanes carry offset 0xFFFFFFFF and read as zero - no select,
//    no live mask; the per-item address lives in the resource descriptor (scalar ALU), the lane offset is a constant;
//  * no rotation of the partial chunk at the end of a row: the chunk is loaded as it lies (the floats behind the row
//    end are the next row's) and the CONSUMER zeroes the operand lanes that would read them (last k-step of a row);
//  * the per-channel affine comes pre-masked per slot (pad chunks 0,0); rows outside the image exist only in the first
//    and the last band of a sample: those items take a slow path that masks offset and shift per slot.

// ZTAIL: the loader itself clears the floats behind the end of a row in the row's last chunk (W % 4 != 0), for consumers
// that read them as zero padding in every k-step (conv_down): 1 + (4 - W % 4) more instructions per slot.
// PSTRIDE != 0: the channel planes sit PSTRIDE floats apart in LDS instead of back to back (bank spreading for
// consumers that read the same pixel of many channels at once); the slots then carry their LDS address.
template <int CK, int ROWS, int W, int WP, int H, bool ZTAIL = false, int PSTRIDE = 0>
struct StageLean {
  static constexpr int QR = WP / 4, PC = ROWS * QR, ITEMS = CK * PC, NPF = (ITEMS + 255) / 256, NP = W % 4;
  static_assert(WP % 4 == 0 && WP >= W && NPF <= 32, "stage geometry");
  struct Geo {
    unsigned voff[NPF];  // byte offset from the band's first row in channel 0, 0xFFFFFFFF for pad chunks / idle lanes
    f32x2 ma[NPF];       // (scale, shift) of the slot's channel; (0, 0) for pad chunks
    unsigned top_bad, bot_bad;  // bit j: slot j lies in a row outside the image in the first / the last band of a sample
    unsigned whole;             // bit j: all 4 floats of slot j lie inside their row (ZTAIL)
    int laddr[PSTRIDE ? NPF : 1];  // LDS float index of the slot (PSTRIDE != 0)
    // top_rows: rows of the first band above the image; bot_row: first row of the last band below the image
    // after_slot(integral_constant<j>) runs when slot j's constants are ready (the caller issues the first item's load
    // of the slot there: the rest of the set-up then overlaps the memory latency)
    template <class F>
    __device__ __forceinline__ void init(int tid, const float* __restrict__ aff, int C, bool has_aff, int top_rows,
                                         int bot_row, F&& after_slot) {
      top_bad = bot_bad = 0;
      whole = 0;
      static_for<0, NPF>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int e = min(tid + 256 * j, ITEMS - 1);
        const int rowi = e / QR, q = e - rowi * QR;
        const int c = rowi / ROWS, rr = rowi - c * ROWS;
        const bool data = tid + 256 * j < ITEMS && 4 * q < W;
        whole |= 4 * q + 4 <= W ? 1u << j : 0u;
        if (PSTRIDE) laddr[PSTRIDE ? j : 0] = c * PSTRIDE + rr * WP + 4 * q;
        voff[j] = data ? (unsigned)(((c * H + rr) * W + 4 * q) * 4) : 0xFFFFFFFFu;
        top_bad |= rr < top_rows ? 1u << j : 0u;
        bot_bad |= rr >= bot_row ? 1u << j : 0u;
        after_slot(jc);
        ma[j] = f32x2{data ? (has_aff ? aff[c] : 1.f) : 0.f, (data && has_aff) ? aff[C + c] : 0.f};
      });
    }
  };
  struct Set {
    f32x4 v[NPF];
  };
  // descriptor of "everything from the band's first row (row ih0 of channel 0 of sample b) to the end of the tensor"
  static __device__ __forceinline__ i32x4 band_rsrc(const float* __restrict__ base, int64_t total_bytes, int64_t elem0) {
    const uint64_t p = (uint64_t)base + (uint64_t)(elem0 * 4);
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)p);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(p >> 32) & 0xFFFF);
    r.z = __builtin_amdgcn_readfirstlane((int)(uint32_t)(total_bytes - elem0 * 4));
    r.w = 0x00020000;  // raw buffer, 32-bit data format (gfx9 family)
    return r;
  }
  // bad = top_bad / bot_bad of the item (EDGE) - unused otherwise
  template <int J, bool EDGE>
  static __device__ __forceinline__ void issue_slot(const Geo& g, Set& s, i32x4 rsrc, unsigned bad) {
    unsigned off = g.voff[J];
    if (EDGE) off |= (unsigned)__builtin_amdgcn_sbfe(bad, J, 1);
    // inline asm: see StageV2::issue_slot (the compiler's s_waitcnt bookkeeping would drain both sets in flight)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(s.v[J]) : "v"(off), "s"(rsrc) : "memory");
  }
  template <int J, bool EDGE, bool AFF>
  static __device__ __forceinline__ void commit_slot(const Geo& g, const Set& s, float* __restrict__ tile, int tid,
                                                     unsigned bad) {
    if (256 * (J + 1) <= ITEMS || tid + 256 * J < ITEMS) {
      f32x4 x = s.v[J];
      if (AFF) {
        f32x2 ma = g.ma[J];
        if (EDGE) ma.y = ((bad >> J) & 1u) ? 0.f : ma.y;  // the data of such a row was read as zero already
        // x = x * scale + shift on both halves, scale / shift broadcast out of the (scale, shift) pair by op_sel
        f32x2 lo = {x.x, x.y}, hi = {x.z, x.w};
        asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(lo) : "v"(lo), "v"(ma));
        asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(hi) : "v"(hi), "v"(ma));
        x = f32x4{lo.x, lo.y, hi.x, hi.y};
      }
      if (ZTAIL && NP != 0) {
        const int km = __builtin_amdgcn_sbfe((int)g.whole, J, 1);  // -1: keep, 0: the row ends inside this chunk
#pragma unroll
        for (int c = NP; c < 4; ++c) x[c] = __int_as_float(__float_as_int(x[c]) & km);
      }
      *reinterpret_cast<f32x4*>(PSTRIDE ? tile + g.laddr[PSTRIDE ? J : 0] : tile + 4 * tid + 1024 * J) = x;
    }
  }
};

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
template <int CB, int CS, int W, int H, int R, int MW, int CK, bool FUSE, bool HAS_AFF, int ACT, bool STG = false>
__global__ __launch_bounds__(512, 2) void conv_down_ws_kernel(int B, const float* __restrict__ big,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            int act, float slope, float* __restrict__ out,
                                                            double* __restrict__ stats, pgv_bn_fuse fuse) {
  using G = DownV2Cfg<CB, CS, W, H, R, MW, CK>;
  constexpr int Ws = G::Ws, Hs = G::Hs, BANDS = G::BANDS, NW = G::NW, MTW = G::MTW, P = G::P, NT = G::NT;
  constexpr int WP = G::WP, PLANE = G::PLANE, NCH = G::NCH, S = G::S, BUF = G::BUF;
  using Stage = StageV2<CK, G::ROWS, W, WP, H>;
  constexpr int NPF = Stage::NPF;
  static_assert(!STG || (NCH == 1 && !FUSE && ACT != 2 && Ws % 4 == 0 && S >= MTW * NT), "deferred stores");
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
    aff[i] = in_scale ? in_scale[i] : 1.f;
    aff[CB + i] = in_shift ? in_shift[i] : 0.f;
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
      static_for<0, NL>([&](auto j) { Lean::template commit_slot<decltype(j)::value, true, HAS_AFF>(geo, sx, dst, ltid, bad); });
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
    geo.init(ltid);
    sA.live = sB.live = 0;
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
    issue_all(sA, 0);
    issue_all(sB, 1);
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
  auto wload = [&](int m, int elem) { return *reinterpret_cast<const float*>(wb + (size_t)elem * 4 + wl[m]); };
  const pgv_act_params actp = pgv_act_setup(act, slope);
  // D^T = X^T W^T: the accumulator of a lane is 4 consecutive pixels (rows (lane>>4)*4 + reg) of one channel (lane & 15)
  const int ech = lane & 15, epx = 4 * (lane >> 4);
  float bias_r[MTW], mean_r[MTW], rstd_r[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int cl = (wm * MTW + m) * 16 + ech;
    bias_r[m] = bias ? bias[cl] : 0.f;
    mean_r[m] = FUSE ? fuse.mean[cl] : 0.f;
    rstd_r[m] = FUSE ? fuse.rstd[cl] : 0.f;
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
      if constexpr (STG)  // [this band of channel 0 of the sample .. end of the tensor)
        prs = StageLean<CK, G::ROWS, W, WP, H, true>::band_rsrc(out, (int64_t)B * CS * (Hs * Ws) * 4,
                                                                  ((int64_t)b * CS * Hs + oh0) * Ws);
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int cl = (wm * MTW + m) * 16 + ech;
        float* orow = out + ((int64_t)b * CS + cl) * (Hs * Ws) + (int64_t)oh0 * Ws;
        const float* arow = FUSE ? fuse.a + ((int64_t)b * CS + cl) * (Hs * Ws) + (int64_t)oh0 * Ws : nullptr;
        if constexpr (FUSE) {
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
              f4u o;
              o.x = x[0], o.y = x[1], o.z = x[2], o.w = x[3];
              *reinterpret_cast<f4u*>(orow + p0) = o;
              st_s[m] += (x[0] + x[1]) + (x[2] + x[3]);
              if constexpr (FUSE) {
                st_q[m] = fmaf(x[0], (av[g].x - mean_r[m]) * rstd_r[m], st_q[m]);
                st_q[m] = fmaf(x[1], (av[g].y - mean_r[m]) * rstd_r[m], st_q[m]);
                st_q[m] = fmaf(x[2], (av[g].z - mean_r[m]) * rstd_r[m], st_q[m]);
                st_q[m] = fmaf(x[3], (av[g].w - mean_r[m]) * rstd_r[m], st_q[m]);
              } else {
                st_q[m] = fmaf(x[0], x[0], st_q[m]);
                st_q[m] = fmaf(x[1], x[1], st_q[m]);
                st_q[m] = fmaf(x[2], x[2], st_q[m]);
                st_q[m] = fmaf(x[3], x[3], st_q[m]);
              }
            } else {  // ragged last tile of the band
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (p0 + k < Pb) {
                  orow[p0 + k] = x[k];
                  st_s[m] += x[k];
                  if constexpr (FUSE)
                    st_q[m] = fmaf(x[k], (arow[p0 + k] - mean_r[m]) * rstd_r[m], st_q[m]);
                  else
                    st_q[m] = fmaf(x[k], x[k], st_q[m]);
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
          f4u av[TG];
          if constexpr (FUSE) {
#pragma unroll
            for (int g = 0; g < TG; ++g)
              if (t0 + g < NT && pl + (t0 + g) * 16 + 4 <= Pb) av[g] = *reinterpret_cast<const f4u*>(arow + pl + (t0 + g) * 16);
          }
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
            if constexpr (STG) {
              pend[m][t] = f32x4{y0.x, y0.y, y1.x, y1.y};
              if (m == 0) p4[t] = p0 + 4 <= Pb ? (unsigned)p0 * 4u : OOR;
            }
            if (p0 + 4 <= Pb) {
              f4u o;
              o.x = y0.x, o.y = y0.y, o.z = y1.x, o.w = y1.y;
              if constexpr (!STG) *reinterpret_cast<f4u*>(orow + p0) = o;
              ss += y0 + y1;
              if constexpr (FUSE) {
                const f32x2 mu2 = {mean_r[m], mean_r[m]}, rs2 = {rstd_r[m], rstd_r[m]};
                qq = __builtin_elementwise_fma(y0, (f32x2{av[g].x, av[g].y} - mu2) * rs2, qq);
                qq = __builtin_elementwise_fma(y1, (f32x2{av[g].z, av[g].w} - mu2) * rs2, qq);
              } else {
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
                  if constexpr (FUSE)
                    qq.x = fmaf(x[e], (arow[p0 + e] - mean_r[m]) * rstd_r[m], qq.x);
                  else
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
  double* dst = FUSE ? fuse.red : stats;
  if (dst) {
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
                   const pgv_bn_fuse* fuse, hipStream_t st) {
  using G = DownV2Cfg<CB, CS, W, H, R, MW, CK>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  if (d->Cb != CB || d->Cs != CS) return 0;
  if (stats && fuse) return 0;  // one reduction slot
  // the two ways the train step calls it: forward of a Conv2D block (producer's BatchNorm folded or not, LeakyReLU,
  // statistics) and input gradient of a TConv2D block (plain product, optional BatchNorm-backward projections)
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, const float*, int, float, float*,
                         double*, pgv_bn_fuse);
  kern_t kern;
  const bool leaky = act == PGV_ACT_LEAKY_RELU && slope >= 0.f && slope <= 1.f;
  const int actk = act == PGV_ACT_NONE ? 0 : (leaky ? 1 : 2);
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
  if constexpr (W == 174 && G::NCH == 1) {
    if (!fuse && actk != 2) {
      if (in_scale)
        kern = actk == 1 ? (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, false, true, 1, true>
                         : (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, false, true, 0, true>;
      else
        kern = actk == 1 ? (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, false, false, 1, true>
                         : (kern_t)conv_down_ws_kernel<CB, CS, W, H, R, MW, CK, false, false, 0, true>;
    }
  }
  if (int rc = raise_lds_once((const void*)kern, "conv_down_v2")) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_v2: memset failed");
    return PGV_E_LAUNCH;
  }
  const int units = d->B * G::BANDS;
  const int grid = min(units, 256);
  const pgv_bn_fuse fz = {nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), bytes, st, d->B, big, in_scale, in_shift, w, bias, act,
                     slope, out, stats, fuse ? *fuse : fz);
  PGV_CHECK_LAUNCH("conv_down_v2");
  return 1;
}

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
template <int CB, int CS, int W, int H, int R, int MW, int CK, bool FUSE, bool HAS_AFF, int ACT, bool STG = false>
__global__ __launch_bounds__(512, 2) void conv_up_ws_kernel(int B, const float* __restrict__ small_in,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          int act, float slope, float* __restrict__ out,
                                                          double* __restrict__ stats, pgv_bn_fuse fuse) {
  using G = UpV2Cfg<CB, CS, W, H, R, MW, CK>;
  constexpr int Ws = G::Ws, Hs = G::Hs, Wg = G::Wg, Hg = G::Hg, Wgp = G::Wgp, BANDS = G::BANDS, NW = G::NW;
  constexpr int MTW = G::MTW, P = G::P, NT = G::NT, WsP = G::WsP, PLANE = G::PLANE, NCH = G::NCH, S = G::S, BUF = G::BUF;
  using Stage = StageV2<CK, G::ROWS, Ws, WsP, Hs, 1>;
  constexpr int NPF = Stage::NPF;
  static_assert(!STG || (NCH == 1 && !FUSE && ACT != 2 && W % 2 == 0 && S >= MTW * NT), "deferred stores");
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
    aff[i] = in_scale ? in_scale[i] : 1.f;
    aff[CS + i] = in_shift ? in_shift[i] : 0.f;
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
      static_for<0, NL>([&](auto j) { Lean::template commit_slot<decltype(j)::value, true, HAS_AFF>(geo, sx, dst, ltid, bad); });
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
    geo.init(ltid);
    sA.live = sB.live = 0;
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
        Stage::template commit_slot<J>(geo, sx, dst, ltid, HAS_AFF, sc[J], sh[J]);
      });
    };
    issue_all(sA, 0);
    issue_all(sB, 1);
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
  auto wload = [&](int m, int cs) { return *reinterpret_cast<const float*>(wb + (size_t)cs * (CB * 16 * 4) + wl[m]); };
  const pgv_act_params actp = pgv_act_setup(act, slope);
  // accumulator layout: column (lane&15) = grid position, rows (lane>>4)*4 + reg = (channel lane>>4 of the M tile,
  // phase reg = ph*2 + pw); after the lane-pair exchange a lane holds 4 consecutive pixels of output row 2u + (lane&1)
  const int ech = lane >> 4, odd = lane & 1;
  float bias_r[MTW], mean_r[MTW], rstd_r[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int cl = (wm * MTW + m) * 4 + ech;
    bias_r[m] = bias ? bias[cl] : 0.f;
    mean_r[m] = FUSE ? fuse.mean[cl] : 0.f;
    rstd_r[m] = FUSE ? fuse.rstd[cl] : 0.f;
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
  f32x4 pend[STG ? MTW : 1][STG ? NT : 1];
  unsigned p4[STG ? NT : 1], p2[STG ? NT : 1], choff[STG ? MTW : 1];
  i32x4 prs = {0, 0, 0, 0x00020000};  // zero bytes: nothing pending yet, every store is dropped
  if constexpr (STG) {
#pragma unroll
    for (int t = 0; t < NT; ++t) p4[t] = p2[t] = OOR;
#pragma unroll
    for (int m = 0; m < MTW; ++m) choff[m] = (unsigned)(((wm * MTW + m) * 4 + (lane >> 4)) * (H * W) * 4);
  }
  auto store_pending = [&](auto qc) {  // tile q = m * NT + t of the pending unit
    constexpr int q = decltype(qc)::value, m = q / NT, t = q - m * NT;
    const unsigned o4 = p4[t] + choff[m], o2 = p2[t] + choff[m];
    const f32x2 lo = {pend[m][t].x, pend[m][t].y};
    const f32x4 all = pend[m][t];
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
    V2_ACC(3);
    static_for<0, S>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      constexpr int sn = st + 2;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (STG && st < MTW * NT) {
        store_pending(st_c);  // one tile of the previous unit leaves per k-step
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
      if (!FUSE && ACT != 2) {
        const int* tof = tofl + (band == BANDS - 1 ? NT * 256 : 0);
        // No wave-uniform per-tile branches and no address arithmetic (a uniform branch per tile costs more than the
        // tile's arithmetic: the general path below spends ~480 clocks per tile): the geometry of the two kinds of band
        // comes from the tables computed at kernel start.  Lanes whose 4 pixels exist store 16 bytes; the lane at a row end of an odd-width image
        // stores its 1-3 pixels one by one; statistics ride in the same exec-masked blocks.
        const f32x2 slope2 = {slope, slope};
        if constexpr (STG)  // descriptor of [this band of channel 0 of the sample .. end of the tensor)
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
            if constexpr (STG) {
              if (m == 0) {
                p4[t] = nv == 4 ? (unsigned)off * 4u : OOR;
                p2[t] = nv == 2 ? (unsigned)off * 4u : OOR;
              }
            }
            if (nv == 4) {
              f4u o;
              o.x = o01.x, o.y = o01.y, o.z = o23.x, o.w = o23.y;
              if constexpr (STG) {
                pend[m][t] = f32x4{o.x, o.y, o.z, o.w};
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
                  if constexpr (STG)
                    pend[m][t] = f32x4{ov[0], ov[1], ov[2], ov[3]};  // (even width: 2 valid pixels, stored as 8 bytes)
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
            const float o0 = odd ? r0 : x[0], o1 = odd ? r1 : x[1], o2 = odd ? x[2] : r0, o3 = odd ? x[3] : r1;
            if (fulls[g]) {  // (wave-uniform) every lane stores 4 valid pixels
              const int off = offs[g];
              f4u o;
              o.x = o0, o.y = o1, o.z = o2, o.w = o3;
#ifndef PGV_V2_NO_STORE
              *reinterpret_cast<f4u*>(obase + off) = o;
#endif
              st_s[m] += (o0 + o1) + (o2 + o3);
              if constexpr (FUSE) {
                st_q[m] = fmaf(o0, (av[g].x - mean_r[m]) * rstd_r[m], st_q[m]);
                st_q[m] = fmaf(o1, (av[g].y - mean_r[m]) * rstd_r[m], st_q[m]);
                st_q[m] = fmaf(o2, (av[g].z - mean_r[m]) * rstd_r[m], st_q[m]);
                st_q[m] = fmaf(o3, (av[g].w - mean_r[m]) * rstd_r[m], st_q[m]);
              } else {
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
                  obase[off + k] = ov[k];
                  st_s[m] += ov[k];
                  if constexpr (FUSE)
                    st_q[m] = fmaf(ov[k], (abase[off + k] - mean_r[m]) * rstd_r[m], st_q[m]);
                  else
                    st_q[m] = fmaf(ov[k], ov[k], st_q[m]);
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
  if constexpr (STG) static_for<0, MTW * NT>([&](auto qc) { store_pending(qc); });  // the last unit
  V2_FLUSH();
  // statistics / projections: one float64 atomic per channel per workgroup (see conv_down_ws_kernel)
  double* dst = FUSE ? fuse.red : stats;
  if (dst) {
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
                 const pgv_bn_fuse* fuse, hipStream_t st) {
  using G = UpV2Cfg<CB, CS, W, H, R, MW, CK>;
  // deferred stores for the layer whose output bursts bound it (129x174: 56 KB per unit), where the variant exists
  constexpr bool STG = W == 174 && G::NCH == 1;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  if (d->Cb != CB || d->Cs != CS) return 0;
  if (stats && fuse) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, const float*, int, float, float*,
                         double*, pgv_bn_fuse);
  kern_t kern;
  const bool leaky = act == PGV_ACT_LEAKY_RELU && slope >= 0.f && slope <= 1.f;
  const int actk = act == PGV_ACT_NONE ? 0 : (leaky ? 1 : 2);
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
  if constexpr (STG) {  // (only the non-fused LeakyReLU / linear forms exist with deferred stores)
    if (fuse || actk == 2) return 0;
    if (in_scale)
      kern = actk == 1 ? (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, false, true, 1, true>
                       : (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, false, true, 0, true>;
    else
      kern = actk == 1 ? (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, false, false, 1, true>
                       : (kern_t)conv_up_ws_kernel<CB, CS, W, H, R, MW, CK, false, false, 0, true>;
  }
  if (int rc = raise_lds_once((const void*)kern, "conv_up_v2")) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_v2: memset failed");
    return PGV_E_LAUNCH;
  }
  const int units = d->B * G::BANDS;
  const int grid = min(units, 256);
  const pgv_bn_fuse fz = {nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), bytes, st, d->B, small_in, in_scale, in_shift, w, bias, act, slope,
                     out, stats, fuse ? *fuse : fz);
  PGV_CHECK_LAUNCH("conv_up_v2");
  return 1;
}

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
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nparts, int n4,
                                                           float* __restrict__ gw, int accumulate) {
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

// ---------------------------------------------------------------------------------------------------------------
// WGRAD of the 1 <-> 8 channel 5x5 layers (enc1 / dec8: big = [B,1,257,347], small = [B,8,129,174]), same structure as
// conv_wgrad_ws_kernel:  gw[cs][kh][kw] = sum_{b,oh,ow} small[b,cs,oh,ow] * big[b,0,2oh-2+kh,2ow-2+kw].
// M = cs (8 of the tile's 16 rows; lanes 8-15 duplicate 0-7 and are not stored), N = 25 taps in two tiles (lanes past
// tap 24 duplicate it), K = output pixels.  There are only two (M, N) tiles, so the four MFMA waves split K: wave w
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

template <int CS, int W, int H, int R, bool AFF_S>
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
      static_for<0, NPB>([&](auto j) { StageB::template commit_slot<decltype(j)::value, false, false>(geoB, bx, dst, ltid, 0u); });
      if (AFF_S && band == BANDS - 1) {
        static_for<0, NPS>([&](auto j) {
          StageS::template commit_slot<decltype(j)::value, true, AFF_S>(geoS, sx, dst + PLANE_B, ltid, geoS.bot_bad);
        });
      } else {
        static_for<0, NPS>([&](auto j) {
          StageS::template commit_slot<decltype(j)::value, false, AFF_S>(geoS, sx, dst + PLANE_B, ltid, 0u);
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
  // wave w multiplies output row w of the unit.  A: lane (m = cs = lane & 7, k = pixel lane>>4) reads
  // small[cs][w][4i + k]; B: lane (tap = 16 n + (lane & 15), k) reads big[2w + kh][2(4i + k) + kw - 2]
  const int offA = PLANE_B + (lane & 7) * PLANE_S + wave * WsP + (lane >> 4);
  int offB[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int tap = min(16 * n + (lane & 15), G::NTAP - 1);
    offB[n] = (2 * wave + tap / G::KS) * WP + tap % G::KS - 2 + 2 * (lane >> 4);
  }
  // operand lanes of the last k-step of a row that lie behind the end of the row (the loader does not clean them)
  const bool keepA = 4 * (SPR - 1) + (lane >> 4) < Ws;
  bool keepB[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) keepB[n] = 8 * (SPR - 1) + 2 * (lane >> 4) + min(16 * n + (lane & 15), G::NTAP - 1) % G::KS - 2 < W;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
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
        float av[3][2], bv[3][2][2];
        auto load_pair = [&](int ps, float (&a)[2], float (&b)[2][2]) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            a[h] = cur[offA + 4 * (2 * ps + h)];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[h][n] = cur[offB[n] + 8 * (2 * ps + h)];
          }
          if (2 * ps + 1 == SPR - 1) {
            if (Ws % 4 != 0) a[1] = keepA ? a[1] : 0.f;
            if (W % 4 != 0)
#pragma unroll
              for (int n = 0; n < 2; ++n) b[1][n] = keepB[n] ? b[1][n] : 0.f;
          }
        };
        load_pair(0, av[0], bv[0]);
        load_pair(1, av[1], bv[1]);
        static_for<0, PPR>([&](auto pc) {
          constexpr int ps = decltype(pc)::value, pn = ps + 2;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (pn < PPR) load_pair(pn, av[pn % 3], bv[pn % 3]);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[n] = PGV_MFMA4(av[ps % 3][h], bv[ps % 3][h][n], acc[n]);
          // 3 operand reads (one per operand: the two k-steps of a pair come from one ds_read2) under 4 MFMAs
          static_for<0, 4>([&](auto kc) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (pn < PPR && decltype(kc)::value < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          });
        });
        __builtin_amdgcn_sched_barrier(0);
      }
      ws_barrier();
    }
  }
  // ---- this wave's partial gradient: D column = lane & 15 = tap - 16 n, rows (lane>>4)*4 + reg = cs (0..7 are real)
  float* pw = partial + ((size_t)blockIdx.x * 4 + wave) * (CS * G::NTAP);
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int cs = (lane >> 4) * 4 + reg, tap = 16 * n + (lane & 15);
      if (cs < CS && tap < G::NTAP) pw[cs * G::NTAP + tap] = acc[n][reg];
    }
}

template <int R>
int launch_wgrad5_v2(const pgv_conv_desc* d, const float* big, const float* small_in, const float* small_scale,
                     const float* small_shift, float* gw, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  using G = Wgrad5Cfg<8, 347, 257, R>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds, "LDS budget");
  const int units = d->B * G::BANDS;
  const int grid = min(units, 256);
  const int nparts = 4 * grid;
  const int64_t need = (int64_t)nparts * 8 * G::NTAP * sizeof(float);
  if (!workspace || workspace_bytes < need || ((uintptr_t)gw & 15) || ((uintptr_t)workspace & 15)) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, float*);
  kern_t kern = small_scale ? (kern_t)conv_wgrad5_ws_kernel<8, 347, 257, R, true>
                            : (kern_t)conv_wgrad5_ws_kernel<8, 347, 257, R, false>;
  if (int rc = raise_lds_once((const void*)kern, "conv_wgrad5_v2")) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), bytes, st, d->B, big, small_in, small_scale, small_shift, (float*)workspace);
  PGV_CHECK_LAUNCH("conv_wgrad5_v2");
  const int n4 = 8 * G::NTAP / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n4 + 7) / 8), dim3(256), 0, st, (const float*)workspace, nparts, n4, gw,
                     (d->flags & PGV_PREZEROED) ? 1 : 0);
  PGV_CHECK_LAUNCH("conv_wgrad5_v2 reduce");
  return 1;
}

template <int CB, int CS, int W, int H, int R>
int launch_wgrad_v2(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                    const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                    void* workspace, int64_t workspace_bytes, hipStream_t st) {
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
  // PGV_PREZEROED: gw holds zeros or an earlier partial sum to add to; otherwise it is overwritten
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n4 + 7) / 8), dim3(256), 0, st, (const float*)workspace, grid, n4, gw,
                     (d->flags & PGV_PREZEROED) ? 1 : 0);
  PGV_CHECK_LAUNCH("conv_wgrad_v2 reduce");
  return 1;
}

}  // namespace

// Returns 1 when handled, 0 when the shape / mode is not covered (the caller falls back to conv_band.hip), < 0 on error.
// Fused BatchNorm-backward projections (pgv_bn_fuse) in the wave-specialised kernels: the saved-activation loads sit in
// the epilogue burst next to the stores and are not overlapped with matrix work (one workgroup per CU), so the fused
// form only pays where the alternative is worse (measured inside the train step, us, fused v2 / fused band / unfused v2
// + separate reduce pass):  down 129x174: 115 / 102 / 130 -> band;  down 65x88: 85 / 83 / 107 -> v2 fused;
// up 65x88: 154 / 89 / 138 -> band;  up 33x45: 133 / (no fused band kernel: 140) / 125 -> v2 unfused + reduce pass.
int pgv_conv_down_v2(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* small_out, double* stats,
                     const pgv_bn_fuse* fuse, hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4) return 0;
  if (d->flags & PGV_COMPUTE_BF16) return 0;
  if (fuse && d->Hb == 129 && d->Wb == 174) return 0;
  if (d->Hb == 33 && d->Wb == 45)   // 32 -> 64 channels, 17x23 outputs: the whole sample per unit, M split 4 ways
    return launch_down_v2<32, 64, 45, 33, 17, 4, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, st);
  if (d->Hb == 65 && d->Wb == 88)   // 16 -> 32 channels, 33x45 outputs: 3 bands of 11 rows, waves 2 (M) x 2 (pixels)
    return launch_down_v2<16, 32, 88, 65, 11, 2, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, st);
  if (d->Hb == 129 && d->Wb == 174)  // 8 -> 16 channels, 65x88 outputs: 13 bands of 5 rows, waves split the pixels
    return launch_down_v2<8, 16, 174, 129, 5, 1, 8>(d, big, in_scale, in_shift, w, bias, act, slope, small_out, stats, fuse, st);
  return 0;
}

// (returns 2 when it handled the call but left the requested projections to a separate reduce pass)
int pgv_conv_up_v2(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                   const float* w, const float* bias, int act, float slope, float* big_out, double* stats,
                   const pgv_bn_fuse* fuse, hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4) return 0;
  if (d->flags & PGV_COMPUTE_BF16) return 0;
  if (fuse && d->Hb == 65 && d->Wb == 88) return 0;
  if (fuse && d->Hb == 33 && d->Wb == 45) {
    const int rc = launch_up_v2<32, 64, 45, 33, 9, 4, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, nullptr, st);
    return rc == 1 ? 2 : rc;
  }
  if (d->Hb == 33 && d->Wb == 45)   // 64 -> 32 channels onto 33x45: 2 bands of 9 / 8 grid rows, M split 4 ways
    return launch_up_v2<32, 64, 45, 33, 9, 4, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, st);
  if (d->Hb == 65 && d->Wb == 88)   // 32 -> 16 channels onto 65x88: 3 bands of 11 grid rows, waves split the positions
    return launch_up_v2<16, 32, 88, 65, 11, 1, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, st);
#ifndef PGV_V2_NO_UP_L2
  // 16 -> 8 channels onto 129x174 (13 bands of 5 grid rows, waves split the positions).  This layer is bound by the CU's
  // store path (56 KB of output per unit): with direct stores from the epilogue the matrix pipe idled 35 % of the time
  // and the band kernel's two co-resident workgroups were faster; with the output staged through LDS and moved out by
  // the loader waves during the next unit's k-steps (STG) this form wins.  Fused projections stay on the band kernel.
  if (d->Hb == 129 && d->Wb == 174)
    return launch_up_v2<8, 16, 174, 129, 5, 1, 16>(d, small_in, in_scale, in_shift, w, bias, act, slope, big_out, stats, fuse, st);
#endif
  return 0;
}

// workspace: one partial gradient per workgroup
int64_t pgv_conv_wgrad_v2_workspace(const pgv_conv_desc* d) {
  if (d->stride == 2 && d->pad == 2 && d->kh == 5 && d->kw == 5 && !(d->flags & PGV_COMPUTE_BF16) && d->Cb == 1 &&
      d->Cs == 8 && d->Hb == 257 && d->Wb == 347)
    return (int64_t)4 * 256 * 8 * 25 * sizeof(float);  // one partial gradient per MFMA wave
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4 || (d->flags & PGV_COMPUTE_BF16)) return 0;
  if ((d->Hb == 33 && d->Wb == 45 && d->Cb == 32 && d->Cs == 64) ||
      (d->Hb == 65 && d->Wb == 88 && d->Cb == 16 && d->Cs == 32) ||
      (d->Hb == 129 && d->Wb == 174 && d->Cb == 8 && d->Cs == 16))
    return (int64_t)256 * d->Cs * d->Cb * 16 * sizeof(float);
  return 0;
}

int pgv_conv_wgrad_v2(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                      const float* small_in, const float* small_scale, const float* small_shift, float* gw,
                      void* workspace, int64_t workspace_bytes, hipStream_t st) {
  if (d->stride == 2 && d->pad == 2 && d->kh == 5 && d->kw == 5 && !(d->flags & PGV_COMPUTE_BF16) && d->B > 0 &&
      d->Cb == 1 && d->Cs == 8 && d->Hb == 257 && d->Wb == 347 && !big_scale)
    return launch_wgrad5_v2<4>(d, big, small_in, small_scale, small_shift, gw, workspace, workspace_bytes, st);
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4) return 0;
  if ((d->flags & PGV_COMPUTE_BF16) || d->B <= 0) return 0;
  if (d->Hb == 33 && d->Wb == 45)
    return launch_wgrad_v2<32, 64, 45, 33, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw,
                                              workspace, workspace_bytes, st);
  if (d->Hb == 65 && d->Wb == 88)
    return launch_wgrad_v2<16, 32, 88, 65, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw,
                                              workspace, workspace_bytes, st);
  if (d->Hb == 129 && d->Wb == 174)
    return launch_wgrad_v2<8, 16, 174, 129, 3>(d, big, big_scale, big_shift, small_in, small_scale, small_shift, gw,
                                               workspace, workspace_bytes, st);
  return 0;
}
