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
#pragma once
#include "conv_tile.h"
#include "band_prefetch.h"

#ifdef PGV_V2_TIMING
// per-wave accumulated phase durations in shader cycles (s_memtime): slot i = time before V2_ACC(i) since the previous
// stamp, summed over the items of a persistent workgroup; slot 7 = number of items (scratch/v2_timing.py)
// (one log pointer and one setter per translation unit: pgv_dbg_set_tlog_v2_down / _up / _wgrad)
static __device__ unsigned long long* pgv_tlog_v2 = nullptr;
#define PGV_V2_CAT2(a, b) a##b
#define PGV_V2_CAT(a, b) PGV_V2_CAT2(a, b)
extern "C" int PGV_V2_CAT(pgv_dbg_set_tlog_v2_, PGV_V2_TU)(void* p) {
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
    // after_slot(integral_constant<j>) runs when slot j's constants are ready: the caller issues the first item's load of
    // the slot there, so the rest of the set-up (two integer divisions per slot, up to 32 slots) overlaps its memory
    // latency instead of preceding it (as in StageLean::Geo::init)
    template <class F>
    __device__ __forceinline__ void init(int tid, F&& after_slot) {
      static_for_geo<0>(tid, after_slot);
    }
    template <int j, class F>
    __device__ __forceinline__ void static_for_geo(int tid, F&& after_slot) {
      if constexpr (j < NPF) {
        const int e = tid + 256 * j;
        const int ee = min(e, ITEMS - 1);
        const int rowi = ee / QR, q = ee - rowi * QR;
        const int c = rowi / ROWS, rr = rowi - c * ROWS;
        const int nc = e < ITEMS ? min(max(W - 4 * q, 0), 4) : 0;
        meta[j] = (unsigned)rr | ((unsigned)nc << 8) | ((unsigned)c << 12);
        // the partial chunk at the end of a row reads the LAST four floats of the row (rotated into place at commit)
        const int col = 4 * q - ((NP != 0 && nc > 0 && nc < 4) ? 4 - NP : 0);
        off0[j] = ((c * H + rr) * W + col) * 4;
        after_slot(std::integral_constant<int, j>{});
        static_for_geo<j + 1>(tid, after_slot);
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
  // bf16: the committed operand is rounded to bfloat16 (PGV_COMPUTE_BF16: products of rounded operands on the fp32 MFMA)
  template <int J>
  static __device__ __forceinline__ void commit_slot(const Geo& g, const Set& s, float* __restrict__ tile, int tid,
                                                     bool has_aff, float scj, float shj, bool bf16 = false) {
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
      if (bf16) x = f32x4{round_bf16(x.x), round_bf16(x.y), round_bf16(x.z), round_bf16(x.w)};
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

// 16-byte buffer load with the hardware range check, invisible to the compiler's s_waitcnt bookkeeping (the caller waits
// by hand before it reads dst); a free function because an asm output operand cannot name a lambda capture
__device__ __forceinline__ void v2_buffer_load_x4(f32x4& dst, unsigned byte_off, i32x4 rsrc) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(byte_off), "s"(rsrc) : "memory");
}

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
  // BF16 (PGV_COMPUTE_BF16): the committed operand is rounded to bfloat16, pairs through v_cvt_pk_bf16_f32 (3 instructions
  // per two floats)
  template <int J, bool EDGE, bool AFF, bool BF16 = false>
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
      if (BF16) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 lo = {(__bf16)x.x, (__bf16)x.y}, hi = {(__bf16)x.z, (__bf16)x.w};
        const unsigned ul = __builtin_bit_cast(unsigned, lo), uh = __builtin_bit_cast(unsigned, hi);
        x = f32x4{__uint_as_float(ul << 16), __uint_as_float(ul & 0xFFFF0000u), __uint_as_float(uh << 16),
                  __uint_as_float(uh & 0xFFFF0000u)};
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

}  // namespace
