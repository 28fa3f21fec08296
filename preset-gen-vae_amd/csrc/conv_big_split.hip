// PGV_COMPUTE_F32_SPLIT kernels of the LARGE-plane k4 s2 p2 layers - the layers the 4-layer headline stack runs
// (model/encoder.py:243-248, model/decoder.py:212-217: 8 <-> 16 channels on 129x174, 16 <-> 32 on 65x88, 32 <-> 64 on 33x45):
// forward products and input gradients with every fp32 product as SIX bf16 matrix instructions on exact three-way splits
// of both operands (conv_deep_split.hip has the arithmetic and its error measurements), with the fused epilogues of the
// train step (BatchNorm finalize in the prologue, BatchNorm statistics, pgv_bwd_fuse incl. class sums and bias-gradient
// copies).
//
// Structure: one 512-thread workgroup per CU, persistent over its units (sample, band of R output rows / UB grid rows).
// Per unit all eight waves go through TWO PHASES, separated by workgroup barriers:
//   matrix phase  - request the next unit's input band (global -> registers) and this unit's saved activations (fused
//                   form), then the wave's share of the products: v_mfma_f32_16x16x32_bf16 x 6 per fragment pair, the B
//                   fragments read from the LDS image PD steps ahead, the accumulators written to the output tile in LDS;
//   vector phase  - split the next unit's band into its three bf16 plane images (the stage is free: everybody has left the
//                   matrix phase), then move this unit's tile out (bias / activation / statistics or the fused backward).
// The phases are NOT overlapped on purpose.  A first version ran two teams of four waves in opposite phases (one matrix
// wave and one vector wave per SIMD at any time): v_mfma_f32_16x16x32_bf16 holds a SIMD's vector issue for 8 of its 16
// cycles, the vector wave got one instruction through per ~10 clocks and the matrix wave ran at half rate beside it - both
// phases took 3 - 4 k clocks per unit where 1.2 k of matrix time was the floor (in-kernel stamps, scratch/bigq_stamps.py).
// With the phases in sequence the vector work issues from two waves per SIMD at full rate and the matrix pipe is shared by
// two waves that hide each other's LDS latency.
//   * weights: a wave's A fragments for its (M tiles, K range), all three planes, live in REGISTERS for the whole kernel
//     (24 - 96 VGPRs), read once from a split shadow in fragment order (shadow_bigq_*_item, conv_deep_common.h);
//   * activations: three plane images.  Convolution: channel-PAIR planar image [pair][row][plane][column], a dword = the
//     bf16 pair (channels 2p, 2p + 1) of one pixel: a loader item (two channels x 4 consecutive pixels, 16-byte loads) is
//     ONE 16-byte LDS store per plane - conflict free, where pixel-major images gave 16- to 32-way conflicts on 4-byte
//     stores - and the B fragment of output pixel ow is the 4-dword window of input columns 2 ow - 2 .. 2 ow + 1 (the four
//     kernel columns) at an 8-byte aligned address.  Transposed convolution: pixel-major image (8 channels = 16 bytes,
//     swizzled), a loader item = one pixel x 8 channels (coalesced 4-byte loads along the row) = one 16-byte store per plane;
//   * outputs: a [channel][pixels] tile in LDS per K group of waves (waves that split K write a tile each; LDS float atomics
//     into one tile cost 3.5 k clocks per unit), added up in a fixed order and moved out as contiguous 16-byte runs.
#include <type_traits>
#include "conv_tile.h"
#include "conv_deep_common.h"

#ifdef PGV_BIGQ_STAMPS   // scratch builds only (scratch/bigq_stamps.py): clock64() at the phase marks of workgroup 0
static unsigned long long* g_bigq_stamps = nullptr;
extern "C" void pgv_dbg_set_bigq_stamps(void* p) { g_bigq_stamps = (unsigned long long*)p; }
#define QSTAMP_ARG , unsigned long long* __restrict__ stamps
#define QSTAMP_PASS , g_bigq_stamps
#define QSTAMP(j, k)                                                                                                    \
  do {                                                                                                                  \
    if (stamps && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && (j) < 60) stamps[((wave >> 2) * 64 + (j) + 2) * 16 + (k)] = clock64(); \
  } while (0)
#else
#define QSTAMP_ARG
#define QSTAMP_PASS
#define QSTAMP(j, k) \
  do {               \
  } while (0)
#endif

namespace {

typedef unsigned short u16;

__device__ __forceinline__ f32x4 six_products(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {
  c = mfma_bf16_k32(a[0], b[2], c);   // smallest terms first
  c = mfma_bf16_k32(a[2], b[0], c);
  c = mfma_bf16_k32(a[1], b[1], c);
  c = mfma_bf16_k32(a[0], b[1], c);
  c = mfma_bf16_k32(a[1], b[0], c);
  return mfma_bf16_k32(a[0], b[0], c);
}

// the six terms in the order they are accumulated (smallest first): (plane of a, plane of b)
__device__ constexpr int kTermA[6] = {0, 2, 1, 0, 1, 0}, kTermB[6] = {2, 0, 1, 1, 0, 0};

// 16-byte global load through a buffer descriptor of the whole tensor: a quad at the ragged end of a row simply reads on
// into the next row (masked where it is used), and one that crosses the end of the tensor reads zeros there (the range
// check is per dword) - no divergent scalar-load branch, whose merge with the 16-byte path made the compiler wait for the
// loads where they were issued (2 - 5 k clocks per unit)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tensor_rsrc(const float* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f4u buffer_load_x4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(f4u, v);
}

// workgroup barrier between phases: this wave's LDS traffic complete, then s_barrier - without the vector-memory wait of
// __syncthreads() (which would expose the latency of every unit's global loads and stores)
__device__ __forceinline__ void ws_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------
// DOWN: small = conv_{s2,p2,k4}(big').  GEMM: M = small channels, K = (kernel row, 8 big channels) x 4 kernel columns,
// N = the band's output pixels.  A wave owns MW M tiles (every B fragment it reads feeds 6 MW instructions), KW K steps and
// TMAX pixel tiles; PD = how many steps ahead it requests fragments.
// NPL = operand planes: 3 = fp32 products as six instructions on exact three-way splits (PGV_COMPUTE_F32_SPLIT), 1 = bf16
// operand mode (PGV_COMPUTE_BF16: one bf16 plane per operand, rounded to nearest, one instruction per fragment pair).
template <int CB_, int CS_, int H_, int W_, int R_, int MW_, int KSPLIT_, int NSPLIT_, int PD_ = 1, int NPL_ = 3>
struct DownQ {
  static constexpr int CB = CB_, CS = CS_, H = H_, W = W_, R = R_, MW = MW_, KSPLIT = KSPLIT_, NSPLIT = NSPLIT_, PD = PD_;
  static constexpr int NPL = NPL_, NTERM = NPL_ == 3 ? 6 : 1;
  static_assert(NPL_ == 3 || NPL_ == 1, "three planes (six product terms) or one");
  static constexpr int XR = 2 * R + 2;
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, BANDS = (Hs + R - 1) / R;
  static constexpr int NCP = CB / 2, NG = CB / 8;                     // channel pairs; K groups of 8 channels (4 pairs)
  // image: dword (pair, row, plane, column), column = input column + 4; rows of WPD dwords, pair planes CPS dwords apart with
  // CPS = 32 (mod 64): the two 16-lane halves of a read group (pairs kq, kq + 1) then sit on disjoint banks
  static constexpr int WPD = (2 * Ws + 4 + 3) / 4 * 4;
  static constexpr int CPS = ((XR * NPL * WPD - 32 + 63) / 64) * 64 + 32;
  static constexpr int STAGE = NCP * CPS * 4;                         // bytes
  static constexpr int KSTEPS = 4 * NG, KHW = 4 / KSPLIT, KW = KHW * NG;   // K steps; kernel rows / K steps of a wave
  static constexpr int MTN = CS / 16, MG = MTN / MW;                  // M tiles; M groups over the waves
  static constexpr int NPX = R * Ws, NT = (NPX + 15) / 16, TMAX = (NT + NSPLIT - 1) / NSPLIT;
  static constexpr int OSTR = (NPX + 3) / 4 * 4, O_SLICE = CS * OSTR, O_FLOATS = KSPLIT * O_SLICE;
  static constexpr int QX = (W + 3) / 4, ITEMS = NCP * XR * QX, QB = (ITEMS + 511) / 512;   // loader items
  static constexpr int LPC = 512 / CS, QO = ((NPX + 3) / 4 + LPC - 1) / LPC;
  static constexpr size_t LDS_BYTES = (size_t)STAGE + (size_t)O_FLOATS * 4 + sizeof(float) * (2 * CB + 8);
  static_assert(MG * KSPLIT * NSPLIT == 8 && MTN % MW == 0 && 4 % KSPLIT == 0, "eight waves");
  static_assert(CPS >= XR * NPL * WPD && WPD >= 4 * QX + 4 && W >= 4 && LPC >= 1 && LPC <= 64 && R % 2 == 0, "tile shapes");
  static_assert(LDS_BYTES <= 160 * 1024 && ((NG - 1) * 4 * CPS + 3 * NPL * WPD + (NPL - 1) * WPD) * 4 + 16 < 65536, "LDS budget / immediate offsets");
};

template <class G, bool FUSE>
__global__ __launch_bounds__(512) void down_q_kernel(int B, const float* __restrict__ big, const float* __restrict__ in_scale,
                                                     const float* __restrict__ in_shift, const u32x4* __restrict__ wsh,
                                                     const float* __restrict__ bias, int act, float slope,
                                                     float* __restrict__ out, double* __restrict__ stats, int stat_stride,
                                                     pgv_bn_src in_bn, pgv_bwd_fuse fuse QSTAMP_ARG) {
  constexpr int CB = G::CB, CS = G::CS, H = G::H, W = G::W, Hs = G::Hs, Ws = G::Ws, TMAX = G::TMAX, R = G::R, NPX = G::NPX;
  constexpr int NG = G::NG, WPD = G::WPD, CPS = G::CPS, OSTR = G::OSTR, LPC = G::LPC, QO = G::QO, MW = G::MW;
  constexpr int NPL = G::NPL, NTERM = G::NTERM;
  typedef unsigned u4a8 __attribute__((ext_vector_type(4), aligned(8)));   // 16-byte LDS load from an 8-byte aligned address
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* lds_x = ldsb;
  float* otile = reinterpret_cast<float*>(ldsb + G::STAGE);
  float* aff = otile + G::O_FLOATS;   // [2*CB]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mtl = (wave % G::MG) * MW, kg = (wave / G::MG) % G::KSPLIT, ng = wave / (G::MG * G::KSPLIT);
  const pgv_split_sel sel = pgv_split_sel_make();

  for (int i = tid; i < CB; i += 512) {
    float sc = 1.f, sh = 0.f;
    if (in_bn.stats)
      pgv_bn_finalize_dev(in_bn, CB, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[CB + i] = sh;
  }
  // ---- this wave's weight fragments: M tiles mtl .. mtl + MW - 1, K steps [kg * KW, + KW), three planes
  u32x4 af[MW][G::KW][NPL];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw) {
    const u32x4* a_src = wsh + ((size_t)((mtl + mw) * G::KSTEPS + kg * G::KW) * NPL) * 64 + lane;
#pragma unroll
    for (int k = 0; k < G::KW; ++k)
#pragma unroll
      for (int p = 0; p < NPL; ++p) af[mw][k][p] = a_src[(k * NPL + p) * 64];
  }
  // ---- this wave's pixel tiles ng, ng + NSPLIT, ...: byte offset of the lane's window (pair kq, the wave's first kernel
  // row, plane 0, image column 2 ow + 2): 16 bytes at an 8-byte aligned address (the compiler reads them as ds_read2_b64;
  // a ds_read_b128 at such an address takes 64 LDS clocks instead of 8, scratch/ubench/lds_read_forms.hip)
  int blo[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const int n = min((ng + G::NSPLIT * t) * 16 + m, NPX - 1), ohl = n / Ws, ow = n - ohl * Ws;
    blo[t] = (kq * CPS + (2 * ohl + kg * G::KHW) * NPL * WPD + 2 * ow + 2) * 4;
  }
  // ---- loader items: (channel pair, band row, quad of 4 columns)
  int l_src[G::QB], l_cr[G::QB], l_row[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 512 * i, G::ITEMS - 1);
    const int cp = q / (G::XR * G::QX), rem = q - cp * (G::XR * G::QX), r = rem / G::QX, qi = rem - r * G::QX;
    l_src[i] = (2 * cp) * (H * W) + 4 * qi;                       // + sample * CB * H * W + image row * W
    l_row[i] = r * W;                                             // (the item's row within the band, in floats)
    l_cr[i] = (cp << 8) | r | ((tid + 512 * i < G::ITEMS) ? 0x8000 : 0) | ((4 * qi + 4 > W) ? 0x4000 : 0);   // (0x4000: ragged row end)
  }
  // ---- move-out role: LPC lanes per channel
  const int och = tid / LPC, part = tid % LPC;
  const pgv_act_params ap = pgv_act_setup(act, slope);
  const float bv = (!FUSE && bias) ? bias[och] : 0.f;
  float ka = 1.f, kb = 0.f, kc = 0.f;
  pgv_actd_params actd = pgv_actd_setup(PGV_ACT_NONE, 0.f);
  if (FUSE) {
    ka = fuse.coef[och], kb = fuse.coef[CS + och], kc = fuse.coef[2 * CS + och];
    actd = pgv_actd_setup(fuse.act, fuse.slope);
  }
  // forward: s[0] / s[1] = sum / sum of squares; fused: s[k] = sum of g_y in (row, column) parity class k (bands start at
  // even rows: a tile element has the same class in every unit - kept per slot, sorted at the end)
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  float slot[QO][4];
#pragma unroll
  for (int i = 0; i < QO; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) slot[i][e] = 0.f;

  const int grid = (int)gridDim.x, u0 = pgv_xcd_block(), units = B * G::BANDS;
  const int J = u0 < units ? (units - 1 - u0) / grid + 1 : 0;   // this workgroup's units u0, u0 + grid, ...

  // The band loads are software-pipelined ITEM BY ITEM through the vector phase: an item of unit j + 1 is converted and
  // committed, and its registers are re-issued at once for the same item of unit j + 2, so loads are in flight through the
  // rest of the vector phase and the whole next matrix phase.  (Issued in one burst at the start of the matrix phase, a
  // unit's 56 - 78 KB were still arriving when the vector phase wanted them: a CU's fair share of HBM moves them in ~6 k
  // clocks, a matrix phase lasts 3 - 4 k.)  A load for a unit beyond the workgroup's last gets an offset outside the buffer:
  // it returns zeros without traffic, and no load sits under a branch the compiler's counter model would have to merge.
  f4u rb[G::QB][2];
  const __amdgpu_buffer_rsrc_t big_rs = tensor_rsrc(big, (int64_t)B * CB * (H * W) * 4);
  struct UnitPos { unsigned sample, kill; int ih0; };
  auto unit_pos = [&](int j) {
    const int u = min(u0 + j * grid, units - 1), b = u / G::BANDS, band = u - b * G::BANDS;
    return UnitPos{(unsigned)b * (unsigned)(CB * H * W), j < J ? 0u : 0x80000000u, 2 * band * R - 2};
  };
  auto issue_item = [&](int i, const UnitPos& up) {
    // (row offset = scalar band part + the item's constant: no vector multiply; a row outside the image reads outside the
    // buffer like a unit beyond the last one; a quad at the ragged end of a row reads on into the next row - masked at the
    // commit)
    const unsigned rk = (unsigned)(up.ih0 + (l_cr[i] & 255)) < (unsigned)H ? up.kill : 0x80000000u;
    const unsigned o = (up.sample + (unsigned)(l_src[i] + l_row[i] + up.ih0 * W)) * 4u;
    rb[i][0] = buffer_load_x4(big_rs, o | rk);
    rb[i][1] = buffer_load_x4(big_rs, (o + (unsigned)(H * W * 4)) | rk);
  };
  // (the affine of an item's channel pair is read from LDS for ALL items before the first one is committed: read inside
  // the item, the wait for it - lgkmcnt counts in order - was also a wait for the previous item's three image stores)
  auto commit_item = [&](int i, const UnitPos& up, const float (&af2)[4]) {
    const int cp = (l_cr[i] >> 8) & 63;
    // (image position: pair cp, band row, plane 0, column 4 qi + 4 - 4 qi recovered from the global offset)
    const int dst = (cp * (CPS - 2 * H * W) + (l_cr[i] & 255) * (NPL * WPD) + l_src[i] + 4) * 4;
    // (a row outside the image arrived as zeros: only the SHIFT has to vanish there - the padding stays zero under an
    // affine too.  FUSE = an input-gradient call: no affine at all, the launcher refuses one)
    const bool in = (unsigned)(up.ih0 + (l_cr[i] & 255)) < (unsigned)H;
    const float s0 = af2[0], s1c = af2[1], h0 = in ? af2[2] : 0.f, h1 = in ? af2[3] : 0.f;
    if (l_cr[i] & 0x8000) {   // (kept under the lane mask for full slots too: without it the 129x174 / 65x88 kernels lost 1 - 2 us)
      u32x4 ph, pm, pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool on = W % 4 == 0 || !(l_cr[i] & 0x4000) || e < W % 4;   // (a quad at the ragged end of a row)
        const float x0 = rb[i][0][e], x1 = rb[i][1][e];
        const float y0 = on ? (FUSE ? x0 : fmaf(x0, s0, h0)) : 0.f, y1 = on ? (FUSE ? x1 : fmaf(x1, s1c, h1)) : 0.f;
        unsigned a1, a2 = 0, a3 = 0;
        if constexpr (NPL == 3)
          pgv_split3_pair(y0, y1, a1, a2, a3, sel);
        else
          a1 = pgv_pack_bf16x2(y0, y1);   // (bf16 operand mode: rounded to nearest, one plane)
        ph[e] = a1, pm[e] = a2, pl[e] = a3;
      }
      *reinterpret_cast<u32x4*>(lds_x + dst) = ph;
      if constexpr (NPL == 3) {
        *reinterpret_cast<u32x4*>(lds_x + dst + WPD * 4) = pm;
        *reinterpret_cast<u32x4*>(lds_x + dst + 2 * WPD * 4) = pl;
      }
    }
  };
  auto vector_items = [&](int jc) {   // commit unit jc from the registers, re-issue them for unit jc + 1
    const UnitPos uc = unit_pos(jc), un = unit_pos(jc + 1);
    float af2[G::QB][4];
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int c = 2 * ((l_cr[i] >> 8) & 63);
      if (!FUSE) af2[i][0] = aff[c], af2[i][1] = aff[c + 1], af2[i][2] = aff[CB + c], af2[i][3] = aff[CB + c + 1];
    }
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      commit_item(i, uc, af2[i]);
      if (i < 3) QSTAMP(jc - 1, 10 + i);
      issue_item(i, un);
    }
  };
  // the saved activation of the fused epilogue: requested in the matrix phase, used in the vector phase
  f4u av[QO];
  float av_t = 0.f;
  auto tile_geom = [&](int j, int& nfl, size_t& goff) {
    const int u = u0 + j * grid, b = u / G::BANDS, band = u - b * G::BANDS, oh0 = band * R;
    nfl = min(R, Hs - oh0) * Ws;
    goff = ((size_t)b * CS + och) * G::P + oh0 * Ws;
  };
  // (buffer loads, a lane without an element reads outside the buffer: every lane issues them, so the compiler's counter
  // model counts them exactly and the waits for the band loads issued before them do not wait for these)
  const __amdgpu_buffer_rsrc_t a_rs = tensor_rsrc(FUSE ? fuse.a : nullptr, FUSE ? (int64_t)B * CS * G::P * 4 : 0);
  const __amdgpu_buffer_rsrc_t out_rs = tensor_rsrc(out, (int64_t)B * CS * G::P * 4);
  auto fetch_a = [&](int j) {
    int nfl;
    size_t goff;
    tile_geom(j, nfl, goff);
    const unsigned a_o = (unsigned)goff * 4u;
#pragma unroll
    for (int i = 0; i < QO; ++i) {
      const int q4 = part + LPC * i;
      av[i] = buffer_load_x4(a_rs, (a_o + 16u * q4) | (4 * q4 + 4 <= nfl ? 0u : 0x80000000u));
    }
    const int tail0 = nfl & ~3;
    av_t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, (int)((a_o + 4u * (tail0 + part)) | (part < nfl - tail0 ? 0u : 0x80000000u)), 0, 0));
  };

  if (J > 0) {
    const UnitPos p0 = unit_pos(0);
#pragma unroll
    for (int i = 0; i < G::QB; ++i) issue_item(i, p0);
  }
  // (the image is cleared behind the first unit's loads: their latency covers it)
  for (int i = tid; i < (int)((G::STAGE + G::O_FLOATS * 4) / 16); i += 512) reinterpret_cast<u32x4*>(ldsb)[i] = u32x4{0, 0, 0, 0};
  __syncthreads();   // image zeroed, affine staged
  if (J > 0) vector_items(0);
  __syncthreads();
  // Everything the prologue loaded is READ here: the compiler may sink those loads (read-only data) below the barriers, and
  // with any of them pending at the loop header its counter model waits for "them" inside the loop - s_waitcnt vmcnt(n)
  // with n counting down through the matrix loop, i.e. for the band loads just issued, and vmcnt(0) in front of the next
  // issue, i.e. for the stores of the vector phase (1.6 - 2.9 k clocks per unit).
#pragma unroll
  for (int mw = 0; mw < MW; ++mw)
#pragma unroll
    for (int k = 0; k < G::KW; ++k)
#pragma unroll
      for (int p = 0; p < NPL; ++p) asm volatile("" ::"v"(af[mw][k][p]));
  asm volatile("" ::"v"(bv), "v"(ka), "v"(kb), "v"(kc));

#pragma unroll 1
  for (int j = 0; j < J; ++j) {
    // ================= matrix phase =================
    QSTAMP(j, 0);   // (the loads of unit j + 1 are in flight)
    // TN = the pixel tiles this wave really has: TMAX, or one less for the last pixel groups when the band's tiles do not
    // divide by NSPLIT (a wave-uniform choice between two complete loops - a branch on `tile < NT` INSIDE the unrolled loop
    // made the compiler copy the accumulators from block to block, spilling; multiplying a clamped copy instead, as before,
    // cost 1 / 12 of the matrix instructions of the 129x174 and 65x88 layers, and matrix time is not hidden by anything)
    auto matrix_phase = [&](auto tn_c) __attribute__((always_inline)) {
      constexpr int TN = decltype(tn_c)::value;
      if constexpr (TN == 0) {   // (a wave without a tile in a sample's last band)
        QSTAMP(j, 2);
        if (FUSE) fetch_a(j);
      } else {
      f32x4 acc[MW][TN];
#pragma unroll
      for (int mw = 0; mw < MW; ++mw)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[mw][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      // steps (kh, g, group of TP tiles) in order; the fragments of step i + PD are requested before the products of step i
      // are issued.  A step multiplies MW x TP = 2 (M tile, pixel tile) pairs: two independent accumulation chains,
      // interleaved term by term, with the LDS reads pinned in front of them (sched_group_barrier: left to itself the
      // scheduler sinks the reads behind most of the previous step's instructions and the wave waits for LDS every step).
      constexpr int TP = 2 / MW, TG = (TN + TP - 1) / TP;
      constexpr int NSTEP = G::KHW * NG * TG, PD = G::PD < NSTEP ? G::PD : NSTEP;
      static_assert(MW == 1 || MW == 2, "two chains per step");
      u32x4 bf[PD + 1][TP][NPL];
      auto frag = [&](int i, u32x4 (&f)[TP][NPL]) {
        const int tg = i % TG, g = (i / TG) % NG, kh = i / (TG * NG);
        const int off = (g * 4 * CPS + kh * NPL * WPD) * 4;
#pragma unroll
        for (int q = 0; q < TP; ++q) {
          const int t = min(tg * TP + q, TN - 1);
#pragma unroll
          for (int p = 0; p < NPL; ++p) f[q][p] = *reinterpret_cast<const u4a8*>(lds_x + blo[t] + off + p * WPD * 4);
        }
      };
#pragma unroll
      for (int i = 0; i < PD; ++i) frag(i, bf[i]);
#pragma unroll
      for (int i = 0; i < NSTEP; ++i) {
        const int tg = i % TG, ks = i / TG;
        if (i + PD < NSTEP) frag(i + PD, bf[(i + PD) % (PD + 1)]);
#pragma unroll
        for (int term = 0; term < NTERM; ++term) {
          const int pa = NPL == 3 ? kTermA[term] : 0, pb = NPL == 3 ? kTermB[term] : 0;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int mw = MW == 2 ? c : 0, q = MW == 2 ? 0 : c, t = tg * TP + q;
            if (t < TN) acc[mw][t] = mfma_bf16_k32(af[mw][ks][pa], bf[i % (PD + 1)][q][pb], acc[mw][t]);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NPL * TP, 0);    // the step's LDS reads ...
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTERM, 0);   // ... then its matrix instructions
        __builtin_amdgcn_sched_barrier(0);
      }
      QSTAMP(j, 2);
      // (the saved activations of the fused epilogue are requested only now: their registers are not live across the matrix
      // loop, and the tile write + the wait for the slowest wave at the barrier cover the latency)
      if (FUSE) fetch_a(j);
      float* ot = otile + kg * G::O_SLICE;
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int n = (ng + G::NSPLIT * t) * 16 + m;
        if (n < NPX) {
#pragma unroll
          for (int mw = 0; mw < MW; ++mw)
#pragma unroll
            for (int i = 0; i < 4; ++i) ot[((mtl + mw) * 16 + 4 * kq + i) * OSTR + n] = acc[mw][t][i];
        }
      }
      }   // (TN > 0)
    };
    // The NT tiles of a band are dealt over the NSPLIT pixel groups: a wave has C0 or C0 - 1 of them.  A sample's LAST band
    // has VRL < R valid rows (1 at all three sizes): only its NTL tiles that hold valid pixels are multiplied.  (The four
    // calls are written out: wrapped in a second generic lambda the same dispatch cost the 129x174 kernel 30 registers.)
    constexpr int C0 = TMAX, NF0 = G::NT - G::NSPLIT * (C0 - 1);
    constexpr int VRL = Hs - (G::BANDS - 1) * R, NTL = (VRL * Ws + 15) / 16;
    constexpr int CL = (NTL + G::NSPLIT - 1) / G::NSPLIT, NFL = NTL - G::NSPLIT * (CL - 1);
    // (not in the fused 129x174 and 33x45 kernels: the extra loop bodies make them spill 3 - 11 registers, for 2 - 4 % of
    // their matrix instructions)
    constexpr bool LAST_BAND_TILES = NTL < G::NT && !(FUSE && (MW == 1 || CS == 64));
    if (LAST_BAND_TILES && (u0 + j * grid) % G::BANDS == G::BANDS - 1) {
      if (NFL == G::NSPLIT || ng < NFL)
        matrix_phase(std::integral_constant<int, CL>());
      else
        matrix_phase(std::integral_constant<int, CL - 1>());
    } else {
      if (NF0 == G::NSPLIT || ng < NF0)
        matrix_phase(std::integral_constant<int, C0>());
      else
        matrix_phase(std::integral_constant<int, (C0 > 1 ? C0 - 1 : 1)>());
    }
    QSTAMP(j, 4);
    ws_sync();
    QSTAMP(j, 5);
    // ================= vector phase =================
    // The items come first (commit unit j + 1, re-issue for unit j + 2), the move-out and its stores after them: the band
    // loads waited for are then the OLDEST vector-memory operations in flight.
    QSTAMP(j, 6);
    if (j + 1 < J) vector_items(j + 1);
    QSTAMP(j, 7);
    {
      int nfl;
      size_t goff;
      tile_geom(j, nfl, goff);
      float* o_p = out + goff;
      const float* t_p = otile + och * OSTR;
      const int tail0 = nfl & ~3;
      // (every lane reads all of its quads - a lane without one reads a clamped quad and stores outside the buffer - so the
      // LDS reads of all quads are in flight together instead of one exec-masked block after the other)
      f32x4 v[QO];
#pragma unroll
      for (int i = 0; i < QO; ++i) v[i] = *reinterpret_cast<const f32x4*>(t_p + 4 * min(part + LPC * i, OSTR / 4 - 1));
#pragma unroll
      for (int k = 1; k < G::KSPLIT; ++k)   // (fixed order)
#pragma unroll
        for (int i = 0; i < QO; ++i) v[i] += *reinterpret_cast<const f32x4*>(t_p + k * G::O_SLICE + 4 * min(part + LPC * i, OSTR / 4 - 1));
#pragma unroll
      for (int i = 0; i < QO; ++i) {
        const int q4 = part + LPC * i;
        const bool on = 4 * q4 + 4 <= nfl;
        const float onf = on ? 1.f : 0.f;
        if (FUSE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[i][e] = pgv_bwd_apply(v[i][e], av[i][e], ka, kb, kc, actd);
            slot[i][e] = fmaf(v[i][e], onf, slot[i][e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = pgv_act_apply(v[i][e] + bv, ap);
          s[0] = fmaf((v[i][0] + v[i][1]) + (v[i][2] + v[i][3]), onf, s[0]);
          s[1] = fmaf((v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]), onf, s[1]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), out_rs,
                                               (int)(((unsigned)goff * 4u + 16u * (unsigned)q4) | (on ? 0u : 0x80000000u)), 0, 0);
      }
      if (part < nfl - tail0) {
        const int idx = tail0 + part;
        float vt = t_p[idx];
#pragma unroll
        for (int k = 1; k < G::KSPLIT; ++k) vt += t_p[k * G::O_SLICE + idx];
        if (FUSE) {
          vt = pgv_bwd_apply(vt, av_t, ka, kb, kc, actd);
          const int rr = idx / Ws, cc = idx - rr * Ws, cls = 2 * (rr & 1) + (cc & 1);
          s[0] += cls == 0 ? vt : 0.f;
          s[1] += cls == 1 ? vt : 0.f;
          s[2] += cls == 2 ? vt : 0.f;
          s[3] += cls == 3 ? vt : 0.f;
        } else {
          vt = pgv_act_apply(vt + bv, ap);
          s[0] += vt;
          s[1] += vt * vt;
        }
        o_p[idx] = vt;
      }
    }
    QSTAMP(j, 8);
    ws_sync();
    QSTAMP(j, 9);
  }

  // ---- per-channel sums of the workgroup
  if (FUSE) {
#pragma unroll
    for (int i = 0; i < QO; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = 4 * (part + LPC * i) + e, rr = idx / Ws, cc = idx - rr * Ws, cls = 2 * (rr & 1) + (cc & 1);
        s[0] += cls == 0 ? slot[i][e] : 0.f;
        s[1] += cls == 1 ? slot[i][e] : 0.f;
        s[2] += cls == 2 ? slot[i][e] : 0.f;
        s[3] += cls == 3 ? slot[i][e] : 0.f;
      }
  }
#pragma unroll
  for (int o = 1; o < LPC; o <<= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += __shfl_xor(s[k], o);
  if (part == 0) {
    const int copy = blockIdx.x & (PGV_CLS_COPIES - 1);
    if (FUSE) {
      if (fuse.gbias) atomicAdd(fuse.gbias + (fuse.gbias_copies ? copy * CS : 0) + och, (s[0] + s[1]) + (s[2] + s[3]));
      if (fuse.cls) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(fuse.cls + ((size_t)copy * CS + och) * 4 + k, s[k]);
      }
    } else if (stats) {
      double* sp = stats + (size_t)copy * stat_stride;
      atomicAdd(&sp[och], (double)s[0]);
      atomicAdd(&sp[CS + och], (double)s[1]);
    }
  }
}

template <class G>
int launch_down_q(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift, const float* bias,
                  int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse, hipStream_t st,
                  const pgv_bn_src* bn) {
  if (fuse && (stats || bias || in_scale || (bn && bn->stats))) return 0;   // (the fused form multiplies a gradient: no input affine)
  if ((int64_t)d->B * d->Cb * G::H * G::W * 4 >= (int64_t)1 << 31 || d->B <= 0) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const u32x4*, const float*, int, float, float*, double*,
                         int, pgv_bn_src, pgv_bwd_fuse QSTAMP_ARG);
  kern_t kern = fuse ? (kern_t)down_q_kernel<G, true> : (kern_t)down_q_kernel<G, false>;
  static bool attr_done[2] = {false, false};   // (per instantiation of the template)
  int rc = raise_lds_limit(kern, &attr_done[fuse ? 1 : 0], "conv_down_big_split");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cs, st) != hipSuccess) {
    pgv_set_error("conv_down_big_split: memset failed");
    return PGV_E_LAUNCH;
  }
  pgv_bwd_fuse f = {};
  if (fuse) f = *fuse;
  const int units = d->B * G::BANDS;
  hipLaunchKernelGGL(kern, dim3((unsigned)min(units, 256)), dim3(512), G::LDS_BYTES, st, d->B, big, in_scale, in_shift,
                     (const u32x4*)d->w_shadow, bias, act, slope, out, stats, (d->flags & PGV_STATS_COPIES) ? 2 * d->Cs : 0,
                     bn ? *bn : pgv_no_bn(), f QSTAMP_PASS);
  PGV_CHECK_LAUNCH("conv_down_big_split");
  return fuse && fuse->cls ? 3 : 1;   // (3: the class sums of the fused result are done)
}

// ---------------------------------------------------------------------------------------------------------------
// UP: big = conv_transpose_{s2,p2,k4}(small').  Four 2x2-tap phase convolutions sharing ONE input gather: GEMM with
// M = (output phase, big channel), K = (8 small channels) x the 4 taps of a phase, N = grid positions (u, v) of the band;
// output (2u + ph, 2v + pw) takes taps kh = ph + 2 th, kw = pw + 2 tw at input (u + 1 - th, v + 1 - tw).
template <int CB_, int CS_, int H_, int W_, int UB_, int MW_, int KSPLIT_, int NSPLIT_, int PD_ = 1, int NPL_ = 3>
struct UpQ {
  static constexpr int CB = CB_, CS = CS_, H = H_, W = W_, UB = UB_, MW = MW_, KSPLIT = KSPLIT_, NSPLIT = NSPLIT_, PD = PD_;
  static constexpr int NPL = NPL_, NTERM = NPL_ == 3 ? 6 : 1;   // (operand planes / product terms: see DownQ)
  static_assert(NPL_ == 3 || NPL_ == 1, "three planes (six product terms) or one");
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, P = Hs * Ws, RB = 2 * UB;
  static constexpr int HU = (H + 1) / 2, WU = (W + 1) / 2, BANDS = (HU + UB - 1) / UB;   // grid rows / columns with an output
  static constexpr int SWP = Ws + 1, SROWS = UB + 1, SPX = SROWS * SWP;                  // small band image (+ zero column)
  static constexpr int NG = CS / 8, PB = CS * 2;
  static constexpr int SH = NG == 8 ? 1 : (NG == 4 ? 2 : 3);   // group g of pixel px sits at g ^ ((px >> SH) & (NG - 1))
  static constexpr int KSTEPS = NG, KW = NG / KSPLIT;
  static constexpr int MTN = CB / 4, MG = MTN / MW;             // M tiles: rows (phase, channel); M groups over the waves
  static constexpr int NPOS = UB * WU, NT = (NPOS + 15) / 16, TMAX = (NT + NSPLIT - 1) / NSPLIT;
  static constexpr int OCH = RB * W, O_SLICE = CB * OCH, O_FLOATS = KSPLIT * O_SLICE;   // tile [CB][RB rows][W]
  static constexpr int IMG = (SPX * PB + 15) / 16 * 16, STAGE = NPL * IMG;
  static constexpr int S_RUN = SROWS * Ws, ITEMS = NG * S_RUN, QB = (ITEMS + 511) / 512;   // loader items: (group, pixel)
  static constexpr int LPC = 512 / CB, O4 = OCH / 4, QO = (O4 + LPC - 1) / LPC;
  static constexpr size_t LDS_BYTES = (size_t)STAGE + (size_t)O_FLOATS * 4 + sizeof(float) * (2 * CS + 8);
  static_assert(MG * KSPLIT * NSPLIT == 8 && MTN % MW == 0 && NG % KSPLIT == 0, "eight waves");
  static_assert((NG == 8 || NG == 4 || NG == 2) && (CB == 32 || CB == 16 || CB == 8) && LPC <= 64 && OCH % 4 == 0, "channel counts of the stack");
  static_assert(LDS_BYTES <= 160 * 1024 && (NPL - 1) * IMG + SPX * PB < 65536, "LDS budget / immediate offsets");
};

template <class G, bool FUSE>
__global__ __launch_bounds__(512) void up_q_kernel(int B, const float* __restrict__ small_in, const float* __restrict__ in_scale,
                                                   const float* __restrict__ in_shift, const u32x4* __restrict__ wsh,
                                                   const float* __restrict__ bias, int act, float slope,
                                                   float* __restrict__ out, double* __restrict__ stats, int stat_stride,
                                                   pgv_bn_src in_bn, pgv_bwd_fuse fuse QSTAMP_ARG) {
  constexpr int CB = G::CB, CS = G::CS, H = G::H, W = G::W, Hs = G::Hs, Ws = G::Ws, TMAX = G::TMAX, UB = G::UB, RB = G::RB;
  constexpr int NG = G::NG, PB = G::PB, SH = G::SH, SWP = G::SWP, OCH = G::OCH, LPC = G::LPC, QO = G::QO, MW = G::MW;
  constexpr int NPL = G::NPL, NTERM = G::NTERM;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char* lds_s = ldsb;
  float* otile = reinterpret_cast<float*>(ldsb + G::STAGE);
  float* aff = otile + G::O_FLOATS;   // [2*CS]
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mtl = (wave % G::MG) * MW, kg = (wave / G::MG) % G::KSPLIT, ng = wave / (G::MG * G::KSPLIT);
  const pgv_split_sel sel = pgv_split_sel_make();

  for (int i = tid; i < CS; i += 512) {
    float sc = 1.f, sh = 0.f;
    if (in_bn.stats)
      pgv_bn_finalize_dev(in_bn, CS, i, blockIdx.x == 0, sc, sh);
    else if (in_scale)
      sc = in_scale[i], sh = in_shift[i];
    aff[i] = sc;
    aff[CS + i] = sh;
  }
  // ---- this wave's weight fragments: M tiles mtl .. mtl + MW - 1, K steps [kg * KW, + KW), three planes
  u32x4 af[MW][G::KW][NPL];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw) {
    const u32x4* a_src = wsh + ((size_t)((mtl + mw) * NG + kg * G::KW) * NPL) * 64 + lane;
#pragma unroll
    for (int k = 0; k < G::KW; ++k)
#pragma unroll
      for (int p = 0; p < NPL; ++p) af[mw][k][p] = a_src[(k * NPL + p) * 64];
  }
  // ---- this lane's accumulator rows: (phase, channels c0 .. c0 + 3) of the row list (phase, channel), per M tile of the wave
  int ph_[MW], pw_[MW], c0[MW];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw) {
    const int rl0 = (mtl + mw) * 16 + 4 * kq, phase = rl0 / CB;
    ph_[mw] = phase >> 1;
    pw_[mw] = phase & 1;
    c0[mw] = rl0 - phase * CB;
  }
  const int th = kq >> 1, twp = kq & 1;
  int boff[TMAX], bsw[TMAX], opos[TMAX];   // (opos: tile position of output (2 ul, 2 v), -1 = no such grid position)
  bool oedge[TMAX];                        // (its column 2 v + 1 lies outside the row)
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const int n = (ng + G::NSPLIT * t) * 16 + m, nn = min(n, G::NPOS - 1), ul = nn / G::WU, v = nn - ul * G::WU;
    const int px = (ul + 1 - th) * SWP + (v + 1 - twp);
    boff[t] = px * PB;
    bsw[t] = (px >> SH) & (NG - 1);
    opos[t] = (ng + G::NSPLIT * t < G::NT && n < G::NPOS) ? (2 * ul) * W + 2 * v : -1;
    oedge[t] = 2 * v + 1 >= W;
  }
  // ---- loader items: channel groups x the pixels of the band's contiguous run; a lane loads the 8 channels of its pixel
  // (4-byte loads, consecutive lanes = consecutive pixels) and stores 16 bytes per plane
  int l_src[G::QB], l_dst[G::QB], l_gi[G::QB];
#pragma unroll
  for (int i = 0; i < G::QB; ++i) {
    const int q = min(tid + 512 * i, G::ITEMS - 1), g = q / G::S_RUN, idx = q - g * G::S_RUN;
    const int rr = idx / Ws, cc = idx - rr * Ws, px = rr * SWP + cc;
    l_src[i] = (8 * g) * G::P + idx;             // + sample * CS * P + first row * Ws; + c * P per channel
    l_dst[i] = px * PB + ((g ^ ((px >> SH) & (NG - 1))) * 16);
    l_gi[i] = (g << 16) | idx | ((tid + 512 * i < G::ITEMS) ? 0x8000 : 0);
  }
  // ---- move-out role: LPC lanes per channel
  const int och = tid / LPC, part = tid % LPC;
  const pgv_act_params ap = pgv_act_setup(act, slope);
  const float bv = (!FUSE && bias) ? bias[och] : 0.f;
  float ka = 1.f, kb = 0.f, kc = 0.f;
  pgv_actd_params actd = pgv_actd_setup(PGV_ACT_NONE, 0.f);
  if (FUSE) {
    ka = fuse.coef[och], kb = fuse.coef[CB + och], kc = fuse.coef[2 * CB + och];
    actd = pgv_actd_setup(fuse.act, fuse.slope);
  }
  float s1 = 0.f, s2 = 0.f;   // forward: sum / sum of squares of the outputs; fused: sum of g_y (bias gradient)

  const int grid = (int)gridDim.x, u0 = pgv_xcd_block(), units = B * G::BANDS;
  const int J = u0 < units ? (units - 1 - u0) / grid + 1 : 0;

  // (software-pipelined item by item through the vector phase, loads outside the run or beyond the last unit read outside
  // the buffer: see down_q_kernel)
  float rb[G::QB][8];
  const __amdgpu_buffer_rsrc_t small_rs = tensor_rsrc(small_in, (int64_t)B * CS * G::P * 4);
  struct UnitPos { unsigned base, kill; int nvalid; };
  auto unit_pos = [&](int j) {
    const int u = min(u0 + j * grid, units - 1), b = u / G::BANDS, band = u - b * G::BANDS, r0 = band * UB;
    // nvalid: floats of the band's run inside the plane (rows beyond it are zeros - and stay zero under an affine)
    return UnitPos{((unsigned)b * (unsigned)(CS * G::P) + (unsigned)(r0 * Ws)) * 4u, j < J ? 0u : 0x80000000u,
                   (min(Hs, r0 + G::SROWS) - r0) * Ws};
  };
  auto issue_item = [&](int i, const UnitPos& up) {
    const unsigned o = (up.base + 4u * (unsigned)l_src[i]) | up.kill | ((l_gi[i] & 0x7fff) < up.nvalid ? 0u : 0x80000000u);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      rb[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(small_rs, (int)(o + (unsigned)(c * G::P * 4)), 0, 0));
  };
  auto commit_item = [&](int i, const UnitPos& up, const f32x4 (&af4)[4]) {   // af4: scale[c .. c+7], shift[c .. c+7]
    if (512 * (i + 1) <= G::ITEMS || (l_gi[i] & 0x8000)) {   // (a full slot of items needs no lane mask)
      // (a pixel outside the plane arrived as zeros: only the shift has to vanish there; FUSE: no affine, see down_q_kernel)
      const float mk = (l_gi[i] & 0x7fff) < up.nvalid ? 1.f : 0.f;
      u32x4 ph, pm, pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y0 = FUSE ? rb[i][2 * e] : fmaf(rb[i][2 * e], af4[e >> 1][2 * (e & 1)], af4[2 + (e >> 1)][2 * (e & 1)] * mk);
        const float y1 = FUSE ? rb[i][2 * e + 1]
                              : fmaf(rb[i][2 * e + 1], af4[e >> 1][2 * (e & 1) + 1], af4[2 + (e >> 1)][2 * (e & 1) + 1] * mk);
        unsigned a1, a2 = 0, a3 = 0;
        if constexpr (NPL == 3)
          pgv_split3_pair(y0, y1, a1, a2, a3, sel);
        else
          a1 = pgv_pack_bf16x2(y0, y1);   // (bf16 operand mode: rounded to nearest, one plane)
        ph[e] = a1, pm[e] = a2, pl[e] = a3;
      }
      *reinterpret_cast<u32x4*>(lds_s + l_dst[i]) = ph;
      if constexpr (NPL == 3) {
        *reinterpret_cast<u32x4*>(lds_s + G::IMG + l_dst[i]) = pm;
        *reinterpret_cast<u32x4*>(lds_s + 2 * G::IMG + l_dst[i]) = pl;
      }
    }
  };
  auto vector_items = [&](int jc) {
    const UnitPos uc = unit_pos(jc), un = unit_pos(jc + 1);
    f32x4 af4[G::QB][4];   // (read for all items before the first commit: see down_q_kernel)
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      const int c = 8 * (l_gi[i] >> 16);
      if (!FUSE) {
        af4[i][0] = *reinterpret_cast<const f32x4*>(aff + c), af4[i][1] = *reinterpret_cast<const f32x4*>(aff + c + 4);
        af4[i][2] = *reinterpret_cast<const f32x4*>(aff + CS + c), af4[i][3] = *reinterpret_cast<const f32x4*>(aff + CS + c + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < G::QB; ++i) {
      commit_item(i, uc, af4[i]);
      issue_item(i, un);
    }
  };
  f4u av[QO];
  float av_t = 0.f;
  auto tile_geom = [&](int j, int& nfl, size_t& goff) {
    const int u = u0 + j * grid, b = u / G::BANDS, band = u - b * G::BANDS, y0 = band * RB;
    nfl = min(RB, H - y0) * W;
    goff = ((size_t)b * CB + och) * (H * W) + y0 * W;
  };
  // (buffer loads, a lane without an element reads outside the buffer: every lane issues them, so the compiler's counter
  // model counts them exactly and the waits for the band loads issued before them do not wait for these)
  const __amdgpu_buffer_rsrc_t a_rs = tensor_rsrc(FUSE ? fuse.a : nullptr, FUSE ? (int64_t)B * CB * (H * W) * 4 : 0);
  const __amdgpu_buffer_rsrc_t out_rs = tensor_rsrc(out, (int64_t)B * CB * (H * W) * 4);
  auto fetch_a = [&](int j) {
    int nfl;
    size_t goff;
    tile_geom(j, nfl, goff);
    const unsigned a_o = (unsigned)goff * 4u;
#pragma unroll
    for (int i = 0; i < QO; ++i) {
      const int q4 = part + LPC * i;
      av[i] = buffer_load_x4(a_rs, (a_o + 16u * q4) | (4 * q4 + 4 <= nfl ? 0u : 0x80000000u));
    }
    const int tail0 = nfl & ~3;
    av_t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, (int)((a_o + 4u * (tail0 + part)) | (part < nfl - tail0 ? 0u : 0x80000000u)), 0, 0));
  };

  if (J > 0) {
    const UnitPos p0 = unit_pos(0);
#pragma unroll
    for (int i = 0; i < G::QB; ++i) issue_item(i, p0);
  }
  // (the image is cleared behind the first unit's loads: their latency covers it)
  for (int i = tid; i < (int)((G::STAGE + G::O_FLOATS * 4) / 16); i += 512) reinterpret_cast<u32x4*>(ldsb)[i] = u32x4{0, 0, 0, 0};
  __syncthreads();   // image zeroed, affine staged
  if (J > 0) vector_items(0);
  __syncthreads();
  // Everything the prologue loaded is READ here: the compiler may sink those loads (read-only data) below the barriers, and
  // with any of them pending at the loop header its counter model waits for "them" inside the loop - s_waitcnt vmcnt(n)
  // with n counting down through the matrix loop, i.e. for the band loads just issued, and vmcnt(0) in front of the next
  // issue, i.e. for the stores of the vector phase (1.6 - 2.9 k clocks per unit).
#pragma unroll
  for (int mw = 0; mw < MW; ++mw)
#pragma unroll
    for (int k = 0; k < G::KW; ++k)
#pragma unroll
      for (int p = 0; p < NPL; ++p) asm volatile("" ::"v"(af[mw][k][p]));
  asm volatile("" ::"v"(bv), "v"(ka), "v"(kb), "v"(kc));

#pragma unroll 1
  for (int j = 0; j < J; ++j) {
    // ================= matrix phase =================
    QSTAMP(j, 0);   // (the loads of unit j + 1 are in flight)
    auto matrix_phase = [&](auto tn_c) __attribute__((always_inline)) {   // (TN = the position tiles this wave really has: see down_q_kernel)
      constexpr int TN = decltype(tn_c)::value;
      if constexpr (TN == 0) {
        QSTAMP(j, 2);
        if (FUSE) fetch_a(j);
      } else {
      f32x4 acc[MW][TN];
#pragma unroll
      for (int mw = 0; mw < MW; ++mw)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[mw][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      // (steps of two accumulation chains, reads pinned in front: see down_q_kernel)
      constexpr int TP = 2 / MW, TG = (TN + TP - 1) / TP;
      constexpr int NSTEP = G::KW * TG, PD = G::PD < NSTEP ? G::PD : NSTEP;
      static_assert(MW == 1 || MW == 2, "two chains per step");
      u32x4 bf[PD + 1][TP][NPL];
      auto frag = [&](int i, u32x4 (&f)[TP][NPL]) {
        const int tg = i % TG, g = kg * G::KW + i / TG;
#pragma unroll
        for (int q = 0; q < TP; ++q) {
          const int t = min(tg * TP + q, TN - 1);
          const unsigned char* bp = lds_s + boff[t] + ((g ^ bsw[t]) * 16);
#pragma unroll
          for (int p = 0; p < NPL; ++p) f[q][p] = *reinterpret_cast<const u32x4*>(bp + p * G::IMG);
        }
      };
#pragma unroll
      for (int i = 0; i < PD; ++i) frag(i, bf[i]);
#pragma unroll
      for (int i = 0; i < NSTEP; ++i) {
        const int tg = i % TG, ks = i / TG;
        if (i + PD < NSTEP) frag(i + PD, bf[(i + PD) % (PD + 1)]);
#pragma unroll
        for (int term = 0; term < NTERM; ++term) {
          const int pa = NPL == 3 ? kTermA[term] : 0, pb = NPL == 3 ? kTermB[term] : 0;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int mw = MW == 2 ? c : 0, q = MW == 2 ? 0 : c, t = tg * TP + q;
            if (t < TN) acc[mw][t] = mfma_bf16_k32(af[mw][ks][pa], bf[i % (PD + 1)][q][pb], acc[mw][t]);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NPL * TP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTERM, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      QSTAMP(j, 2);
      if (FUSE) fetch_a(j);   // (requested only now: see down_q_kernel)
      float* ot = otile + kg * G::O_SLICE;   // (waves that split K write a tile each, added up in the move-out)
#pragma unroll
      for (int t = 0; t < TN; ++t) {
#pragma unroll
        for (int mw = 0; mw < MW; ++mw) {
          // (a position whose output column 2 v + pw lies outside the row is dropped; rows outside the plane are never moved out)
          if (opos[t] >= 0 && !(oedge[t] && pw_[mw])) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ot[(c0[mw] + i) * OCH + opos[t] + ph_[mw] * W + pw_[mw]] = acc[mw][t][i];
          }
        }
      }
      }   // (TN > 0)
    };
    // (tiles of this wave, a sample's last band: see down_q_kernel)
    constexpr int C0 = TMAX, NF0 = G::NT - G::NSPLIT * (C0 - 1);
    constexpr int VRL = G::HU - (G::BANDS - 1) * UB, NTL = (VRL * G::WU + 15) / 16;   // valid grid rows / tiles of the last band
    constexpr int CL = (NTL + G::NSPLIT - 1) / G::NSPLIT, NFL = NTL - G::NSPLIT * (CL - 1);
    if (NTL < G::NT && (u0 + j * grid) % G::BANDS == G::BANDS - 1) {
      if (NFL == G::NSPLIT || ng < NFL)
        matrix_phase(std::integral_constant<int, CL>());
      else
        matrix_phase(std::integral_constant<int, CL - 1>());
    } else {
      if (NF0 == G::NSPLIT || ng < NF0)
        matrix_phase(std::integral_constant<int, C0>());
      else
        matrix_phase(std::integral_constant<int, (C0 > 1 ? C0 - 1 : 1)>());
    }
    QSTAMP(j, 4);
    ws_sync();
    QSTAMP(j, 5);
    // ================= vector phase ================= (items first, then the move-out: see down_q_kernel)
    QSTAMP(j, 6);
    if (j + 1 < J) vector_items(j + 1);
    QSTAMP(j, 7);
    {
      int nfl;
      size_t goff;
      tile_geom(j, nfl, goff);
      float* o_p = out + goff;
      const float* t_p = otile + och * OCH;
      const int tail0 = nfl & ~3;
      f32x4 v[QO];   // (all quads read before the first is processed: see down_q_kernel)
#pragma unroll
      for (int i = 0; i < QO; ++i) v[i] = *reinterpret_cast<const f32x4*>(t_p + 4 * min(part + LPC * i, G::O4 - 1));
#pragma unroll
      for (int k = 1; k < G::KSPLIT; ++k)   // (fixed order)
#pragma unroll
        for (int i = 0; i < QO; ++i) v[i] += *reinterpret_cast<const f32x4*>(t_p + k * G::O_SLICE + 4 * min(part + LPC * i, G::O4 - 1));
#pragma unroll
      for (int i = 0; i < QO; ++i) {
        const int q4 = part + LPC * i;
        const bool on = 4 * q4 + 4 <= nfl;
        const float onf = on ? 1.f : 0.f;
        float q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (FUSE) {
            v[i][e] = pgv_bwd_apply(v[i][e], av[i][e], ka, kb, kc, actd);
          } else {
            v[i][e] = pgv_act_apply(v[i][e] + bv, ap);
            q2 += v[i][e] * v[i][e];
          }
          q1 += v[i][e];
        }
        s1 = fmaf(q1, onf, s1);
        if (!FUSE) s2 = fmaf(q2, onf, s2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), out_rs,
                                               (int)(((unsigned)goff * 4u + 16u * (unsigned)q4) | (on ? 0u : 0x80000000u)), 0, 0);
      }
      if (part < nfl - tail0) {   // (a last band of an odd number of odd-width rows)
        const int idx = tail0 + part;
        float vt = t_p[idx];
#pragma unroll
        for (int k = 1; k < G::KSPLIT; ++k) vt += t_p[k * G::O_SLICE + idx];
        if (FUSE) {
          vt = pgv_bwd_apply(vt, av_t, ka, kb, kc, actd);
        } else {
          vt = pgv_act_apply(vt + bv, ap);
          s2 += vt * vt;
        }
        s1 += vt;
        o_p[idx] = vt;
      }
    }
    QSTAMP(j, 8);
    ws_sync();
    QSTAMP(j, 9);
  }

  // ---- per-channel sums of the workgroup: BatchNorm statistics (forward) or the bias gradient (fused backward)
#pragma unroll
  for (int o = 1; o < LPC; o <<= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if (part == 0) {
    const int copy = blockIdx.x & (PGV_CLS_COPIES - 1);
    if (FUSE) {
      if (fuse.gbias) atomicAdd(fuse.gbias + (fuse.gbias_copies ? copy * CB : 0) + och, s1);
    } else if (stats) {
      double* sp = stats + (size_t)copy * stat_stride;
      atomicAdd(&sp[och], (double)s1);
      atomicAdd(&sp[CB + och], (double)s2);
    }
  }
}

template <class G>
int launch_up_q(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift, const float* bias,
                int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse, hipStream_t st, const pgv_bn_src* bn) {
  // (class sums of a big-tensor result: the band kernels keep those calls; the fused form has no input affine)
  if (fuse && (fuse->cls || stats || bias || in_scale || (bn && bn->stats))) return 0;
  if ((int64_t)d->B * d->Cb * G::H * G::W * 4 >= (int64_t)1 << 31 || d->B <= 0) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const u32x4*, const float*, int, float, float*, double*,
                         int, pgv_bn_src, pgv_bwd_fuse QSTAMP_ARG);
  kern_t kern = fuse ? (kern_t)up_q_kernel<G, true> : (kern_t)up_q_kernel<G, false>;
  static bool attr_done[2] = {false, false};
  int rc = raise_lds_limit(kern, &attr_done[fuse ? 1 : 0], "conv_up_big_split");
  if (rc) return rc;
  if (stats && !(d->flags & PGV_PREZEROED) && hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cb, st) != hipSuccess) {
    pgv_set_error("conv_up_big_split: memset failed");
    return PGV_E_LAUNCH;
  }
  pgv_bwd_fuse f = {};
  if (fuse) f = *fuse;
  const int units = d->B * G::BANDS;
  // (after the down layout: NPL planes x 16 taps x 2 bytes per weight = 96 bytes per weight with three planes, 32 with one)
  const u32x4* up = (const u32x4*)d->w_shadow + (size_t)d->Cs * d->Cb * 2 * G::NPL;
  hipLaunchKernelGGL(kern, dim3((unsigned)min(units, 256)), dim3(512), G::LDS_BYTES, st, d->B, small_in, in_scale, in_shift, up,
                     bias, act, slope, out, stats, (d->flags & PGV_STATS_COPIES) ? 2 * d->Cb : 0, bn ? *bn : pgv_no_bn(),
                     f QSTAMP_PASS);
  PGV_CHECK_LAUNCH("conv_up_big_split");
  return 1;
}

}  // namespace

static bool big_plane_layer(const pgv_conv_desc* d) {
  if (d->kh != 4 || d->kw != 4 || d->stride != 2 || d->pad != 2) return false;
  return (d->Hb == 33 && d->Wb == 45 && d->Cb == 32 && d->Cs == 64) || (d->Hb == 65 && d->Wb == 88 && d->Cb == 16 && d->Cs == 32) ||
         (d->Hb == 129 && d->Wb == 174 && d->Cb == 8 && d->Cs == 16);
}
// fp32 mode, products as six bf16 instructions (three operand planes)
bool pgv_big_split_shape(const pgv_conv_desc* d) {
  return (d->flags & PGV_COMPUTE_F32_SPLIT) && !(d->flags & PGV_COMPUTE_BF16) && big_plane_layer(d);
}
// bf16 operand mode on the same kernels with ONE operand plane (round 6: the mode ran round-4 kernels that had become slower
// than the six-instruction fp32 ones - 100.9 / 76 us against 90.5 / 83 us on 129x174 with a sixth of the matrix work)
bool pgv_big_bf16q_shape(const pgv_conv_desc* d) { return (d->flags & PGV_COMPUTE_BF16) && big_plane_layer(d); }

// 1 / 3 = launched (3: with the class sums of the fused epilogue), 0 = not this kernel family's case
int pgv_conv_down_big_split(const pgv_conv_desc* d, const float* big, const float* in_scale, const float* in_shift,
                            const float* bias, int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse,
                            hipStream_t st, const pgv_bn_src* bn) {
  if (!d->w_shadow) return 0;
  if (pgv_big_bf16q_shape(d)) {   // one operand plane (same tilings)
    if (d->Hb == 33) return launch_down_q<DownQ<32, 64, 33, 45, 2, 2, 4, 1, 1, 1>>(d, big, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
    if (d->Hb == 65) return launch_down_q<DownQ<16, 32, 65, 88, 4, 2, 2, 4, 1, 1>>(d, big, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
    return launch_down_q<DownQ<8, 16, 129, 174, 4, 1, 1, 8, 2, 1>>(d, big, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
  }
  if (!pgv_big_split_shape(d)) return 0;
  if (d->Hb == 33)   // 32 -> 64 channels: bands of 2 rows (46 pixels = 3 tiles), waves = 2 M pairs x 4 kernel rows
    return launch_down_q<DownQ<32, 64, 33, 45, 2, 2, 4, 1, 1>>(d, big, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
  if (d->Hb == 65)   // 16 -> 32 channels: bands of 4 rows (180 pixels = 12 tiles), waves = 2 K halves x 4 pixel groups
    return launch_down_q<DownQ<16, 32, 65, 88, 4, 2, 2, 4, 1>>(d, big, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
  // 8 -> 16 channels: bands of 4 rows (352 pixels = 22 tiles), one M tile, waves = 8 pixel groups over the whole K (one
  // tile slice to write and move out instead of two K halves: -1 us)
  return launch_down_q<DownQ<8, 16, 129, 174, 4, 1, 1, 8, 2>>(d, big, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
}

int pgv_conv_up_big_split(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                          const float* bias, int act, float slope, float* out, double* stats, const pgv_bwd_fuse* fuse,
                          hipStream_t st, const pgv_bn_src* bn) {
  if (!d->w_shadow) return 0;
  if (pgv_big_bf16q_shape(d)) {   // one operand plane (same tilings)
    if (d->Hb == 33) return launch_up_q<UpQ<32, 64, 33, 45, 2, 2, 2, 1, 1, 1>>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
    if (d->Hb == 65) return launch_up_q<UpQ<16, 32, 65, 88, 4, 2, 1, 4, 1, 1>>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
    return launch_up_q<UpQ<8, 16, 129, 174, 4, 2, 1, 8, 2, 1>>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
  }
  if (!pgv_big_split_shape(d)) return 0;
  if (d->Hb == 33)   // 64 -> 32 channels: bands of 2 grid rows (46 positions = 3 tiles), waves = 4 M pairs x 2 K halves (bands of 4 spill)
    return launch_up_q<UpQ<32, 64, 33, 45, 2, 2, 2, 1, 1>>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
  if (d->Hb == 65)   // 32 -> 16 channels: bands of 4 grid rows (176 positions = 11 tiles), waves = 2 M pairs x 4 position groups
    return launch_up_q<UpQ<16, 32, 65, 88, 4, 2, 1, 4, 1>>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
  // 16 -> 8 channels: bands of 4 grid rows (348 positions = 22 tiles), one M pair, waves = 8 position groups
  return launch_up_q<UpQ<8, 16, 129, 174, 4, 2, 1, 8, 2>>(d, small_in, in_scale, in_shift, bias, act, slope, out, stats, fuse, st, bn);
}
