// Weight gradient of the stride-2 k=4 layers in bf16 operand mode (PGV_COMPUTE_BF16), bf16-NATIVE: both operands are
// rounded once, where they are committed to LDS, and live there as bfloat16; the contraction runs on
// v_mfma_f32_16x16x32_bf16 with 16-byte fragment reads and no packing in the loop.
//   gW[cs][cb][kh][kw] = sum_{b,oh,ow} S[b,cs,oh,ow] * X[b,cb,2oh-2+kh,2ow-2+kw]        (model/layer.py:10-46, autograd)
// GEMM view: M = cs, N = (cb, kh, kw), K = 32 consecutive output pixels ow of one output row per MFMA.  The stride-2 column
// walk becomes contiguous by splitting X into its EVEN and ODD columns while it is committed (Xe[j] = X[2j], Xo[j] = X[2j+1]):
//   kw = 2 + p (p = 0, 1):  column 2ow + p      = parity p, j = ow        ->  sum_ow S[ow]  Xp[ow]
//   kw = p:                 column 2(ow-1) + p  = parity p, j = ow - 1    ->  sum_j  S1[j] Xp[j],   S1[j] = S[j+1]
// so a column tile of the MFMA is (two big channels) x (4 kernel rows) x (2 parities) of ONE half of the kernel columns:
// the half decides whether the A fragment comes from S or from its copy shifted by one pixel (S1, written next to S at
// commit), and every fragment is eight consecutive bf16 at a 16-byte aligned address.
// Plane strides are padded so that both ds_read_b128 patterns are bank-conflict free (16 distinct 16-byte slots per lane
// group; the search is in DESIGN.md 3.10).  Partial gradients go to the workspace in the layout of gw, one slot per
// workgroup, and the reduce launch of conv_v2_wgrad.hip adds them (with the tap-sum / bias roles of the step).
#include "conv_tile.h"
#include "band_prefetch.h"

namespace {

typedef unsigned short u16;

template <int CB_, int CS_, int W_, int H_, int R_, int XCH_, int XPL_, int SCH_, int WPC_ = 2>
struct WgB16 {
  static constexpr int CB = CB_, CS = CS_, W = W_, H = H_, R = R_;
  static constexpr int WPC = WPC_;   // workgroups per CU the register budget is sized for
  static constexpr int Ws = W / 2 + 1, Hs = H / 2 + 1, BANDS = (Hs + R - 1) / R;
  static constexpr int KC = (Ws + 31) / 32, SROW = 32 * KC, XROW = SROW, XR = 2 * R + 2;
  static constexpr int XCH = XCH_, XPL = XPL_, SCH = SCH_;   // bf16 elements: X channel / parity-plane strides, S channel stride
  static constexpr int WPB = (W + 2 + 3) / 4 * 4, WPS = (Ws + 2 + 3) / 4 * 4;   // chunk grids of the two prefetchers
  static constexpr int MT = CS / 16, CT = CB / 4;             // M tiles; column tiles per wave (CB tiles over 4 waves)
  static constexpr size_t S_ELEMS = (size_t)CS * SCH, X_ELEMS = 2 * (size_t)XPL;
  static constexpr size_t LDS_BYTES = 2 * (2 * S_ELEMS + X_ELEMS) + sizeof(float) * 2 * (CB + CS);
  static_assert(XCH >= XR * XROW && XPL >= CB * XCH && SCH >= R * SROW, "plane strides");
  static_assert(XCH % 8 == 0 && XPL % 8 == 0 && SCH % 8 == 0, "16-byte aligned fragments");
  static_assert(WPB / 2 <= XROW && WPS <= SROW, "committed columns stay inside a row");
  static_assert(CS % 16 == 0 && CB % 8 == 0, "tiles");
};

template <class G, bool BIG_AFF, bool SMALL_AFF>
__global__ __launch_bounds__(256, G::WPC) void conv_wgrad_bf16_kernel(int B, const float* __restrict__ big,
                                                               const float* __restrict__ big_scale,
                                                               const float* __restrict__ big_shift,
                                                               const float* __restrict__ small_in,
                                                               const float* __restrict__ small_scale,
                                                               const float* __restrict__ small_shift,
                                                               float* __restrict__ partial) {
  constexpr int CB = G::CB, CS = G::CS, W = G::W, H = G::H, R = G::R, Ws = G::Ws, Hs = G::Hs, BANDS = G::BANDS;
  constexpr int KC = G::KC, SROW = G::SROW, XROW = G::XROW, XCH = G::XCH, XPL = G::XPL, SCH = G::SCH, MT = G::MT, CT = G::CT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  u16* s0 = reinterpret_cast<u16*>(lds_raw);          // S  [CS][SCH]
  u16* s1 = s0 + G::S_ELEMS;                            // S1 [CS][SCH]: S shifted by one pixel
  u16* xp = s1 + G::S_ELEMS;                            // [2 parities][XPL]
  float* aff_b = reinterpret_cast<float*>(xp + G::X_ELEMS);   // [2][CB]
  float* aff_s = aff_b + 2 * CB;                              // [2][CS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int units = B * BANDS;

  // everything that is never committed (pad columns, pad rows of the strides) must read as finite zeros
  for (int i = tid; i < (int)((2 * (2 * G::S_ELEMS + G::X_ELEMS)) / 16); i += 256)
    reinterpret_cast<f32x4*>(lds_raw)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (BIG_AFF) stage_affine(aff_b, big_scale, big_shift, CB, tid);
  if (SMALL_AFF) stage_affine(aff_s, small_scale, small_shift, CS, tid);

  FlatPrefetch<CB, G::XR, W, G::WPB, H> pfx;
  FlatPrefetch<CS, R, Ws, G::WPS, Hs> pfs;
  pfx.init(tid);
  pfs.init(tid);
  auto issue_unit = [&](int u) {
    const int b = u / BANDS, band = u - b * BANDS;
    pfx.issue(big + (int64_t)b * CB * (H * W), 2 * band * R - 2, CB);
    pfs.issue(small_in + (int64_t)b * CS * (Hs * Ws), band * R, CS);
  };
  auto pack2 = [](float a, float b) -> unsigned {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)a, (__bf16)b};   // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, v);
  };
  auto commit_unit = [&]() {
    pfx.each(BIG_AFF ? aff_b : nullptr, CB, 0, tid, [&](int c, int rr, int q, f32x4 x) {
      const int e = c * XCH + rr * XROW + 2 * q;                 // even / odd columns 4q, 4q+2 / 4q+1, 4q+3 -> j = 2q, 2q+1
      *reinterpret_cast<unsigned*>(xp + e) = pack2(x.x, x.z);
      *reinterpret_cast<unsigned*>(xp + XPL + e) = pack2(x.y, x.w);
    });
    pfs.each(SMALL_AFF ? aff_s : nullptr, CS, 0, tid, [&](int c, int rr, int q, f32x4 x) {
      const int e = c * SCH + rr * SROW + 4 * q;
      const unsigned lo = pack2(x.x, x.y), hi = pack2(x.z, x.w);
      *reinterpret_cast<uint2*>(s0 + e) = uint2{lo, hi};
      // S1[j] = S[j + 1]: the four values land one element earlier (2-byte stores; index -1 does not exist)
      if (q > 0) s1[e - 1] = (u16)(lo & 0xFFFFu);
      s1[e] = (u16)(lo >> 16);
      s1[e + 1] = (u16)(hi & 0xFFFFu);
      s1[e + 2] = (u16)(hi >> 16);
    });
  };

  // per-lane fragment bases (bf16 elements): A row i = lane & 15 of an M tile; B column n = lane & 15 = (big channel of the
  // pair, kernel row, parity); both: eight pixels from 8 * (lane >> 4) on
  const int n = lane & 15, kq = lane >> 4;
  const int offA = n * SCH + 8 * kq;
  const int offB = (n & 1) * XPL + (n >> 3) * XCH + ((n >> 1) & 3) * XROW + 8 * kq;
  f32x4 acc[MT][CT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  int u = pgv_xcd_block();
  if (u < units) issue_unit(u);
#pragma unroll 1
  for (; u < units; u += gridDim.x) {
    __syncthreads();   // the previous unit's fragment reads are complete (first pass: zero fill and tables are visible)
    commit_unit();
    if (u + (int)gridDim.x < units) issue_unit(u + gridDim.x);
    __syncthreads();
#pragma unroll
    for (int ohl = 0; ohl < R; ++ohl) {
#pragma unroll
      for (int m32 = 0; m32 < KC; ++m32) {
        u32x4 a0[MT], a1[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int e = offA + m * 16 * SCH + ohl * SROW + 32 * m32;
          a1[m] = *reinterpret_cast<const u32x4*>(s0 + e);   // kernel columns 2, 3: S itself
          a0[m] = *reinterpret_cast<const u32x4*>(s1 + e);   // kernel columns 0, 1: S shifted by one pixel
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          const int pair = (wave * CT + t) >> 1;               // (wave * CT is even: the half t & 1 is compile-time)
          const u32x4 bfrag = *reinterpret_cast<const u32x4*>(xp + offB + 2 * pair * XCH + 2 * ohl * XROW + 32 * m32);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][t] = mfma_bf16_k32((t & 1) ? a1[m] : a0[m], bfrag, acc[m][t]);
        }
      }
    }
  }
  // ---- this workgroup's partial gradient, layout of gw: D row (lane >> 4) * 4 + reg = cs within the M tile, column n
  float* pw = partial + (size_t)blockIdx.x * ((size_t)CS * CB * 16);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int ct = wave * CT + t, cb = 2 * (ct >> 1) + (n >> 3), kh = (n >> 1) & 3, kw = 2 * (ct & 1) + (n & 1);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int cs = m * 16 + kq * 4 + reg;
        pw[(cs * CB + cb) * 16 + kh * 4 + kw] = acc[m][t][reg];
      }
    }
}

template <class G>
int launch_wgb16(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                 const float* small_in, const float* small_scale, const float* small_shift, float* partial,
                 int64_t partial_bytes, int* nparts, hipStream_t st) {
  static_assert(G::LDS_BYTES <= (size_t)kMaxLds, "LDS budget");
  if (d->Cb != G::CB || d->Cs != G::CS || d->B <= 0) return 0;
  if (big_scale && small_scale) return 0;   // not a case of the train step
  const int units = d->B * G::BANDS;
  const int per_cu = (int)min((size_t)G::WPC, (size_t)kMaxLds / G::LDS_BYTES);
  const int grid = min(units, 256 * per_cu);
  if ((int64_t)grid * G::CS * G::CB * 16 * (int64_t)sizeof(float) > partial_bytes) return 0;
  typedef void (*kern_t)(int, const float*, const float*, const float*, const float*, const float*, const float*, float*);
  kern_t kern = big_scale ? (kern_t)conv_wgrad_bf16_kernel<G, true, false>
                          : (small_scale ? (kern_t)conv_wgrad_bf16_kernel<G, false, true>
                                         : (kern_t)conv_wgrad_bf16_kernel<G, false, false>);
  static const void* raised[3];
  const int slot = big_scale ? 0 : (small_scale ? 1 : 2);
  if (raised[slot] != (const void*)kern) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
      pgv_set_error("conv_wgrad_bf16: cannot raise the dynamic LDS limit");
      return PGV_E_LAUNCH;
    }
    raised[slot] = (const void*)kern;
  }
  *nparts = grid;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, st, d->B, big, big_scale, big_shift, small_in, small_scale,
                     small_shift, partial);
  PGV_CHECK_LAUNCH("conv_wgrad_bf16");
  return 1;
}

}  // namespace

// 1 = launched (*nparts partial gradients in ``partial``), 0 = shape not covered.  Strides (bf16 elements) from the
// conflict search: X channel / parity-plane stride, S channel stride.
int pgv_conv_wgrad_bf16_partial(const pgv_conv_desc* d, const float* big, const float* big_scale, const float* big_shift,
                                const float* small_in, const float* small_scale, const float* small_shift, float* partial,
                                int64_t partial_bytes, int* nparts, hipStream_t st) {
  if (d->stride != 2 || d->pad != 2 || d->kh != 4 || d->kw != 4 || !(d->flags & PGV_COMPUTE_BF16) || !partial) return 0;
  if (d->Hb == 129 && d->Wb == 174)
    return launch_wgb16<WgB16<8, 16, 174, 129, 4, 976, 7872, 400>>(d, big, big_scale, big_shift, small_in, small_scale,
                                                                 small_shift, partial, partial_bytes, nparts, st);
  if (d->Hb == 65 && d->Wb == 88)
    return launch_wgb16<WgB16<16, 32, 88, 65, 4, 656, 10528, 272>>(d, big, big_scale, big_shift, small_in, small_scale,
                                                                 small_shift, partial, partial_bytes, nparts, st);
  if (d->Hb == 33 && d->Wb == 45)
    return launch_wgb16<WgB16<32, 64, 45, 33, 3, 256, 8208, 112, 1>>(d, big, big_scale, big_shift, small_in, small_scale,
                                                                small_shift, partial, partial_bytes, nparts, st);
  return 0;
}
