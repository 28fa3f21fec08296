// Third-generation kernels of the 1 <-> 8 channel 5x5 stride-2 end layers at the reference size (round 6): the input
// band lives in an LDS RING of image rows that is filled by LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS with no
// register stage and no commit pass), a workgroup walks a sample top to bottom, and while it multiplies the rows of step n
// the rows of step n + 1 are in flight.
//
// Why (phase toggles on up_c1_v2_kernel, conv_direct2.hip, B = 256, 88 us): without its loads 68 us, without its FMAs 61,
// without its stores 71 - the phases of a unit (18 loads per lane, commit through registers with the producer's BatchNorm
// affine, 480 packed FMAs per output quad, stores) ADD UP instead of overlapping, and every unit re-reads 2 halo rows of
// 13.  Here
//   * the affine never touches the band: out = sum w (sc x + sh) = sum w (sc x) + sh-term; the scale goes onto the 18
//     inputs a lane reads per channel, the shift term is a constant per output phase - except on the image border, where
//     taps that fall on the zero padding must not contribute it: a table T[row case][column case][ph][pw] of 36 sums;
//   * the weights are scalar loads (uniform addresses of a read-only tensor -> SGPR operands of the FMAs): as an LDS
//     table they were 7 of the 16 ds_read_b128 per channel and quad, and the LDS pipe was the busiest unit (77 -> 71 us);
//   * rows are DMA'd as they lie: a ring slot = [8 channels][192 floats] = 6 wave-instructions of 1 KB, chunks 44..47 of a
//     channel row are out-of-range reads (zeros = right padding = left padding of the next row), floats 174, 175 are the
//     next image row's first two and are cleared by the wave that issued the row, rows outside the image are cleared by
//     that wave with LDS stores;
//   * no halo: a row is fetched once per segment of a sample (2 seam rows per 65 at B = 256).
// The bf16 operand mode (operands rounded after the affine) stays on conv_direct2.hip.
//
// Only the transposed (8 -> 1 channel) direction is built this way.  The same ring for the 1 -> 8 channel convolution
// (enc1 forward, the output layer's fused input gradient) was written and measured in round 6 and dropped: 74 - 78 us
// against 70 - 78 for down_c1_v2 (forward), 138 - 166 against 125 (fused; its epilogue registers leave 3 workgroups per
// CU).  Those launches write 184 MB; 16-byte stores stream at 4.6 TB/s aligned and 3.8 TB/s on these 8-byte aligned rows
// (scratch/ubench/store_align.hip), so their floor is ~60 / ~90 us and the staging scheme is not what holds them.
#include <stdlib.h>
#include "conv_tile.h"

namespace {

constexpr int K5 = 5, KK5 = 25;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS byte address of a pointer into the dynamic LDS array
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}

// descriptor of "the bytes [p, p + bytes)" (raw buffer, 32-bit data format; range-checked per lane)
__device__ __forceinline__ i32x4 make_rsrc(const float* p, int bytes) {
  const uint64_t a = (uint64_t)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32) & 0xFFFF);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000;
  return r;
}

// One image row of all CS channels into a ring slot: NI x (64 lanes x 16 bytes), LDS destination = slot + 1 KB * i + 16 *
// lane.  rsrc = the bytes from the row's first float (channel 0) to the end of the sample; voff[i] = byte offset of this
// lane's chunk from there (beyond every buffer for the pad chunks: those lanes fetch nothing, and the pad floats of the
// slot, zeroed once, stay zero whether the hardware writes a zero there or nothing).
// Inline asm ON PURPOSE: hipcc orders every later read of the ring behind a __builtin_amdgcn_raw_ptr_buffer_load_lds with
// s_waitcnt vmcnt(0) (it cannot tell the slots apart), which would serialise the rows in flight with the multiply phase
// they are meant to overlap.  The kernel waits by hand (vmcnt) before the barrier that publishes the rows.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int NI>
__device__ __forceinline__ void dma_row(i32x4 rsrc, const float* slot, const int (&voff)[NI]) {
  const unsigned base = lds_addr(slot);
#pragma unroll
  for (int i = 0; i < NI; ++i)
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(base + 1024u * i), "v"(voff[i]), "s"(rsrc) : "memory", "m0");
}
#pragma clang diagnostic pop

// ---- UP: out[b,0,2u+ph,2v+pw] = act(bias + sum_{cs,th,tw} x'[b,cs,u+1-th,v+1-tw] * w[cs,0,ph+2th,pw+2tw]) ----------
template <int CS, int H, int W, int S>
struct UpRingCfg {
  static constexpr int Hs = H / 2 + 1, Ws = W / 2 + 1, Hg = (H + 1) / 2, Wg = (W + 1) / 2;
  static constexpr int WsP = 192, ROWF = CS * WsP, NI = ROWF / 256;   // floats per ring slot, DMA instructions per row
  static constexpr int NR = 2 * S + 2;                                // rows u-1 .. u+S in use + S rows in flight
  static constexpr int QW = (Wg + 3) / 4, STEPS = (Hg + S - 1) / S;
  static constexpr int FRONT = 4;
  static constexpr int TILE = NR * ROWF;
  // FRONT | ring | T [3][3][4] (+ pad) | aff [2 CS] | wsh [25] (+ pad)
  static constexpr size_t LDS_FLOATS = FRONT + (size_t)TILE + 36 + 4 + 2 * CS + 28;
  static_assert(ROWF % 256 == 0 && 4 * QW <= WsP - 8 && Ws <= 4 * 44 && Ws > 4 * 43 && Ws % 4 == 2, "row image");
  static_assert(S * QW <= 256, "one quad per lane");
};

// SQ (round 6, pgv_conv_up_sqerr): the squared-error criterion and the output activation's backward in this epilogue - the
// kernel has the output in registers, so the criterion's pass over it (read output + target, write gradient: 275 MB, 53 us)
// becomes one more 92 MB stream in and one out of a kernel that waits for its multiply phase: 73 + 53 -> 96 us.  Per element
// exactly sqerr_act_bwd_cls_kernel (bn.hip) with an upstream gradient of 1: g = act'(o) 2 scale (o - target); by-products: the
// sums of g by (row parity, column parity) class - an output quad's lanes ARE the four classes - the bias gradient (their
// total) and the criterion's value, one set of atomics per workgroup.
template <int CS, int H, int W, int S, bool SQ = false>
__global__ __launch_bounds__(256, 2) void up_c1_ring_kernel(int B, int nseg, const float* __restrict__ small_in,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int act, float slope,
                                                            float* __restrict__ out, pgv_bn_src bn, pgv_ring_sq sq) {
  using G = UpRingCfg<CS, H, W, S>;
  constexpr int Hs = G::Hs, Ws = G::Ws, Hg = G::Hg, WsP = G::WsP, ROWF = G::ROWF, NI = G::NI, NR = G::NR, QW = G::QW;
  constexpr int STEPS = G::STEPS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds + G::FRONT;
  float* Tt = tile + G::TILE;            // [3][3][ph][pw]
  float* aff = Tt + 40;                  // [2 CS]
  float* wsh = aff + 2 * CS;             // [25] sum_cs shift[cs] * w[cs][k]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float sq_c4[4] = {0.f, 0.f, 0.f, 0.f}, sq_err = 0.f;   // (SQ) sums of g by class [2 row parity + column parity]; sum of d^2
  if (tid < G::FRONT) lds[tid] = 0.f;
  if (tid < CS) {
    float sc = 1.f, sh = 0.f;
    if (in_scale) {
      // (pgv_conv_up_bn: the producer's BatchNorm is finalized here, from its statistics, instead of by a launch of its own)
      if (bn.stats)
        pgv_bn_finalize_dev(bn, CS, tid, blockIdx.x == 0, sc, sh);
      else
        sc = in_scale[tid], sh = in_shift[tid];
    }
    aff[tid] = sc;
    aff[CS + tid] = sh;
  }
  __syncthreads();
  if (tid < KK5) {
    float s = 0.f;
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) s = fmaf(aff[CS + cs], w[cs * KK5 + tid], s);
    wsh[tid] = s;
  }
  __syncthreads();
  if (tid < 36) {
    // row case rc: 0 = first grid row (th = 2 reads row -1: padding), 1 = interior, 2 = last (th = 0 reads row Hs);
    // column case the same with tw
    const int rc = tid / 12, cc = (tid / 4) % 3, ph = (tid >> 1) & 1, pw = tid & 1;
    float s = 0.f;
    for (int th = 0; th < 3; ++th)
      for (int tw = 0; tw < 3; ++tw) {
        const bool rok = !(rc == 0 && th == 2) && !(rc == 2 && th == 0) && ph + 2 * th < K5;
        const bool cok = !(cc == 0 && tw == 2) && !(cc == 2 && tw == 0) && pw + 2 * tw < K5;
        if (rok && cok) s += wsh[(ph + 2 * th) * K5 + pw + 2 * tw];
      }
    Tt[tid] = s;
  }
  const pgv_act_params actp = pgv_act_setup(act, slope);
  const float bv = bias ? bias[0] : 0.f;
  // this lane's DMA chunk offsets: chunk c = 64 i + lane of a row slot = (channel c / 48, 16-byte piece c % 48)
  int voff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = 64 * i + lane, cs = c / 48, q = c - cs * 48;
    voff[i] = q < 44 ? (cs * Hs * Ws + 4 * q) * 4 : 0x7FFFFFF0;
  }
  // quad of this lane inside a step
  const bool okl = tid < S * QW;
  const int rq = okl ? tid / QW : 0, vq = okl ? tid - rq * QW : 0, v0 = 4 * vq;
  const int seg_steps = (STEPS + nseg - 1) / nseg;
  const int units = B * nseg;
  const int bid = pgv_xcd_block();
  for (int i = tid; i < G::TILE / 4; i += 256) *reinterpret_cast<f32x4*>(tile + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int un = bid; un < units; un += gridDim.x) {
    const int b = un / nseg, seg = un - b * nseg;
    const int st0 = seg * seg_steps, st1 = min(STEPS, st0 + seg_steps);
    if (st0 >= st1) continue;
    const float* sample = small_in + (int64_t)b * CS * Hs * Ws;
    // rows [r0, r1) of the sample into their ring slots, row j of the range by wave j % 4; every wave clears the two
    // floats behind the end of the rows it fetched once they have landed (fix_rows)
    auto issue_rows = [&](int r0, int r1) {
      for (int r = r0 + wave; r < r1; r += 4) {
        float* slot = tile + ((r + NR) % NR) * ROWF;
        if ((unsigned)r < (unsigned)Hs) {
          dma_row<NI>(make_rsrc(sample + r * Ws, (CS * Hs - r) * Ws * 4), slot, voff);
        } else {   // a row of the zero padding above / below the image
#pragma unroll
          for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(slot + 256 * i + 4 * lane) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    auto fix_rows = [&](int r0, int r1) {
      for (int r = r0 + wave; r < r1; r += 4) {
        const int slot = (r + NR) % NR;
        if (lane < CS) *reinterpret_cast<f32x2*>(tile + slot * ROWF + lane * WsP + Ws) = f32x2{0.f, 0.f};
      }
    };
    ring_barrier();   // (the previous unit's reads of the ring are complete)
    const int ub = st0 * S;
    issue_rows(ub - 1, ub + S + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fix_rows(ub - 1, ub + S + 1);
    ring_barrier();
#pragma unroll 1
    for (int st = st0; st < st1; ++st) {
      const int u0 = st * S, Rb = min(S, Hg - u0);
      if (st + 1 < st1) issue_rows(u0 + S + 1, u0 + 2 * S + 1);
      const int u = u0 + rq;
      const bool okq = okl && rq < Rb;
      // shift-term constants of the quad's four grid columns (column case 1 except at the two ends of a row)
      const int rc = u == 0 ? 0 : (u == Hs - 1 ? 2 : 1);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(Tt + rc * 12 + 4);
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(Tt + rc * 12);
      const f32x4 t2 = *reinterpret_cast<const f32x4*>(Tt + rc * 12 + 8);
      f32x2 acc[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = v0 + j;
        const f32x4 t = v == 0 ? t0 : (v == Ws - 1 ? t2 : t1);
        acc[0][j] = f32x2{bv + t.x, bv + t.y};
        acc[1][j] = f32x2{bv + t.z, bv + t.w};
      }
      f32x2 acc4[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // kernel column 4: (ph 0, ph 1)
      // ring slots of the three input rows u+1-th
      int srow[3];
#pragma unroll
      for (int th = 0; th < 3; ++th) srow[th] = ((u + 1 - th + NR) % NR) * ROWF + v0;
#pragma unroll 1
      for (int cs = 0; cs < CS; ++cs) {
        // the 25 weights of the channel: uniform addresses of a read-only tensor = scalar loads into SGPRs, which the FMAs
        // take as operands (as an LDS table they were 7 of the 16 ds_read_b128 per channel and quad, and the LDS pipe -
        // one per CU - was the busiest unit of the multiply phase); the producer's BatchNorm scale goes onto the 18 inputs
        float wc[KK5];
#pragma unroll
        for (int i = 0; i < KK5; ++i) wc[i] = w[cs * KK5 + i];
        const float scv = aff[cs];
#pragma unroll
        for (int th = 0; th < 3; ++th) {
          const float* row = tile + srow[th] + cs * WsP;
          const float2 xm = *reinterpret_cast<const float2*>(row - 2);
          const f32x4 xc = *reinterpret_cast<const f32x4*>(row);
          const float2 xp = *reinterpret_cast<const float2*>(row + 4);
          const float x[6] = {xm.y * scv, xc.x * scv, xc.y * scv, xc.z * scv, xc.w * scv, xp.x * scv};  // columns v0-1 .. v0+4
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
              const float xin = x[j + 2 - tw];  // column v0+j+1-tw
              const f32x2 xx = {xin, xin};
              if (tw < 2) {
                // kernel columns kw = 2 tw (pw = 0) and 2 tw + 1 (pw = 1): one packed FMA
                const f32x2 w0 = {wc[(2 * th) * K5 + 2 * tw], wc[(2 * th) * K5 + 2 * tw + 1]};
                acc[0][j] = __builtin_elementwise_fma(xx, w0, acc[0][j]);
                if (th < 2) {
                  const f32x2 w1 = {wc[(2 * th + 1) * K5 + 2 * tw], wc[(2 * th + 1) * K5 + 2 * tw + 1]};
                  acc[1][j] = __builtin_elementwise_fma(xx, w1, acc[1][j]);
                }
              } else if (th < 2) {
                // kw = 4 exists for pw = 0 only: the two output rows (kh = 2 th, 2 th + 1) as the halves of one packed FMA
                // into accumulators of their own (a plain FMA with a scalar-register weight issues no faster than a packed one)
                const f32x2 w4 = {wc[(2 * th) * K5 + 4], wc[(2 * th + 1) * K5 + 4]};
                acc4[j] = __builtin_elementwise_fma(xx, w4, acc4[j]);
              } else {
                acc4[j].x = fmaf(xin, wc[4 * K5 + 4], acc4[j].x);
              }
            }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j].x += acc4[j].x, acc[1][j].x += acc4[j].y;
      // the rows of the next step have had the whole multiply phase to land
      if (st + 1 < st1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fix_rows(u0 + S + 1, u0 + 2 * S + 1);
      }
      if (okq) {
        float* ob = out + (int64_t)b * H * W;
        const int oh = 2 * u, ow = 2 * v0;
        const int n = W - ow;  // valid output columns from ow on
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          if (oh + ph < H) {
#ifdef PGV_RING_EXP_ALIGNED   // timing experiment (wrong addresses): what 16-byte aligned row starts would be worth
            float* o = out + ((int64_t)b * H * W) / 352 * 352 + (int64_t)min(oh + ph, 250) * 352 + ow;
#else
            float* o = ob + (int64_t)(oh + ph) * W + ow;
#endif
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = pgv_act_apply_nan(acc[ph][i >> 1][i & 1], actp);
            float gq[8];
            if constexpr (SQ) {
              const float* tg = sq.target + ((int64_t)b * H * W + (int64_t)(oh + ph) * W + ow);
              float t[8];
              if (n >= 8) {
                const f4u t0 = *reinterpret_cast<const f4u*>(tg), t1 = *reinterpret_cast<const f4u*>(tg + 4);
                t[0] = t0.x, t[1] = t0.y, t[2] = t0.z, t[3] = t0.w, t[4] = t1.x, t[5] = t1.y, t[6] = t1.z, t[7] = t1.w;
              } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = i < n ? tg[i] : 0.f;
              }
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float dlt = y[i] - t[i];
                float g = sq.k * dlt;
                if (act == PGV_ACT_LEAKY_RELU)
                  g = y[i] > 0.f ? g : slope * g;
                else if (act == PGV_ACT_HARDTANH)
                  g = (y[i] > -1.f && y[i] < 1.f) ? g : 0.f;
                const bool on = i < n;
                gq[i] = g;
                sq_err = on ? fmaf(dlt, dlt, sq_err) : sq_err;
                sq_c4[2 * ph + (i & 1)] += on ? g : 0.f;   // (oh and ow are even: the row parity is ph, the column parity i & 1)
              }
            }
            float* gr = SQ ? sq.gy + ((int64_t)b * H * W + (int64_t)(oh + ph) * W + ow) : nullptr;
            if (n >= 8) {
              f4u a, c;
              a.x = y[0], a.y = y[1], a.z = y[2], a.w = y[3];
              c.x = y[4], c.y = y[5], c.z = y[6], c.w = y[7];
              *reinterpret_cast<f4u*>(o) = a;
              *reinterpret_cast<f4u*>(o + 4) = c;
              if constexpr (SQ) {
                a.x = gq[0], a.y = gq[1], a.z = gq[2], a.w = gq[3];
                c.x = gq[4], c.y = gq[5], c.z = gq[6], c.w = gq[7];
                *reinterpret_cast<f4u*>(gr) = a;
                *reinterpret_cast<f4u*>(gr + 4) = c;
              }
            } else {
#pragma unroll
              for (int i = 0; i < 8; ++i)
                if (i < n) {
                  o[i] = y[i];
                  if constexpr (SQ) gr[i] = gq[i];
                }
            }
          }
        }
      }
      ring_barrier();
    }
  }
  if constexpr (SQ) {
    // the six sums of the workgroup (as sqerr_act_bwd_cls_kernel): 4 classes, their total = the bias gradient, the squared error
    __syncthreads();
    float* red6 = lds;   // (the ring is dead)
    const float v6[6] = {sq_c4[0], sq_c4[1], sq_c4[2], sq_c4[3], (sq_c4[0] + sq_c4[1]) + (sq_c4[2] + sq_c4[3]), sq_err};
    const float r = pgv_block_sums<6>(v6, red6);
    if (tid < 4) {
      if (sq.cls) atomicAdd(&sq.cls[(blockIdx.x & (PGV_CLS_COPIES - 1)) * 4 + tid], r);   // (the copy of this workgroup's XCD)
    } else if (tid == 4) {
      if (sq.gbias) atomicAdd(&sq.gbias[0], r);
    } else if (tid == 5) {
      if (sq.loss_acc) atomicAdd(sq.loss_acc, sq.scale * r);
    }
  }
}

}  // namespace

int pgv_conv_up_ring(const pgv_conv_desc* d, const float* small_in, const float* in_scale, const float* in_shift,
                     const float* w, const float* bias, int act, float slope, float* out, double* stats, hipStream_t st,
                     const pgv_bn_src* bn, const pgv_ring_sq* sq) {
  if (d->kh != 5 || d->kw != 5 || d->stride != 2 || d->pad != 2 || d->Cb != 1 || d->Cs != 8 || stats) return 0;
  if (d->Hb != 257 || d->Wb != 347 || d->B <= 0 || (d->flags & PGV_COMPUTE_BF16)) return 0;
  if (bn && !in_scale) return 0;
  if (((uintptr_t)small_in & 3) || (int64_t)d->B * 8 * 129 * 174 * 4 >= ((int64_t)1 << 40)) return 0;
  using G = UpRingCfg<8, 257, 347, 5>;
  constexpr size_t bytes = sizeof(float) * G::LDS_FLOATS;
  static_assert(bytes <= (size_t)kMaxLds / 2, "two workgroups per CU");
  auto kern = sq ? up_c1_ring_kernel<8, 257, 347, 5, true> : up_c1_ring_kernel<8, 257, 347, 5, false>;
  static bool raised[2] = {false, false};
  if (!raised[sq ? 1 : 0]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) {
      pgv_set_error("conv_up_ring: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      return PGV_E_LAUNCH;
    }
    raised[sq ? 1 : 0] = true;
  }
  // segments of a sample: enough workgroups for two per CU, no more segments than steps
  int nseg = (512 + d->B - 1) / d->B;
  nseg = nseg < 1 ? 1 : (nseg > G::STEPS ? G::STEPS : nseg);
  const int units = d->B * nseg;
  hipLaunchKernelGGL(kern, dim3(units < 512 ? units : 512), dim3(256), bytes, st, d->B, nseg, small_in, in_scale, in_shift,
                     w, bias, act, slope, out, bn ? *bn : pgv_no_bn(), sq ? *sq : pgv_ring_sq{});
  PGV_CHECK_LAUNCH("conv_up_ring");
  return 1;
}
