// Strided fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
// Serves the fully-connected layers of the VAE (nn.Linear: model/encoder.py:85, model/decoder.py:64) forward,
// input-gradient and weight-gradient products through strides, so no transposed copies are ever made.
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 waves, one 32x32 MFMA accumulator each), K staged
// through LDS 16 at a time with register prefetch of the next slab; split-K over blockIdx.z with float atomics
// onto a bias-initialised C (K = 24576/25024 with M = batch is the regime that matters here: the weight matrix
// is streamed exactly once from HBM).
#include "pgv_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 16, LDP = 65;  // LDP: padded LDS row (odd => conflict-free column writes)

__global__ void init_c_kernel(float* __restrict__ C, int M, int N, int64_t ldc, const float* __restrict__ bias_n) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N;
    const int n = (int)(i - m * N);
    C[m * ldc + n] = bias_n ? bias_n[n] : 0.f;
  }
}

// A_KFAST: A is k-contiguous (sak==1) -> loader lanes run along k; otherwise along m. Same for B with n/k.
template <bool A_KFAST, bool B_NFAST>
__global__ __launch_bounds__(256) void gemm_kernel(int M, int N, int K, const float* __restrict__ A, int64_t sam,
                                                   int64_t sak, const float* __restrict__ Bm, int64_t sbk,
                                                   int64_t sbn, float* __restrict__ C, int64_t ldc,
                                                   const float* __restrict__ bias_n, int k_per_split, int atomic,
                                                   int bf16) {
  __shared__ float As[BK][LDP];
  __shared__ float Bs[BK][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

  // loader coordinates: 4 elements of each operand per thread per slab
  int a_m[4], a_k[4], b_k[4], b_n[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * 256;  // 0..1023
    if (A_KFAST) {
      a_k[i] = e % BK;
      a_m[i] = e / BK;
    } else {
      a_m[i] = e % BM;
      a_k[i] = e / BM;
    }
    if (B_NFAST) {
      b_n[i] = e % BN;
      b_k[i] = e / BN;
    } else {
      b_k[i] = e % BK;
      b_n[i] = e / BK;
    }
  }
  float ra[4], rb[4];
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + a_m[i], k = k0 + a_k[i];
      ra[i] = (m < M && k < kend) ? A[(int64_t)m * sam + (int64_t)k * sak] : 0.f;
      const int n = n0 + b_n[i], kb = k0 + b_k[i];
      rb[i] = (n < N && kb < kend) ? Bm[(int64_t)kb * sbk + (int64_t)n * sbn] : 0.f;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  if (kbeg < kend) load_slab(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[a_k[i]][a_m[i]] = pgv_opnd(ra[i], bf16 != 0);  // PGV_COMPUTE_BF16: operand precision of the product
      Bs[b_k[i]][b_n[i]] = pgv_opnd(rb[i], bf16 != 0);
    }
    __syncthreads();
    if (k0 + BK < kend) load_slab(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[kk + (lane >> 5)][wm + (lane & 31)];
      const float b = Bs[kk + (lane >> 5)][wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int n = n0 + wn + (lane & 31);
  if (n < N) {
    const float bias = ((!atomic || (atomic == 2 && blockIdx.z == 0)) && bias_n) ? bias_n[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M) {
        if (atomic)
          atomicAdd(&C[(int64_t)m * ldc + n], acc[r] + bias);
        else
          C[(int64_t)m * ldc + n] = acc[r] + bias;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Fast path for the shapes the VAE actually runs (all extents multiples of the tile, 16-byte aligned rows): 16-byte
// global loads along whichever index is contiguous, K slabs of 32 double-buffered in LDS with register prefetch (one
// barrier per slab), v_mfma_f32_16x16x4_f32 with 4 column tiles per wave (64x64 tile per workgroup).  The MFMA k index
// is laid out so that lane group j = lane>>4 owns k = 16g + 4j .. +3: a k-contiguous operand row is one ds_read_b128
// per 16 k, an m/n-contiguous operand is staged k-major and read 4 bytes at a time.
//   A_K: A(m,k) k-contiguous (sak == 1), else m-contiguous (sam == 1);  B_K: B(k,n) k-contiguous (sbk == 1), else
//   n-contiguous (sbn == 1).  nn.Linear forward = (A_K, B_K), input gradient = (A_K, B_N), weight gradient = (A_M, B_N).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int HALF>
__device__ __forceinline__ void set_half(u32x4& dst, s16x4 v) {   // (conv_tile.h has the same helper for the conv kernels)
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 u = __builtin_bit_cast(u32x2, v);
  dst[2 * HALF] = u[0];
  dst[2 * HALF + 1] = u[1];
}
__device__ __forceinline__ s16x4 pack4(f32x4 v) {
  const bf16x2_t lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
  const unsigned u[2] = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
  return __builtin_bit_cast(s16x4, u);
}

constexpr int FK = 32;                 // K slab
constexpr int KROW = FK + 4;           // row stride of a k-contiguous tile [T][KROW]
// (m-contiguous A tile: [FK][TM + 16]; n-contiguous B tile: [FK][TN + 4])

// TM x TN tile per workgroup (64 or 128 each): four waves in 2 x 2, a wave owns a (TM/2) x (TN/2) quarter = NIM x NIN MFMA
// tiles.  128-wide tiles halve the reads of the operand they span (the z = 512 products moved 805 MB from L2 to LDS at
// 64 x 64 for 125 MB of operands); DEPTH slabs in flight in registers (4 at 64 x 64, 2 beyond: 8 floats4 per slab).
// TM = 256 ("batch-resident": all rows of a minibatch of 256 in one workgroup, the weight operand streamed once per
// K split): 512 threads, waves 4 x 2, the same 16 MFMA tiles per wave as 128 x 128 at 256 threads.
template <int TM>
constexpr int GemmThreads = TM >= 256 ? 512 : 256;

template <bool A_K, bool B_K, bool BF16, int TM, int TN, int DEPTH>
__global__ __launch_bounds__((GemmThreads<TM>), (TM >= 256 ? 2 : ((TM + TN > 128) ? 2 : 3))) void gemm_fast_kernel(
    int M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
    float* __restrict__ C, int64_t ldc, const float* __restrict__ bias_n, int k_per_split, int nsplit, int atomic) {
  constexpr int MROW_A = TM + 16, NROW_B = TN + 4;
  constexpr int A_FLOATS = A_K ? TM * KROW : FK * MROW_A;
  constexpr int B_FLOATS = B_K ? TN * KROW : FK * NROW_B;
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  constexpr int NTHR = GemmThreads<TM>, WROWS = NTHR / 128;   // waves as WROWS x 2
  constexpr int LA = TM * 8 / NTHR, LB = TN * 8 / NTHR;   // float4 per thread per slab and operand
  static_assert(LA >= 1 && LB >= 1, "loader");
  constexpr int WM = TM / WROWS, WN = TN / 2, NIM = WM / 16, NIN = WN / 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 * STAGE floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, j = lane >> 4;
  // Workgroup -> (m tile, n tile, K split).  Workgroups L, L + 8, L + 16, ... share an XCD and its L2 (round-robin
  // placement): the ones that read the same slice of the LARGE operand - all tiles of a K split, or the m tiles of an n
  // panel when K is not split - are given to one XCD, next to each other in time, so that slice is fetched from HBM once
  // and not once per tile (with n tiles fastest, as a plain 3-D grid has them: 100 MB of fabric reads for the 38 MB of
  // the encoder's Linear forward).
  const int Mt = M / TM, Nt = N / TN;
  int mt, nt, zs;
  {
    const int L = blockIdx.x, inner = nsplit > 1 ? Mt * Nt : Mt, outer = nsplit > 1 ? nsplit : Nt;
    int in_i, out_i;
    if ((outer & 7) == 0) {
      const int idx = L >> 3;
      in_i = idx % inner, out_i = (L & 7) + 8 * (idx / inner);
    } else {
      in_i = L % inner, out_i = L / inner;
    }
    if (nsplit > 1) mt = in_i / Nt, nt = in_i - mt * Nt, zs = out_i;
    else mt = in_i, nt = out_i, zs = 0;
  }
  const int m0 = mt * TM, n0 = nt * TN;
  const int kbeg = zs * k_per_split, kend = min(K, kbeg + k_per_split);

  // loaders: T * 8 float4 per operand per slab
  int64_t a_src[LA], b_src[LB];
  int a_dst[LA], b_dst[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int q = tid + NTHR * i;
    if (A_K) {  // A[m0 + r][k + 4f]: 8 float4 per row
      const int r = q >> 3, f = q & 7;
      a_src[i] = (int64_t)(m0 + r) * lda + 4 * f;
      a_dst[i] = r * KROW + 4 * f;
    } else {    // A stored [k][m]: A(m,k) = A[k*lda + m]; TM / 4 float4 per k row
      const int r = q / (TM / 4), f = q % (TM / 4);
      a_src[i] = (int64_t)r * lda + m0 + 4 * f;
      a_dst[i] = r * MROW_A + 4 * f;
    }
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int q = tid + NTHR * i;
    if (B_K) {  // B(k,n) = B[n*ldb + k]
      const int r = q >> 3, f = q & 7;
      b_src[i] = (int64_t)(n0 + r) * ldb + 4 * f;
      b_dst[i] = r * KROW + 4 * f;
    } else {    // B[k*ldb + n]
      const int r = q / (TN / 4), f = q % (TN / 4);
      b_src[i] = (int64_t)r * ldb + n0 + 4 * f;
      b_dst[i] = r * NROW_B + 4 * f;
    }
  }
  // DEPTH slabs in flight in registers ahead of the one being multiplied: these products are short-K or split-K with a
  // few slabs per workgroup, and one slab ahead left every workgroup waiting a memory latency per slab (all six Linear
  // products of the step took ~22 us whatever their size)
  f32x4 ra[DEPTH][LA], rb[DEPTH][LB];
  auto issue = [&](int slot, int k0) {
#pragma unroll
    for (int i = 0; i < LA; ++i)
      ra[slot][i] = *reinterpret_cast<const f32x4*>(A + a_src[i] + (A_K ? (int64_t)k0 : (int64_t)k0 * lda));
#pragma unroll
    for (int i = 0; i < LB; ++i)
      rb[slot][i] = *reinterpret_cast<const f32x4*>(Bm + b_src[i] + (B_K ? (int64_t)k0 : (int64_t)k0 * ldb));
  };
  auto commit = [&](int slot, float* st) {
#pragma unroll
    for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(st + a_dst[i]) = ra[slot][i];
#pragma unroll
    for (int i = 0; i < LB; ++i) *reinterpret_cast<f32x4*>(st + A_FLOATS + b_dst[i]) = rb[slot][i];
  };
  // a wave owns a quarter of the tile (NIM x NIN MFMA tiles): at 64 x 64, 8 LDS fragment reads per 16 k against 20 for a
  // 16 x 64 strip (the strip layout moved 80 KB of fragments per slab, 3/4 of the LDS rate at MFMA speed)
  const int wm = wave >> 1, wn = wave & 1;
  const int a_frag = A_K ? (wm * WM + m) * KROW + 4 * j : 4 * j * MROW_A + wm * WM + m;
  const int b_frag = B_K ? (wn * WN + m) * KROW + 4 * j : 4 * j * NROW_B + wn * WN + m;
  f32x4 acc[NIM * NIN];   // [im][in]
#pragma unroll
  for (int t = 0; t < NIM * NIN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kbeg < kend) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
      if (kbeg + u * FK < kend) issue(u, kbeg + u * FK);
    commit(0, lds);
    __syncthreads();
    int stage = 0;
    for (int kb = kbeg; kb < kend; kb += DEPTH * FK) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {   // slab kb + u*FK: its ring slot is u, free again once it sits in LDS
      const int k0 = kb + u * FK;
      if (k0 >= kend) break;
      const float* st = lds + stage * STAGE;
      const bool more = k0 + FK < kend;
      if (k0 + DEPTH * FK < kend) issue(u, k0 + DEPTH * FK);
      const float* ap = st + a_frag;
      const float* bp = st + A_FLOATS + b_frag;
      auto frags = [&](int g, f32x4 (&a)[NIM], f32x4 (&b)[NIN]) {
#pragma unroll
        for (int h = 0; h < NIM; ++h) {
          if (A_K) {
            a[h] = *reinterpret_cast<const f32x4*>(ap + 16 * h * KROW + 16 * g);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[h][e] = ap[(16 * g + e) * MROW_A + 16 * h];
          }
        }
#pragma unroll
        for (int h = 0; h < NIN; ++h) {
          if (B_K) {
            b[h] = *reinterpret_cast<const f32x4*>(bp + 16 * h * KROW + 16 * g);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) b[h][e] = bp[(16 * g + e) * NROW_B + 16 * h];
          }
        }
      };
      if constexpr (BF16) {
        // v_mfma_f32_16x16x32_bf16: the slab's two 16-deep halves in one instruction per tile (a lane's eight values = its
        // four of the first half, then its four of the second: both operands are built the same way)
        static_assert(FK == 32, "one K = 32 MFMA per tile and slab");
        f32x4 a0[NIM], b0[NIN], a1[NIM], b1[NIN];
        frags(0, a0, b0);
        frags(1, a1, b1);
        u32x4 a8[NIM], b8[NIN];
#pragma unroll
        for (int h = 0; h < NIM; ++h) set_half<0>(a8[h], pack4(a0[h])), set_half<1>(a8[h], pack4(a1[h]));
#pragma unroll
        for (int h = 0; h < NIN; ++h) set_half<0>(b8[h], pack4(b0[h])), set_half<1>(b8[h], pack4(b1[h]));
#pragma unroll
        for (int t = 0; t < NIM * NIN; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a8[t / NIN]),
                                                           __builtin_bit_cast(bf16x8_t, b8[t % NIN]), acc[t], 0, 0, 0);
      } else {
#pragma unroll
        for (int g = 0; g < FK / 16; ++g) {
          f32x4 a[NIM], b[NIN];
          frags(g, a, b);
          // round-robin over the accumulators: v_mfma_f32_16x16x4_f32 issues every 32 cycles but its result feeds a
          // dependent one only after 40
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NIM * NIN; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t / NIN][e], b[t % NIN][e], acc[t], 0, 0, 0);
        }
      }
      if (more) commit((u + 1) % DEPTH, lds + (stage ^ 1) * STAGE);
      __syncthreads();
      stage ^= 1;
    }
    }
  }
  // acc[NIN im + in][i]: row WM wm + 16 im + 4j + i, column WN wn + 16 in + m of the tile.  Out through LDS (the staging
  // buffers are free: the K loop ended with a barrier) so that a wave instruction covers whole 128-byte lines of C:
  // 64 consecutive floats per atomic instruction (the split-K products spend a third of their time in these atomics:
  // with 4 half-lines per instruction, as the accumulator layout gives them, ~8 us; laid out 16 quarter-lines wide 35 us),
  // 16 bytes per lane for plain stores.
  constexpr int TROW = TN + 4;
  // (the launcher sizes the dynamic LDS as max(2 stages, this tile): 256-row tiles need more than their stages)
  float* tile = lds;
#pragma unroll
  for (int t = 0; t < NIM * NIN; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      tile[(WM * wm + 16 * (t / NIN) + 4 * j + i) * TROW + WN * wn + 16 * (t % NIN) + m] = acc[t][i];
  __syncthreads();
  // atomic == 2: C held zeros on entry (PGV_PREZEROED) - no clearing launch, the first K split brings the bias
  const bool with_bias = (!atomic || (atomic == 2 && zs == 0)) && bias_n;
  if (atomic) {
    constexpr int RPP = NTHR / TN;   // rows per pass
    const int col = tid % TN;
    const float bias = with_bias ? bias_n[n0 + col] : 0.f;
#pragma unroll 4
    for (int k = 0; k < TM / RPP; ++k) {
      const int row = tid / TN + RPP * k;
      atomicAdd(C + (int64_t)(m0 + row) * ldc + n0 + col, tile[row * TROW + col] + bias);
    }
  } else {
    constexpr int QPR = TN / 4, RPP = NTHR / QPR;   // 16-byte groups per row, rows per pass
    const int c4 = (tid % QPR) * 4;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (with_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bias[i] = bias_n[n0 + c4 + i];
    }
#pragma unroll 4
    for (int k = 0; k < TM / RPP; ++k) {
      const int row = tid / QPR + RPP * k;
      *reinterpret_cast<f32x4*>(C + (int64_t)(m0 + row) * ldc + n0 + c4) =
          *reinterpret_cast<const f32x4*>(tile + row * TROW + c4) + bias;
    }
  }
}

template <bool A_K, bool B_K, bool BF16, int TM, int TN>
int launch_tile(int nsplit, hipStream_t st, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                float* C, int64_t ldc, const float* bias_n, int k_per_split, int atomic) {
  constexpr int DEPTH = (TM + TN > 128) ? 2 : 4;
  constexpr int A_FLOATS = A_K ? TM * KROW : FK * (TM + 16), B_FLOATS = B_K ? TN * KROW : FK * (TN + 4);
  constexpr size_t stage_bytes = sizeof(float) * 2 * (A_FLOATS + B_FLOATS), tile_bytes = sizeof(float) * TM * (TN + 4);
  constexpr size_t bytes = stage_bytes > tile_bytes ? stage_bytes : tile_bytes;
  static_assert(bytes <= 160 * 1024, "LDS budget");
  auto kern = gemm_fast_kernel<A_K, B_K, BF16, TM, TN, DEPTH>;
  if (bytes > 48 * 1024) {   // (above the default dynamic-LDS limit: raised once per kernel)
    static bool raised = false;
    if (!raised) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        pgv_set_error("pgv_gemm: cannot raise the dynamic LDS limit");
        return PGV_E_LAUNCH;
      }
      raised = true;
    }
  }
  const dim3 grid((unsigned)((M / TM) * (N / TN) * nsplit));
  hipLaunchKernelGGL(kern, grid, dim3(GemmThreads<TM>), bytes, st, M, N, K, A, lda, B, ldb, C, ldc, bias_n, k_per_split, nsplit, atomic);
  return PGV_OK;
}

template <bool A_K, bool B_K>
int launch_fast(int tm, int tn, int nsplit, hipStream_t st, int bf16, int M, int N, int K, const float* A, int64_t lda,
                const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias_n, int k_per_split, int atomic) {
#define PGV_GT(BF, TMv, TNv) \
  return launch_tile<A_K, B_K, BF, TMv, TNv>(nsplit, st, M, N, K, A, lda, B, ldb, C, ldc, bias_n, k_per_split, atomic)
  if (bf16) {
    if (tm == 256 && tn == 128) PGV_GT(true, 256, 128);
    if (tm == 256) PGV_GT(true, 256, 64);
    if (tm == 128 && tn == 128) PGV_GT(true, 128, 128);
    if (tm == 128) PGV_GT(true, 128, 64);
    if (tn == 128) PGV_GT(true, 64, 128);
    PGV_GT(true, 64, 64);
  }
  if (tm == 128 && tn == 128) PGV_GT(false, 128, 128);
  if (tm == 128) PGV_GT(false, 128, 64);
  if (tn == 128) PGV_GT(false, 64, 128);
  PGV_GT(false, 64, 64);
#undef PGV_GT
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int init_c_launch(float* C, int M, int N, int64_t ldc, const float* bias_n, hipStream_t st) {
  hipLaunchKernelGGL(init_c_kernel, dim3((unsigned)min((int64_t)1024, pgv_cdiv((int64_t)M * N, 256))), dim3(256), 0, st, C,
                     M, N, ldc, bias_n);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace

// gemm_frag.hip: the LDS-free fragment-streaming kernels for the nn.Linear shapes (1 = launched, 0 = shape not covered)
int pgv_gemm_frag(int M, int N, int K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                  float* C, int64_t ldc, const float* bias_n, int flags, hipStream_t st,
                  int (*init_c)(float*, int, int, int64_t, const float*, hipStream_t));

static int g_gemm_tiles = 0;
extern "C" int pgv_dbg_set_gemm_tiles(int v) {
  g_gemm_tiles = v;
  return 0;
}

extern "C" {

int64_t pgv_gemm_workspace(int, int, int) { return 0; }

int pgv_gemm(int M, int N, int K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
             float* C, int64_t ldc, const float* bias_n, int flags, void* /*workspace*/, int64_t /*workspace_bytes*/,
             void* stream) {
  PGV_CHECK_ARG(M >= 0 && N > 0 && K >= 0 && A && B && C && ldc >= N, "pgv_gemm: bad argument");
  if (M == 0) return PGV_OK;
  hipStream_t st = pgv_stream(stream);
  // ---- nn.Linear shapes, fp32: fragments streamed straight from global memory, no LDS tiles (gemm_frag.hip)
  if (pgv_kernel_policy() == 0 && M > 0 && K > 0) {
    const int rc = pgv_gemm_frag(M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias_n, flags, st, init_c_launch);
    if (rc < 0) return rc;
    if (rc == 1) {
      PGV_CHECK_LAUNCH("gemm_frag");
      return PGV_OK;
    }
  }
  // ---- fast path: tile-aligned extents, one unit stride per operand, 16-byte aligned rows
  {
    const bool a_k = sak == 1, a_m = sam == 1 && !a_k, b_k = sbk == 1, b_n = sbn == 1 && !b_k;
    const int64_t lda = a_k ? sam : sak, ldb = b_k ? sbn : sbk;
    if (pgv_kernel_policy() == 0 && (a_k || a_m) && (b_k || b_n) && M % 64 == 0 && N % 64 == 0 && K % FK == 0 && K > 0 && lda % 4 == 0 &&
        ldb % 4 == 0 && ldc % 4 == 0 && aligned16(A) && aligned16(B) && aligned16(C)) {
      // tile edges: 128 where the extent allows it and enough tiles remain to fill the chip (with the K splits)
      // 128-wide tile edges where the extent divides: in bf16 operand mode only - there the products are bound by the
      // operand traffic between L2 and LDS, which a 128-wide tile halves (z = 512, six products of the step: 405 -> 348 us;
      // z = 64: 125 -> 121); on the fp32 matrix pipe they are MFMA-bound and the larger workgroups only lose to wave
      // quantisation (z = 512: 625 -> 670 us, z = 64: no change)
      const bool big_ok = (flags & PGV_COMPUTE_BF16) != 0;
      // 256-row tiles (one 512-thread workgroup per CU holds every row of a 256-row minibatch, so that the weight operand
      // passes L2 -> LDS once per K split): measured SLOWER than two 128 x 128 workgroups per CU - six z = 512 products
      // 385-393 us against 320-335 (same box, alternating) - one workgroup per CU leaves nobody to cover its barriers.
      // Kept behind pgv_dbg_set_gemm_tiles(1) for A/B timing and covered by test_linear_gemm_bf16_operand_mode.
      int tm = (big_ok && M % 128 == 0 && M >= 256) ? 128 : 64, tn = (big_ok && N % 128 == 0 && N >= 256) ? 128 : 64;
      if (big_ok && M % 256 == 0 && (g_gemm_tiles & 1)) tm = 256;
      const int tiles = (M / tm) * (N / tn);
      const int target = tm == 256 ? 256 : ((tm + tn > 128) ? 512 : 768);   // workgroups that fit at once (1 / 2 / 3 per CU)
      // K splits only while the tiles alone leave at least half of the workgroup slots empty: 390 tiles of the z = 512
      // input gradient [256 x 25 024] were split in two (ceil(512 / 390)), i.e. a 25.6 MB output initialised by a launch of
      // its own and then written twice with float atomics
      int splits = (int)max((int64_t)1, min((int64_t)target / tiles, (int64_t)K / (FK * 4)));
      int k_per_split_v = (int)(pgv_cdiv(pgv_cdiv(K, splits), FK) * FK);
      splits = (int)pgv_cdiv(K, k_per_split_v);
      if (splits > 8 && (splits & 7)) {   // a multiple of 8 splits: one XCD per K slice (see the kernel)
        const int s8 = splits & ~7, kp = (int)(pgv_cdiv(pgv_cdiv(K, s8), FK) * FK);
        if (pgv_cdiv(K, kp) == s8) splits = s8, k_per_split_v = kp;
      }
      const int k_per_split = k_per_split_v;
      const int atomic = splits > 1 ? ((flags & PGV_PREZEROED) ? 2 : 1) : 0;
      if (atomic == 1) {
        hipLaunchKernelGGL(init_c_kernel, dim3((unsigned)min((int64_t)1024, pgv_cdiv((int64_t)M * N, 256))), dim3(256),
                           0, st, C, M, N, ldc, bias_n);
        PGV_CHECK_LAUNCH("gemm_init_c");
      }
      const int bf16 = (flags & PGV_COMPUTE_BF16) ? 1 : 0;
      int rc;
      if (a_k && b_k)
        rc = launch_fast<true, true>(tm, tn, splits, st, bf16, M, N, K, A, lda, B, ldb, C, ldc, bias_n, k_per_split, atomic);
      else if (a_k)
        rc = launch_fast<true, false>(tm, tn, splits, st, bf16, M, N, K, A, lda, B, ldb, C, ldc, bias_n, k_per_split, atomic);
      else if (b_k)
        rc = launch_fast<false, true>(tm, tn, splits, st, bf16, M, N, K, A, lda, B, ldb, C, ldc, bias_n, k_per_split, atomic);
      else
        rc = launch_fast<false, false>(tm, tn, splits, st, bf16, M, N, K, A, lda, B, ldb, C, ldc, bias_n, k_per_split, atomic);
      if (rc) return rc;
      PGV_CHECK_LAUNCH("gemm_fast");
      return PGV_OK;
    }
  }
  const int tiles = (int)(pgv_cdiv(M, BM) * pgv_cdiv(N, BN));
  // Split K until the grid holds ~2 workgroups per CU, keeping >= 8 slabs per split.
  int splits = 1;
  if (K > 0) splits = (int)max((int64_t)1, min(pgv_cdiv(512, tiles), pgv_cdiv(K, BK * 8)));
  int k_per_split = (int)(pgv_cdiv(pgv_cdiv(max(K, 1), splits), BK) * BK);
  splits = (int)pgv_cdiv(max(K, 1), k_per_split);
  const int atomic = splits > 1 ? ((flags & PGV_PREZEROED) && K > 0 ? 2 : 1) : 0;
  if (atomic == 1 || K == 0) {
    hipLaunchKernelGGL(init_c_kernel, dim3((unsigned)min((int64_t)1024, pgv_cdiv((int64_t)M * N, 256))), dim3(256), 0,
                       st, C, M, N, ldc, bias_n);
    PGV_CHECK_LAUNCH("gemm_init_c");
    if (K == 0) return PGV_OK;
  }
  dim3 grid((unsigned)pgv_cdiv(N, BN), (unsigned)pgv_cdiv(M, BM), (unsigned)splits);
  const bool akf = (sak == 1), bnf = (sbn == 1);
#define LAUNCH(AK, BNF)                                                                                         \
  hipLaunchKernelGGL((gemm_kernel<AK, BNF>), grid, dim3(256), 0, st, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, \
                     bias_n, k_per_split, atomic, (flags & PGV_COMPUTE_BF16) ? 1 : 0)
  if (akf && bnf)
    LAUNCH(true, true);
  else if (akf)
    LAUNCH(true, false);
  else if (bnf)
    LAUNCH(false, true);
  else
    LAUNCH(false, false);
#undef LAUNCH
  PGV_CHECK_LAUNCH("gemm");
  return PGV_OK;
}

}  // extern "C"
