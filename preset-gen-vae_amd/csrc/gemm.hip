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
    const float bias = (!atomic && bias_n) ? bias_n[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M) {
        if (atomic)
          atomicAdd(&C[(int64_t)m * ldc + n], acc[r]);
        else
          C[(int64_t)m * ldc + n] = acc[r] + bias;
      }
    }
  }
}

}  // namespace

extern "C" {

int64_t pgv_gemm_workspace(int, int, int) { return 0; }

int pgv_gemm(int M, int N, int K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
             float* C, int64_t ldc, const float* bias_n, int flags, void* /*workspace*/, int64_t /*workspace_bytes*/,
             void* stream) {
  PGV_CHECK_ARG(M >= 0 && N > 0 && K >= 0 && A && B && C && ldc >= N, "pgv_gemm: bad argument");
  if (M == 0) return PGV_OK;
  hipStream_t st = pgv_stream(stream);
  const int tiles = (int)(pgv_cdiv(M, BM) * pgv_cdiv(N, BN));
  // Split K until the grid holds ~2 workgroups per CU, keeping >= 8 slabs per split.
  int splits = 1;
  if (K > 0) splits = (int)max((int64_t)1, min(pgv_cdiv(512, tiles), pgv_cdiv(K, BK * 8)));
  int k_per_split = (int)(pgv_cdiv(pgv_cdiv(max(K, 1), splits), BK) * BK);
  splits = (int)pgv_cdiv(max(K, 1), k_per_split);
  const int atomic = splits > 1;
  if (atomic || K == 0) {
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
