// An fp32 product as SIX bf16 matrix instructions: accuracy and instruction rate against v_mfma_f32_16x16x4_f32.
//   x = x1 + x2 + x3 exactly, x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)   (3 x 8 significant bits = 24)
//   x y ~= x1 y1 + x1 y2 + x2 y1 + x2 y2 + x1 y3 + x3 y1      (the dropped terms are below 2^-23 |x y|)
// Part 1: one wave per 16x16 output tile of C = A B^T (A [M][K], B [N][K], K-contiguous, fp32), K = 4096: the same tile by
//         16x16x4_f32, by the six-instruction split and by ONE bf16 instruction (operands rounded), against float64.
// Part 2: instruction streams only (registers as operands), per K = 32 and tile: 8 fp32 instructions against 6 bf16 ones.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

template <int MODE>   // 0: fp32 instruction, 1: six bf16 instructions, 2: one bf16 instruction on rounded operands
__global__ __launch_bounds__(64) void gemm_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                int N, int K) {
  const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) {
      a[j] = A[(size_t)(m0 + i) * K + k0 + 8 * kq + j];
      b[j] = B[(size_t)(n0 + i) * K + k0 + 8 * kq + j];
    }
    if (MODE == 0) {
      // (any assignment of k to (lane group, step) is a valid contraction as long as both operands use the same one)
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    } else {
      bf8 a1, a2, a3, b1, b2, b3;
      for (int j = 0; j < 8; ++j) {
        __bf16 p, q, r;
        split3(a[j], p, q, r);
        a1[j] = p, a2[j] = q, a3[j] = r;
        split3(b[j], p, q, r);
        b1[j] = p, b2[j] = q, b3[j] = r;
      }
      if (MODE == 1) {
        // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 4; ++r) C[(size_t)(m0 + 4 * kq + r) * N + n0 + i] = acc[r];
}

template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1.f + lane, b = 0.5f * lane;
  bf8 pa, pb;
  for (int j = 0; j < 8; ++j) pa[j] = (__bf16)(a + j), pb[j] = (__bf16)(b - j);
  for (int it = 0; it < iters; ++it) {   // one iteration = K = 32 on eight tiles
#pragma unroll
    for (int rep = 0; rep < (MODE == 0 ? 8 : 6); ++rep)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (MODE == 0)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        else
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[t], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.f) out[0] = s;
}

int main() {
  const int M = 64, N = 64, K = 4096;
  std::vector<float> A((size_t)M * K), B((size_t)N * K);
  srand(7);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f + 0.25f;    // not centred: the sums do not cancel
  for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f + 0.01f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<double> ref((size_t)M * N);
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
      ref[(size_t)m * N + n] = s;
    }
  const char* names[3] = {"v_mfma_f32_16x16x4_f32      ", "six v_mfma_f32_16x16x32_bf16", "one bf16 instruction        "};
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 0) hipLaunchKernelGGL(gemm_tile<0>, dim3(N / 16, M / 16), dim3(64), 0, 0, dA, dB, dC, N, K);
    if (mode == 1) hipLaunchKernelGGL(gemm_tile<1>, dim3(N / 16, M / 16), dim3(64), 0, 0, dA, dB, dC, N, K);
    if (mode == 2) hipLaunchKernelGGL(gemm_tile<2>, dim3(N / 16, M / 16), dim3(64), 0, 0, dA, dB, dC, N, K);
    std::vector<float> C((size_t)M * N);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0, mx = 0;
    for (size_t i = 0; i < C.size(); ++i) {
      const double e = C[i] - ref[i];
      num += e * e, den += ref[i] * ref[i];
      mx = fmax(mx, fabs(e) / fabs(ref[i]));
    }
    printf("%s K=%d: rel L2 error %.3e, max rel error %.3e\n", names[mode], K, sqrt(num / den), mx);
  }
  float* dout;
  hipMalloc(&dout, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 4000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256 * 2), dim3(256), 0, 0, dout, iters);
      else hipLaunchKernelGGL(rate<1>, dim3(256 * 2), dim3(256), 0, 0, dout, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double macs = 512.0 * 4 * iters * 8 * 16 * 16 * 32;   // workgroups x waves x iterations x tiles x (16 x 16 x 32)
    printf("%s: %.3f ms for %.1f GMAC of fp32-equivalent products = %.1f TFLOP/s (fp32-equivalent)\n", names[mode], ms, macs / 1e9,
           2 * macs / ms / 1e9);
  }
  return 0;
}
