// Can the matrix pipe and the vector ALU of a SIMD work at the same time?  512 threads per workgroup, one workgroup per CU
// (two waves per SIMD, as in the large-plane kernels).  Streams per wave:
//   M: NM v_mfma_f32_16x16x32_bf16 on 4 independent accumulators;  V: NV v_fma_f32 on 8 independent registers;
//   I: the two interleaved in one wave, 1 matrix instruction : R vector instructions;
//   S: waves 0-3 (one per SIMD) run M, waves 4-7 run V  (different waves of a SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int R>
__global__ __launch_bounds__(512) void k(float* out, long long* ticks, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float c1 = 1.0001f, c2 = 0.0001f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  __syncthreads();
  const long long t0 = clock64();
  if (MODE == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q & 3], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) v[(q * R + r) & 7] = __builtin_fmaf(v[(q * R + r) & 7], c1, c2);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, R, 0);
      }
    }
  } else {
    if (do_m)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q & 3], 0, 0, 0);
      }
    if (do_v)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8 * R; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], c1, c2);
      }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ticks[wave] = t1 - t0;
}

template <int MODE, int R>
void run(const char* name, float* out, long long* ticks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, R><<<256, 512>>>(out, ticks, iters);
  hipEventRecord(e0);
  k<MODE, R><<<256, 512>>>(out, ticks, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double nm = 8.0 * iters, nv = 8.0 * R * iters;
  printf("%-44s R=%d  %7.1f us   wave0 %8lld ticks  wave4 %8lld ticks   per matrix instr (wave0) %.1f ticks, per vector instr (wave4) %.2f ticks\n",
         name, R, ms * 1e3, h[0], h[4], h[0] / nm, h[4] / nv);
}

int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 64);
  const int iters = 2000;
  run<0, 1>("M: matrix only, 2 waves / SIMD", out, ticks, iters);
  run<1, 2>("V: vector only, 2 waves / SIMD", out, ticks, iters);
  run<1, 4>("V: vector only, 2 waves / SIMD", out, ticks, iters);
  run<2, 1>("I: interleaved in every wave", out, ticks, iters);
  run<2, 2>("I: interleaved in every wave", out, ticks, iters);
  run<2, 4>("I: interleaved in every wave", out, ticks, iters);
  run<3, 1>("S: matrix waves + vector waves on a SIMD", out, ticks, iters);
  run<3, 2>("S: matrix waves + vector waves on a SIMD", out, ticks, iters);
  run<3, 4>("S: matrix waves + vector waves on a SIMD", out, ticks, iters);
  return 0;
}
