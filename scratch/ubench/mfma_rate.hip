// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 from one wave per SIMD, alone and fed from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* stamps, int iters, int stride) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  int off[NACC];
  for (int n = 0; n < NACC; ++n) off[n] = (lane & 15) * 2 + (lane >> 4) + n * 32;
  float a = 1.0f + lane, b[NACC];
  for (int n = 0; n < NACC; ++n) b[n] = 2.0f;
  const bool ok = lane != 77;
  unsigned long long t0 = wall_clock64(), c0 = clock64();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[n], 0, 0, 0);
    }
  } else {
    float b0[NACC], b1[NACC];
    const float* p = lds;
#pragma unroll
    for (int n = 0; n < NACC; ++n) b0[n] = p[off[n]];
    for (int it = 0; it < iters; it += 2) {
      p = lds + ((it + 1) & 31) * stride;
#pragma unroll
      for (int n = 0; n < NACC; ++n) b1[n] = p[off[n]];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        float v = (MODE == 2) ? (ok ? b0[n] : 0.f) : b0[n];
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v, acc[n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      p = lds + ((it + 2) & 31) * stride;
#pragma unroll
      for (int n = 0; n < NACC; ++n) b0[n] = p[off[n]];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        float v = (MODE == 2) ? (ok ? b1[n] : 0.f) : b1[n];
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v, acc[n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long c1 = clock64(), t1 = wall_clock64();
  float s = 0;
  for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = c1 - c0;
  }
}

template <int MODE, int NACC>
void run(const char* name, int blocks_per_cu) {
  float* out;
  unsigned long long* st;
  const int nb = 256 * blocks_per_cu;
  hipMalloc(&out, nb * 256 * 4);
  hipMalloc(&st, nb * 16);
  const int iters = 512;
  auto kern = k<MODE, NACC>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : 70 * 1024;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, out, st, iters, 174);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * NACC;
  printf("%-28s wg/CU %d: %.1f ns wall, %.0f clk -> %.2f ns/MFMA, %.1f clk/MFMA, clock %.2f GHz (hipErr %d)\n", name,
         blocks_per_cu, h[0] * 10.0, (double)h[1], h[0] * 10.0 / n_mfma, h[1] / n_mfma, h[1] / (h[0] * 10.0),
         (int)hipGetLastError());
  hipFree(out);
  hipFree(st);
}

int main() {
  run<0, 6>("regs only, 6 acc", 1);
  run<0, 6>("regs only, 6 acc", 2);
  run<1, 6>("lds b32 reads, 6 acc", 1);
  run<1, 6>("lds b32 reads, 6 acc", 2);
  run<2, 6>("lds + cndmask, 6 acc", 1);
  run<2, 6>("lds + cndmask, 6 acc", 2);
  run<2, 3>("lds + cndmask, 3 acc", 1);
  run<2, 12>("lds + cndmask, 12 acc", 1);
  return 0;
}
