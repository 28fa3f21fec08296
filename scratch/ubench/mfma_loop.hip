// Micro-benchmark of candidate inner loops for the band conv kernels: one 16x16x4 f32 MFMA per (t, step), B operand
// read from an LDS tile with compile-time strides (immediate ds_read offsets), no masks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NT, int CK, int KS, int WP, int PLANE>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* stamps, const float* wsrc, int reps) {
  extern __shared__ float lds[];
  float* in_tile = lds;
  float* w_tile = lds + CK * PLANE;  // [CK*KS*4][17]
  for (int i = threadIdx.x; i < CK * PLANE; i += 256) in_tile[i] = (float)(i & 7);
  for (int i = threadIdx.x; i < CK * KS * 4 * 17; i += 256) w_tile[i] = wsrc[i & 1023];
  __syncthreads();
  f32x4 acc[NT];
  for (int n = 0; n < NT; ++n) acc[n] = f32x4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int offB[NT];
  for (int t = 0; t < NT; ++t) {
    const int p = (wave * NT + t) * 16 + (lane & 15);
    const int r = p / 88, c = p - r * 88;
    offB[t] = 2 * r * WP + 2 * c + (lane >> 4);
  }
  const int offA = (lane >> 4) * 17 + (lane & 15);
  unsigned long long t0 = wall_clock64(), c0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    constexpr int S = CK * KS;
    float a0, a1, b0[NT], b1[NT];
    auto load_step = [&](int st, float& av, float (&bv)[NT]) {
      const int c = st / KS, kh = st - c * KS;
      av = w_tile[st * 4 * 17 + offA];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = in_tile[offB[t] + c * PLANE + kh * WP];
    };
    auto compute_step = [&](const float& av, const float (&bv)[NT]) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
    };
    load_step(0, a0, b0);
#pragma unroll
    for (int st = 0; st < S; st += 2) {
      load_step(st + 1, a1, b1);
      if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
      compute_step(a0, b0);
      if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
      if (st + 2 < S) load_step(st + 2, a0, b0);
      if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
      compute_step(a1, b1);
      if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long c1 = clock64(), t1 = wall_clock64();
  float s = 0;
  for (int n = 0; n < NT; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = c1 - c0;
  }
}

template <int MODE, int NT, int CK, int KS, int WP, int PLANE>
void run(const char* name, int blocks_per_cu) {
  float *out, *w;
  unsigned long long* st;
  const int nb = 256 * blocks_per_cu;
  (void)hipMalloc(&out, nb * 256 * 4);
  (void)hipMalloc(&w, 4096);
  (void)hipMemset(w, 0, 4096);
  (void)hipMalloc(&st, nb * 16);
  const int reps = 8;
  auto kern = k<MODE, NT, CK, KS, WP, PLANE>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  size_t lds = 4 * (CK * PLANE + CK * KS * 4 * 17);
  if (blocks_per_cu == 1) lds = lds > 90 * 1024 ? lds : 90 * 1024;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, out, st, w, reps);
  (void)hipDeviceSynchronize();
  unsigned long long h[2];
  (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
  const double n_mfma = (double)reps * NT * CK * KS;
  printf("%-36s wg/CU %d lds %zu: %.2f ns/MFMA/wave, %.1f clk/MFMA/wave (hipErr %d)\n", name, blocks_per_cu, lds,
         h[0] * 10.0 / n_mfma, h[1] / n_mfma, (int)hipGetLastError());
  (void)hipFree(out);
  (void)hipFree(st);
  (void)hipFree(w);
}

int main() {
  run<0, 6, 8, 4, 178, 1784>("const strides, compiler sched", 1);
  run<0, 6, 8, 4, 178, 1784>("const strides, compiler sched", 2);
  run<1, 6, 8, 4, 178, 1784>("const strides, sched_barrier", 1);
  run<1, 6, 8, 4, 178, 1784>("const strides, sched_barrier", 2);
  return 0;
}
