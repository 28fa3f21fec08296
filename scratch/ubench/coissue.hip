// How much does a VALU/LDS-heavy wave slow down when the other wave of its SIMD runs an MFMA loop (and vice versa)?
// Even workgroups run MFMA, odd workgroups run the VALU loop (2 workgroups per CU -> one wave of each kind per SIMD,
// if the dispatcher pairs wg i and i+256 on a CU; we launch [256 x kindA][256 x kindB]).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* stamps, int kind_first, int kind_second,
                                            int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  const int kind = blockIdx.x < 256 ? kind_first : kind_second;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  unsigned long long t0 = wall_clock64();
  if (kind == 0) {  // MFMA loop fed from LDS
    f32x4 acc[6];
    for (int n = 0; n < 6; ++n) acc[n] = f32x4{0, 0, 0, 0};
    const float a = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
      float b[6];
#pragma unroll
      for (int n = 0; n < 6; ++n) b[n] = lds[((it & 31) * 174 + n * 32 + lane) & 8191];
#pragma unroll
      for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[n], 0, 0, 0);
    }
    for (int n = 0; n < 6; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  } else if (kind == 1) {  // VALU loop: 24 dependent-free fma per iteration
    float x[24];
    for (int n = 0; n < 24; ++n) x[n] = (float)(lane + n);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < 24; ++n) x[n] = fmaf(x[n], 1.0001f, 0.5f);
    }
    for (int n = 0; n < 24; ++n) s += x[n];
  } else if (kind == 2) {  // LDS write/read loop (b128)
    f32x4 v = f32x4{1, 2, 3, 4};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(lds + 4 * (threadIdx.x + 256 * n)) = v;
#pragma unroll
      for (int n = 0; n < 2; ++n) v += *reinterpret_cast<f32x4*>(lds + 4 * (threadIdx.x + 256 * n));
    }
    s = v.x + v.y + v.z + v.w;
  } else {  // idle
  }
  unsigned long long t1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

int main() {
  float* out;
  unsigned long long* st;
  (void)hipMalloc(&out, 512 * 256 * 4);
  (void)hipMalloc(&st, 512 * 8);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const char* names[] = {"MFMA", "VALU", "LDS", "idle"};
  const int iters = 2000;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 4; ++b) {
      for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(512), dim3(256), 70 * 1024, 0, out, st, a, b, iters);
      (void)hipDeviceSynchronize();
      unsigned long long h[512];
      (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
      double ta = 0, tb = 0;
      for (int i = 0; i < 256; ++i) ta += h[i], tb += h[256 + i];
      printf("%-5s next to %-5s: %8.1f ns per iteration   (partner: %8.1f)\n", names[a], names[b], ta / 256 * 10 / iters,
             tb / 256 * 10 / iters);
    }
  return 0;
}
