// issue rate of v_pk_fma_f32 against v_fma_f32 (wave64, gfx950): N independent accumulator chains per wave, W waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b) {
  f32x2 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)threadIdx.x + i, 1.f};
  const f32x2 x = {a, a * 1.0001f}, y = {b, b * 0.999f};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
        else { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "v"(y.x));
               asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(x.y), "v"(y.y)); }
      }
  }
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  if (s == 1234.5f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}
int main() {
  float* out; (void)hipMalloc(&out, 64);
  for (int waves = 4; waves <= 16; waves *= 2)
    for (int pk = 0; pk < 2; ++pk) {
      const int iters = 20000;
      if (pk) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, iters, 1.0001f, 1e-6f);
      else hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, iters, 1.0001f, 1e-6f);
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) {
      if (pk) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, iters, 1.0001f, 1e-6f);
      else hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, iters, 1.0001f, 1e-6f); }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("   wall: %.3f ms per launch -> %.2f ns per instruction(-pair) per wave, %.1f TFLOP/s\n", ms / 10, ms / 10 * 1e6 / (iters * 32.0), 256.0 * waves * 64 * 4 * iters * 32.0 / (ms / 10 * 1e-3) / 1e12);
      float h[2]; (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
      // 64 FMA-pairs per iteration per lane
      printf("%s waves/CU %2d: %.2f clk per wave-instruction-equivalent (pk: per v_pk_fma; plain: per 2 v_fma) at %d waves/SIMD\n",
             pk ? "v_pk_fma_f32" : "2 x v_fma_f32", waves, h[1] / (iters * 32.0) / (waves / 4.0), waves / 4);
    }
  return 0;
}
