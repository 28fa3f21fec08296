// v_mfma_f32_16x16x32_bf16 throughput against the number of INDEPENDENT accumulation chains in flight on a SIMD:
// W waves per SIMD (256 * W threads per workgroup, one workgroup per CU) x C chains per wave, instructions of a wave issued
// round-robin over its chains (chain c's next instruction depends on its previous one).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int C>
__global__ void k(float* out, long long* ticks, int iters) {
  f32x4 acc[C];
  for (int c = 0; c < C; ++c) acc[c] = f32x4{0, 0, 0, 0};
  u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      acc[q % C] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q % C], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < C; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int C>
void run(int waves, float* out, long long* ticks) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<C><<<256, 256 * waves>>>(out, ticks, iters);
  hipEventRecord(e0);
  k<C><<<256, 256 * waves>>>(out, ticks, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  const double flops = 256.0 * 4 * waves * 8.0 * iters * 16384.0;
  printf("%d wave(s) / SIMD x %d chain(s): %6.1f ticks per instruction of a wave, %5.1f per instruction of the SIMD | %7.1f us = %6.0f TFLOP/s\n", waves, C,
         h / (8.0 * iters), h / (8.0 * iters * waves), ms * 1e3, flops / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 64);
  run<1>(1, out, ticks); run<2>(1, out, ticks); run<4>(1, out, ticks); run<8>(1, out, ticks);
  run<1>(2, out, ticks); run<2>(2, out, ticks); run<4>(2, out, ticks);
  return 0;
}
