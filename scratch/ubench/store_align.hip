// Streaming-write bandwidth of 16-byte-per-lane global stores: 16-byte aligned vs 4-byte aligned addresses, one contiguous
// 1 KB per wave-instruction vs two instructions that each write 16 of every 32 bytes (the up_c1_v2 output pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int MODE>
__global__ __launch_bounds__(256) void wr(float* __restrict__ dst, size_t n8, int off) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    f4u a = {1.f, 2.f, 3.f, (float)i}, b = {4.f, 5.f, 6.f, (float)i};
    if (MODE == 0) {   // lane owns 32 contiguous bytes, two instructions (each writes 16 of every 32 bytes)
      *reinterpret_cast<f4u*>(dst + 8 * i + off) = a;
      *reinterpret_cast<f4u*>(dst + 8 * i + 4 + off) = b;
    } else {           // each instruction writes 1 KB contiguous per wave
      const size_t w0 = (i & ~(size_t)63) * 8, l = i & 63;
      *reinterpret_cast<f4u*>(dst + w0 + 4 * l + off) = a;
      *reinterpret_cast<f4u*>(dst + w0 + 256 + 4 * l + off) = b;
    }
  }
}
template <int MODE>
void run(const char* name, float* dst, size_t n, int off) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t n8 = n / 8 - 64;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(wr<MODE>, dim3(2048), dim3(256), 0, 0, dst, n8, off);
  (void)hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(wr<MODE>, dim3(2048), dim3(256), 0, 0, dst, n8, off);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s off %d: %.2f TB/s\n", name, off, n * 4.0 * reps / (ms * 1e-3) / 1e12);
}
int main() {
  const size_t n = (size_t)1 << 28;
  float* dst; (void)hipMalloc(&dst, n * 4 + 4096);
  for (int off = 0; off < 4; ++off) { run<0>("store 2x16 of 32 per lane", dst, n, off); run<1>("store 1 KB per instruction", dst, n, off); }
  return 0;
}
