// Streaming-write bandwidth by store width and alignment: dwordx4 at 16 / 8 / 4-byte alignment, dwordx2 at 8-byte
// alignment, dword.  Each lane owns 16 contiguous bytes per iteration; a wave covers 1 KB contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
template <int MODE>
__global__ __launch_bounds__(256) void wr(float* __restrict__ dst, size_t n4, int off) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float* p = dst + 4 * i + off;
    const float v = (float)i;
    if (MODE == 0) { *reinterpret_cast<f4u*>(p) = f4u{1.f, 2.f, 3.f, v}; }
    if (MODE == 1) { *reinterpret_cast<f2u*>(p) = f2u{1.f, v}; *reinterpret_cast<f2u*>(p + 2) = f2u{2.f, v}; }
    if (MODE == 2) { p[0] = v; p[1] = 1.f; p[2] = 2.f; p[3] = 3.f; }
    if (MODE == 3) {   // lane-contiguous dwordx2: instruction A covers 512 B, instruction B the next 512 B of the wave's 1 KB
      const size_t w0 = (i & ~(size_t)63) * 4, l = i & 63;
      *reinterpret_cast<f2u*>(dst + w0 + 2 * l + off) = f2u{1.f, v};
      *reinterpret_cast<f2u*>(dst + w0 + 128 + 2 * l + off) = f2u{2.f, v};
    }
  }
}
template <int MODE> void run(const char* name, float* dst, size_t n, int off) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t n4 = n / 4 - 64;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(wr<MODE>, dim3(2048), dim3(256), 0, 0, dst, n4, off);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(wr<MODE>, dim3(2048), dim3(256), 0, 0, dst, n4, off);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s off %d floats: %.2f TB/s\n", name, off, n * 4.0 * 5 / (ms * 1e-3) / 1e12);
}
int main() {
  const size_t n = (size_t)1 << 28;
  float* dst; (void)hipMalloc(&dst, n * 4 + 4096);
  for (int off = 0; off < 4; ++off) run<0>("dwordx4", dst, n, off);
  for (int off = 0; off < 4; off += 2) run<1>("2 x dwordx2 per lane (16 B per lane)", dst, n, off);
  for (int off = 0; off < 4; off += 2) run<3>("2 x dwordx2, lane-contiguous 512 B each", dst, n, off);
  run<2>("4 x dword per lane", dst, n, 0);
  run<2>("4 x dword per lane", dst, n, 1);
  return 0;
}
