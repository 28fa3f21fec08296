// Streaming-read bandwidth of 16-byte-per-lane global loads at 16-byte aligned vs 4-byte aligned addresses, and of
// the band-tile pattern (8 channel planes x 10 rows of 174 floats per workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int U>
__global__ __launch_bounds__(256) void stream(const float* __restrict__ src, float* out, size_t n4, int off) {
  // each workgroup reads contiguous 256*U*16-byte blocks
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i0 = (size_t)blockIdx.x * 256 * U + threadIdx.x; i0 + 256 * (U - 1) < n4; i0 += stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f4u t = *reinterpret_cast<const f4u*>(src + 4 * (i0 + 256 * u) + off);
      v[u] = f32x4{t.x, t.y, t.z, t.w};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (s == 1234.5f) out[0] = s;
}

template <int U>
void run(const char* name, const float* src, float* out, size_t n, int off, int wg_per_cu) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t n4 = n / 4 - 1;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(stream<U>, dim3(256 * wg_per_cu), dim3(256), 0, 0, src, out, n4, off);
  (void)hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(stream<U>, dim3(256 * wg_per_cu), dim3(256), 0, 0, src, out, n4, off);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s off %d U %2d wg/CU %d: %.2f TB/s\n", name, off, U, wg_per_cu, n * 4.0 * reps / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t n = (size_t)1 << 28;  // 1 GiB of floats: far beyond the 256 MB infinity cache
  float *src, *out;
  (void)hipMalloc(&src, n * 4 + 64);
  (void)hipMalloc(&out, 64);
  (void)hipMemset(src, 0, n * 4 + 64);
  for (int off = 0; off < 4; off += 1) {
    run<4>("stream dwordx4", src, out, n, off, 2);
    run<4>("stream dwordx4", src, out, n, off, 8);
    run<8>("stream dwordx4", src, out, n, off, 2);
    run<16>("stream dwordx4", src, out, n, off, 2);
    run<16>("stream dwordx4", src, out, n, off, 8);
  }
  return 0;
}
