// Range check of buffer_load_dwordx4 on a raw buffer (stride 0) on gfx950: is it per dword or per access?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, int nbytes, f32x4* out) {
  const uint64_t p = (uint64_t)src;
  i32x4 rsrc = {__builtin_amdgcn_readfirstlane((int)(uint32_t)p), __builtin_amdgcn_readfirstlane((int)(uint32_t)(p >> 32) & 0xFFFF),
                __builtin_amdgcn_readfirstlane(nbytes), 0x00020000};
  unsigned off = nbytes - 16 + 4 * threadIdx.x;  // lane 0 fully inside, lanes 1..3 partly, lane 4+ outside
  if (threadIdx.x == 7) off = 0xFFFFFFFFu;
  if (threadIdx.x == 8) off = 0xFFFFFFF0u;
  if (threadIdx.x == 9) off = 6;  // unaligned (not a multiple of 4? 6 is 2-byte aligned) - expect truncation or fine
  if (threadIdx.x == 10) off = 20;  // 4-byte aligned, not 16
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(rsrc) : "memory");
  out[threadIdx.x] = v;
}
int main() {
  float* src; f32x4* out;
  const int n = 64;
  (void)hipMalloc(&src, n * 4 + 64); (void)hipMalloc(&out, 64 * 16);
  float h[n + 16];
  for (int i = 0; i < n + 16; ++i) h[i] = i + 1;
  (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, n * 4, out);
  f32x4 o[64];
  (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  for (int i = 0; i < 11; ++i) printf("lane %d: %g %g %g %g\n", i, o[i].x, o[i].y, o[i].z, o[i].w);
  return 0;
}
