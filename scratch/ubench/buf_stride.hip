// Range check of a STRUCTURED buffer (stride = row bytes) on gfx950: does a 16-byte load that runs over the end of its
// record return zeros for the dwords behind the record (offset >= stride)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* src, int rows, int stride_bytes, f32x4* out, int swz) {
  const uint64_t p = (uint64_t)src;
  i32x4 rsrc = {__builtin_amdgcn_readfirstlane((int)(uint32_t)p),
                __builtin_amdgcn_readfirstlane(((int)(uint32_t)(p >> 32) & 0xFFFF) | (stride_bytes << 16)),
                __builtin_amdgcn_readfirstlane(rows), __builtin_amdgcn_readfirstlane(0x00020000 | swz)};
  // lane l: row l / 4, offset: the last 16 bytes of the row shifted by (l % 4) * 4 bytes
  const unsigned row = threadIdx.x / 4, off = stride_bytes - 16 + (threadIdx.x % 4) * 4;
  u32x2 io = {row, off};
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(io), "s"(rsrc) : "memory");
  out[threadIdx.x] = v;
}
int main() {
  const int W = 10, rows = 6;  // row = 40 bytes
  float* src; f32x4* out;
  (void)hipMalloc(&src, W * rows * 4 + 64); (void)hipMalloc(&out, 64 * 16);
  float h[W * rows + 16];
  for (int i = 0; i < W * rows + 16; ++i) h[i] = i;
  (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  for (int swz = 0; swz < 2; ++swz) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, rows, W * 4, out, 0);
    f32x4 o[64];
    (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("row %d shift %d: %g %g %g %g\n", i / 4, i % 4, o[i].x, o[i].y, o[i].z, o[i].w);
    printf("row 6 (out of range): %g %g\n", o[24].x, o[24].y);
    break;
  }
  return 0;
}
