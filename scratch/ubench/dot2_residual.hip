// Is x - bf16(x) through v_dot2c_f32_bf16 (one instruction on the packed pair, no unpacking shift / mask) bit-identical to
// the subtraction from the unpacked term?  All exponents, both halves of the pair, second-level residuals too.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack(float a, float b) {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  v2 r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return __builtin_bit_cast(unsigned, r);
}
__global__ void k(const float* x, int n, unsigned* bad, float* ex) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  const unsigned H = pack(a, b);
  const float ra = a - __builtin_bit_cast(float, H << 16), rb = b - __builtin_bit_cast(float, H & 0xffff0000u);
  // (in registers: as an inline constant the packed {-1.0, 0} is printed "-1.0" by the assembler and taken as the fp32
  // pattern 0xbf800000 = {0, -1.0} by the hardware)
  unsigned c_lo = 0x0000bf80u, c_hi = 0xbf800000u;
  asm volatile("" : "+v"(c_lo), "+v"(c_hi));
  const bf2 lo1 = __builtin_bit_cast(bf2, c_lo), hi1 = __builtin_bit_cast(bf2, c_hi);
  const float da = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, H), lo1, a, false);
  const float db = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, H), hi1, b, false);
  const unsigned M = pack(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, M << 16), sb = rb - __builtin_bit_cast(float, M & 0xffff0000u);
  const float ea = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, M), lo1, ra, false);
  const float eb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, M), hi1, rb, false);
  if (__builtin_bit_cast(unsigned, da) != __builtin_bit_cast(unsigned, ra) || __builtin_bit_cast(unsigned, db) != __builtin_bit_cast(unsigned, rb) ||
      __builtin_bit_cast(unsigned, ea) != __builtin_bit_cast(unsigned, sa) || __builtin_bit_cast(unsigned, eb) != __builtin_bit_cast(unsigned, sb)) {
    unsigned s = atomicAdd(bad, 1u);
    if (s < 8) { ex[4 * s] = a; ex[4 * s + 1] = ra; ex[4 * s + 2] = da; ex[4 * s + 3] = b; }
  }
}
int main() {
  const int n = 1 << 24;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    if (i % 3 == 0) { float f = (rand() / (float)RAND_MAX - 0.5f) * 8.f; memcpy(&u, &f, 4); }
    unsigned e = (u >> 23) & 255;
    if (e == 255) u &= ~(1u << 30);   // no inf / nan
    memcpy(&h[i], &u, 4);
  }
  float *x, *ex; unsigned* bad;
  hipMalloc(&x, n * 4); hipMalloc(&bad, 4); hipMalloc(&ex, 32 * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
  k<<<n / 2 / 256, 256>>>(x, n, bad, ex);
  unsigned hb; float hex_[32];
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hex_, ex, 32 * 4, hipMemcpyDeviceToHost);
  printf("pairs %d mismatching %u\n", n / 2, hb);
  for (unsigned s = 0; s < (hb < 8 ? hb : 8); ++s) printf("  a %.9g (%a) sub %a dot2 %a  b %a\n", hex_[4*s], hex_[4*s], hex_[4*s+1], hex_[4*s+2], hex_[4*s+3]);
  return 0;
}
