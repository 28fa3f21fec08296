// v_pk_fma_f32 operand forms at 2 waves per SIMD (512-thread blocks, one per CU): VGPR operands, an SGPR pair as the
// multiplier, op_sel_hi broadcast of the low half, and the kernel's mix (pk + fma with SGPR multipliers)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int F>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
  f32x2 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)threadIdx.x + i, 1.f};
  f32x2 x = {a, a * 1.0001f}, y = {b, b * 0.999f};
  f32x2 sy = {__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(b))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(b * 0.999f)))};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (F == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
        if (F == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "s"(sy));
        if (F == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(sy));
        if (F == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "v"(y));
        if (F == 4) asm volatile("v_fma_f32 %0, %2, %1, %0" : "+v"(acc[i].x) : "v"(x.x), "s"(sy.x));
        if (F == 5) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(x), "v"(y));
      }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  if (s == 1234.5f) out[0] = s;
}
template <int F> void run(const char* name, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<F>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 1e-6f);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<F>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 1e-6f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.2f ns per instruction per SIMD (2 waves)\n", name, ms / 5 * 1e6 / (iters * 32.0) / 2);
}
int main() {
  float* out; (void)hipMalloc(&out, 64);
  run<0>("v_pk_fma_f32 v, v, v", out);
  run<1>("v_pk_fma_f32 v, v, s[pair]", out);
  run<2>("v_pk_fma_f32 v, v, s[pair] op_sel_hi:[0,1,1]", out);
  run<3>("v_pk_fma_f32 v, v, v op_sel_hi:[0,1,1]", out);
  run<4>("v_fma_f32 v, s, v, v", out);
  run<5>("v_pk_mul_f32 v, v, v", out);
  return 0;
}
