// One workgroup of 8 waves per CU: waves 0-3 run a dense MFMA stream (32 independent accumulator tiles), waves 4-7 run
// VALU work.  How many cycles does a VALU instruction of the second wave of a SIMD cost while the first one keeps the
// MFMA pipeline full, and what does it cost the MFMA wave?  MODE 0 dependent chain, 1 independent (8 chains),
// 2 = 1 + v_pk_fma, 3 = ds_write_b128 stream, 4 = idle partner.  PRIO = s_setprio of the VALU waves (MFMA waves 0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int PRIO, int MPRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* stamps, int iters) {
  __shared__ f32x4 lds[2048];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.f;
  __syncthreads();
  unsigned long long c0 = clock64();
  if (wave < 4) {
    __builtin_amdgcn_s_setprio(MPRIO);
    f32x4 acc[32];
    for (int n = 0; n < 32; ++n) acc[n] = f32x4{0, 0, 0, 0};
    const float a = 1.0f + lane, b = 0.5f * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < 32; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
    }
    for (int n = 0; n < 32; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  } else {
    __builtin_amdgcn_s_setprio(PRIO);
    if (MODE == 0) {
      float x = lane;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) x = fmaf(x, 1.0001f, 0.5f);
      }
      s = x;
    } else if (MODE == 1) {
      float x[8];
      for (int n = 0; n < 8; ++n) x[n] = lane + n;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int n = 0; n < 8; ++n) x[n] = fmaf(x[n], 1.0001f, 0.5f);
      }
      for (int n = 0; n < 8; ++n) s += x[n];
    } else if (MODE == 2) {
      f32x2 x[8];
      for (int n = 0; n < 8; ++n) x[n] = f32x2{(float)lane, (float)n};
      const f32x2 m = {1.0001f, 1.0002f}, a = {0.5f, 0.25f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int n = 0; n < 8; ++n) x[n] = __builtin_elementwise_fma(x[n], m, a);
      }
      for (int n = 0; n < 8; ++n) s += x[n].x + x[n].y;
    } else if (MODE == 3) {
      f32x4 v = {1, 2, 3, (float)lane};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) lds[(threadIdx.x - 256) + 256 * (n & 7)] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      s = lds[lane].x;
    }
  }
  unsigned long long c1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) stamps[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int MODE, int PRIO, int MPRIO>
void run(float* out, unsigned long long* st, const char* name) {
  const int iters = 500;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, PRIO, MPRIO>), dim3(256), dim3(512), 0, 0, out, st, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[256 * 8];
  (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  double tm = 0, tv = 0;
  for (int i = 0; i < 256; ++i)
    for (int w = 0; w < 4; ++w) tm += h[i * 8 + w], tv += h[i * 8 + 4 + w];
  printf("%-28s prio %d/%d: %6.1f clk per MFMA, %6.1f clk per partner instruction\n", name, MPRIO, PRIO, tm / 1024 / iters / 32,
         tv / 1024 / iters / 16);
}

int main() {
  float* out;
  unsigned long long* st;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&st, 256 * 8 * 8);
  run<4, 0, 0>(out, st, "idle partner");
  run<0, 0, 0>(out, st, "dependent fma chain");
  run<0, 2, 0>(out, st, "dependent fma chain");
  run<0, 3, 0>(out, st, "dependent fma chain");
  run<0, 0, 2>(out, st, "dependent fma chain");
  run<1, 0, 0>(out, st, "8 independent fma chains");
  run<1, 2, 0>(out, st, "8 independent fma chains");
  run<2, 2, 0>(out, st, "8 independent pk_fma chains");
  run<3, 2, 0>(out, st, "ds_write_b128 x16 + wait");
  return 0;
}
