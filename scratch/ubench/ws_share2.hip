// Follow-up of ws_share.hip: the MFMA waves interleave own ds_reads (RD reads per 32 MFMAs) like the real kernels; the
// partner runs MODE 0 v_fma chain, 1 s_add chain (SALU), 2 buffer_load_dwordx4 stream (inline asm, vmcnt waits every 16),
// 3 ds_write stream.  Also checks that out-of-range buffer lanes return zero.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int RD>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* stamps, const float* src, int nbytes, int iters) {
  __shared__ f32x4 lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = f32x4{1, 2, 3, 4};
  float s = 0.f;
  __syncthreads();
  unsigned long long c0 = clock64();
  if (wave < 4) {
    f32x4 acc[32];
    for (int n = 0; n < 32; ++n) acc[n] = f32x4{0, 0, 0, 0};
    const float a = 1.0f + lane;
    const float* l = reinterpret_cast<const float*>(lds) + lane;
    float b[2][12];
    for (int n = 0; n < 12; ++n) b[0][n] = b[1][n] = 0.5f;
    for (int it = 0; it < iters; ++it) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < RD; ++n) b[(it + 1) & 1][n] = l[n * 64 + (it & 15) * 1024];
#pragma unroll
      for (int n = 0; n < 32; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[it & 1][n % 12], acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < 32; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (n < RD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    for (int n = 0; n < 32; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  } else {
    if (MODE == 0) {
      float x = lane;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) x = fmaf(x, 1.0001f, 0.5f);
      }
      s = x;
    } else if (MODE == 1) {
      int x = __builtin_amdgcn_readfirstlane(iters);
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 7" : "+s"(x));
      }
      s = x;  // 32 SALU per iteration
    } else if (MODE == 2) {
      const uint64_t p = (uint64_t)src;
      i32x4 rsrc = {(int)(uint32_t)p, (int)(uint32_t)(p >> 32), nbytes, 0x00020000};
      rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x);
      rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
      rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z);
      rsrc.w = __builtin_amdgcn_readfirstlane(rsrc.w);
      f32x4 v[16];
      unsigned off = (threadIdx.x - 256) * 16 + blockIdx.x * 4096;
      if (lane >= 60) off = 0xFFFFFFF0u;  // out of range: must read as zero
      f32x4 sum = {0, 0, 0, 0};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n)
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[n]) : "v"(off), "s"(rsrc) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int n = 0; n < 16; ++n) asm volatile("" : "+v"(v[n]));
        sum += v[it & 15];
      }
      s = sum.x + sum.y + sum.z + sum.w;
    } else if (MODE == 3) {
      f32x4 v = {1, 2, 3, (float)lane};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) lds[(threadIdx.x - 256) + 256 * (n & 7) + 2048] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      s = lds[lane].x;
    }
  }
  unsigned long long c1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) stamps[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int MODE, int RD>
void run(float* out, unsigned long long* st, const float* src, const char* name) {
  const int iters = 500;
  for (int r = 0; r < 2; ++r)
    hipLaunchKernelGGL((k<MODE, RD>), dim3(256), dim3(512), 0, 0, out, st, src, 256 * 4096, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[256 * 8];
  (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  double tm = 0, tv = 0;
  for (int i = 0; i < 256; ++i)
    for (int w = 0; w < 4; ++w) tm += h[i * 8 + w], tv += h[i * 8 + 4 + w];
  static float ho[256 * 512];
  (void)hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  printf("%-24s %2d own reads per 32 MFMA: %6.1f clk per MFMA, %6.1f clk per partner instruction   (lane 60 sum %g, lane 0 sum %g)\n",
         name, RD, tm / 1024 / iters / 32, tv / 1024 / iters / (MODE == 1 ? 32 : 16), ho[256 + 60], ho[256]);
}

int main() {
  float *out, *src;
  unsigned long long* st;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&src, 256 * 4096);
  (void)hipMalloc(&st, 256 * 8 * 8);
  static float hs[256 * 1024];
  for (int i = 0; i < 256 * 1024; ++i) hs[i] = 1.0f;
  (void)hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
  run<0, 0>(out, st, src, "v_fma chain");
  run<0, 12>(out, st, src, "v_fma chain");
  run<1, 0>(out, st, src, "SALU chain");
  run<1, 12>(out, st, src, "SALU chain");
  run<2, 0>(out, st, src, "buffer_load x16 + wait");
  run<2, 12>(out, st, src, "buffer_load x16 + wait");
  run<3, 0>(out, st, src, "ds_write x16 + wait");
  run<3, 12>(out, st, src, "ds_write x16 + wait");
  return 0;
}
