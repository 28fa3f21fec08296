// ws_share.hip with the MFMA shape as a parameter: does a partner wave (VALU / LDS stores / LDS reads) get more issue slots
// beside a stream of v_mfma_f32_32x32x2_f32 (64 clk each, 2048 MACs) than beside v_mfma_f32_16x16x4_f32 (32 clk, 1024 MACs)?
// And what does an operand read cost the MFMA wave itself per MFMA (1 ds_read_b32 per MFMA = the conv kernels' k-step)?
// SHAPE 0 = 16x16x4 (32 accumulator tiles of 4 regs), 1 = 32x32x2 (8 tiles of 16 regs): the same 128 accumulator registers.
// PART: 0 idle partner, 1 eight independent fma chains, 2 ds_write_b128 x16 + wait, 3 ds_read_b128 x16 + wait,
//       4 "lean loader": per 16-byte slot two v_pk_fma + one ds_write_b128.
// OWN: ds_read_b32 issued by the MFMA wave itself per MFMA (0, 1 or 2), consumed two MFMAs later.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int SHAPE, int PART, int OWN>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* stamps, int iters) {
  __shared__ f32x4 lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = f32x4{1.f * i, 2.f, 3.f, 4.f};
  __syncthreads();
  unsigned long long c0 = clock64();
  if (wave < 4) {
    const float* lf = reinterpret_cast<const float*>(lds);
    float a = 1.0f + lane, b = 0.5f * lane;
    if (SHAPE == 0) {
      f32x4 acc[32];
      for (int n = 0; n < 32; ++n) acc[n] = f32x4{0, 0, 0, 0};
      float q[4] = {a, a, a, a};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 32; ++n) {
          if (OWN >= 1) q[(n + 2) & 3] = lf[lane + 64 * ((n + it) & 31)];
          if (OWN >= 2) b = lf[2048 + lane + 64 * ((n + it) & 31)];
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(OWN ? q[n & 3] : a, b, acc[n], 0, 0, 0);
        }
      }
      for (int n = 0; n < 32; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    } else {
      f32x16 acc[8];
      for (int n = 0; n < 8; ++n)
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
      float q[4] = {a, a, a, a};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            if (OWN >= 1) q[(n + 2) & 3] = lf[lane + 64 * ((n + it + 8 * r) & 31)];
            if (OWN >= 2) b = lf[2048 + lane + 64 * ((n + it) & 31)];
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(OWN ? q[n & 3] : a, b, acc[n], 0, 0, 0);
          }
      }
      for (int n = 0; n < 8; ++n)
        for (int e = 0; e < 16; ++e) s += acc[n][e];
    }
  } else {
    __builtin_amdgcn_s_setprio(2);
    const int lt = threadIdx.x - 256;
    if (PART == 1) {
      float x[8];
      for (int n = 0; n < 8; ++n) x[n] = lane + n;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int n = 0; n < 8; ++n) x[n] = fmaf(x[n], 1.0001f, 0.5f);
      }
      for (int n = 0; n < 8; ++n) s += x[n];
    } else if (PART == 2) {
      f32x4 v = {1, 2, 3, (float)lane};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) lds[2048 + lt + 256 * (n & 7)] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      s = lds[2048 + lane].x;
    } else if (PART == 3) {
      f32x4 v = {0, 0, 0, 0};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) v += lds[lt + 256 * ((n + it) & 7)];
      }
      s = v.x + v.y + v.z + v.w;
    } else if (PART == 4) {
      f32x4 v = {1, 2, 3, (float)lane};
      const f32x2 m = {1.0001f, 1.0002f}, ad = {0.5f, 0.25f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {   // 5 slots x 3 instructions + the wait = 16 instructions per iteration
          f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
          lo = __builtin_elementwise_fma(lo, m, ad);
          hi = __builtin_elementwise_fma(hi, m, ad);
          v = f32x4{lo.x, lo.y, hi.x, hi.y};
          lds[2048 + lt + 256 * n] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      s = lds[2048 + lane].x;
    }
  }
  unsigned long long c1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) stamps[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int SHAPE, int PART, int OWN>
void run(float* out, unsigned long long* st, const char* name) {
  const int iters = 400;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<SHAPE, PART, OWN>), dim3(256), dim3(512), 0, 0, out, st, iters);
  (void)hipDeviceSynchronize();
  static unsigned long long h[256 * 8];
  (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  double tm = 0, tv = 0;
  for (int i = 0; i < 256; ++i)
    for (int w = 0; w < 4; ++w) tm += h[i * 8 + w], tv += h[i * 8 + 4 + w];
  const int mf = SHAPE == 0 ? 32 : 16;   // MFMAs per iteration (the same 32768 MACs per lane-row either way)
  printf("%s own_reads %d  %-26s: %6.1f clk per MFMA = %6.2f clk per 1024 MACs, %6.1f clk per partner instruction\n",
         SHAPE ? "32x32x2" : "16x16x4", OWN, name, tm / 1024 / iters / mf, tm / 1024 / iters / 32, tv / 1024 / iters / 16);
}

int main() {
  float* out;
  unsigned long long* st;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&st, 256 * 8 * 8);
#define ALL(S, O)                                     \
  run<S, 0, O>(out, st, "idle partner");              \
  run<S, 1, O>(out, st, "8 independent fma chains");  \
  run<S, 2, O>(out, st, "ds_write_b128 x16 + wait");  \
  run<S, 3, O>(out, st, "ds_read_b128 x16");          \
  run<S, 4, O>(out, st, "lean loader (2 pk_fma + ds_write_b128)");
  ALL(0, 0) ALL(1, 0) ALL(0, 1) ALL(1, 1) ALL(0, 2) ALL(1, 2)
  return 0;
}
