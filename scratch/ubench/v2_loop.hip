// Micro-benchmark of the conv_v2 inner loop: one wave per SIMD (256 threads, 1 workgroup per CU), NT accumulator tiles,
// per k-step one A register and NT B operands read from LDS (ds_read_b32, per-lane base + immediate).
//   MODE 0: MFMA only (operands constant)            MODE 1: source-order interleave, compiler scheduled
//   MODE 2: read burst of step s+1, then MFMAs of s   MODE 3: 1 : 1 interleave pinned with sched_group_barrier per step
//   MODE 4: reads two steps ahead, pinned 1 : 1
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int MODE, int NT, int S, int WP, int PLANE, int WS>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* stamps, const float* wsrc, int reps) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < (S / 4) * PLANE + 64; i += 256) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const float* tile = lds + 4;
  f32x4 acc[NT];
  for (int n = 0; n < NT; ++n) acc[n] = f32x4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  int offB[NT];
  for (int t = 0; t < NT; ++t) {
    const int p = t * 16 + (lane & 15);
    const int r = p / WS, c = p - r * WS;
    offB[t] = 2 * r * WP + 2 * c - 2 + (lane >> 4);
  }
  float a[S];
  for (int s = 0; s < S; ++s) a[s] = wsrc[(s * 64 + lane) & 1023];
  unsigned long long t0 = wall_clock64(), c0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    if (MODE == 0) {
      float bc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bc[t] = tile[offB[t]];
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = MF(a[st], bc[t], acc[t]);
    } else if (MODE == 1 || MODE == 3) {
      float b0[NT], b1[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b0[t] = tile[offB[t]];
#pragma unroll
      for (int st = 0; st < S; ++st) {
        const int sn = st + 1, cn = sn / 4, khn = sn - cn * 4;
        float(&bc)[NT] = (st & 1) ? b1 : b0;
        float(&bn)[NT] = (st & 1) ? b0 : b1;
        if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (sn < S) bn[t] = tile[cn * PLANE + khn * WP + offB[t]];
          acc[t] = MF(a[st], bc[t], acc[t]);
        }
        if (MODE == 3) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (sn < S) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (MODE == 2) {
      float b0[NT], b1[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b0[t] = tile[offB[t]];
#pragma unroll
      for (int st = 0; st < S; ++st) {
        const int sn = st + 1, cn = sn / 4, khn = sn - cn * 4;
        float(&bc)[NT] = (st & 1) ? b1 : b0;
        float(&bn)[NT] = (st & 1) ? b0 : b1;
        __builtin_amdgcn_sched_barrier(0);
        if (sn < S) {
#pragma unroll
          for (int t = 0; t < NT; ++t) bn[t] = tile[cn * PLANE + khn * WP + offB[t]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = MF(a[st], bc[t], acc[t]);
      }
    } else if (MODE == 4) {
      // three rotating operand sets: reads of step s+2 ride along the MFMAs of step s
      float b[3][NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[0][t] = tile[offB[t]];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[1][t] = tile[WP + offB[t]];
#pragma unroll
      for (int st = 0; st < S; ++st) {
        const int sn = st + 2, cn = sn / 4, khn = sn - cn * 4;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (sn < S) b[sn % 3][t] = tile[cn * PLANE + khn * WP + offB[t]];
          acc[t] = MF(a[st], b[st % 3][t], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (sn < S) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  unsigned long long c1 = clock64(), t1 = wall_clock64();
  float s = 0;
  for (int n = 0; n < NT; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = c1 - c0;
  }
}

template <int MODE, int NT, int S, int WP, int PLANE, int WS>
void run(const char* name) {
  float *out, *w;
  unsigned long long* st;
  const int nb = 256;
  (void)hipMalloc(&out, nb * 256 * 4);
  (void)hipMalloc(&w, 4096);
  (void)hipMemset(w, 0, 4096);
  (void)hipMalloc(&st, nb * 16);
  const int reps = 16;
  auto kern = k<MODE, NT, S, WP, PLANE, WS>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  size_t lds = 4 * ((S / 4) * PLANE + 64);
  if (lds < 90 * 1024) lds = 90 * 1024;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, out, st, w, reps);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, out, st, w, reps);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2 * 256];
  (void)hipMemcpy(h, st, 16 * 256, hipMemcpyDeviceToHost);
  double wall = 0, clk = 0;
  for (int i = 0; i < 256; ++i) wall += h[2 * i], clk += h[2 * i + 1];
  wall /= 256, clk /= 256;
  const double n_mfma = (double)reps * NT * S;
  printf("%-44s NT %2d: %6.2f ns/MFMA  %5.1f clk/MFMA  (clock %.2f GHz, kernel %.1f us, hipErr %d)\n", name, NT,
         wall * 10.0 / n_mfma, clk / n_mfma, clk / (wall * 10.0), ms * 1e3, (int)hipGetLastError());
  (void)hipFree(out);
  (void)hipFree(st);
  (void)hipFree(w);
}

int main() {
  run<0, 25, 32, 48, 1728, 23>("0 MFMA only");
  run<1, 25, 32, 48, 1728, 23>("1 source interleave, compiler sched");
  run<2, 25, 32, 48, 1728, 23>("2 read burst then MFMAs");
  run<3, 25, 32, 48, 1728, 23>("3 pinned 1:1, one step ahead");
  run<4, 25, 32, 48, 1728, 23>("4 pinned 1:1, two steps ahead");
  run<0, 13, 32, 48, 1728, 23>("0 MFMA only");
  run<3, 13, 32, 48, 1728, 23>("3 pinned 1:1, one step ahead");
  run<4, 13, 32, 48, 1728, 23>("4 pinned 1:1, two steps ahead");
  return 0;
}
