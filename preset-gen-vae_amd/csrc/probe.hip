// Measured-peak probes for bench.py ("use the measured-peak and nominal-peak both"): what this box
// sustains on the two resources the rooflines are priced against - the matrix pipe (a dependency-free stream of
// v_mfma_f32_16x16x4_f32 or v_mfma_f32_16x16x32_bf16 from one wave per SIMD on every CU, lane-dependent non-zero operands:
// zeros clock higher) and HBM reads (a 16-byte-per-lane grid-stride reduction over a buffer larger than the caches).
// Not part of the train step.
#include "pgv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <bool BF16>
__global__ __launch_bounds__(256) void probe_mfma_kernel(int iters, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[32];
#pragma unroll
  for (int n = 0; n < 32; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + 0.01f * lane, b = 0.5f - 0.003f * lane;
  bf16x8_t a8, b8;
#pragma unroll
  for (int i = 0; i < 8; ++i) a8[i] = (__bf16)(a + 0.1f * i), b8[i] = (__bf16)(b - 0.05f * i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      if constexpr (BF16)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[n], 0, 0, 0);
      else
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < 32; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  if (s == 12345.678f) sink[0] = s;   // (keeps the loop alive; never true in practice)
}

template <int U>
__global__ __launch_bounds__(256) void probe_read_kernel(const f32x4* __restrict__ buf, int64_t n4, float* __restrict__ sink) {
  // U 16-byte loads in flight per lane, every workgroup walks its own contiguous slab (one DRAM page stream per workgroup)
  f32x4 s[U];
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  int64_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] += __builtin_nontemporal_load(buf + i + u * 256);
  }
  for (; i < hi; i += 256) s[0] += buf[i];
  f32x4 t = s[0];
#pragma unroll
  for (int u = 1; u < U; ++u) t += s[u];
  const float r = (t[0] + t[1]) + (t[2] + t[3]);
  if (r == 12345.678f) sink[0] = r;
}

}  // namespace

extern "C" {

int pgv_probe_mfma(int bf16, int iters, float* sink, int64_t* flops, void* stream) {
  PGV_CHECK_ARG(iters > 0 && sink, "pgv_probe_mfma: bad argument");
  const int grid = 256;   // one 4-wave workgroup per CU: one wave per SIMD
  if (flops) *flops = (int64_t)grid * 4 * iters * 32 * (bf16 ? 2 * 16 * 16 * 32 : 2 * 16 * 16 * 4);
  if (bf16)
    hipLaunchKernelGGL(probe_mfma_kernel<true>, dim3(grid), dim3(256), 0, pgv_stream(stream), iters, sink);
  else
    hipLaunchKernelGGL(probe_mfma_kernel<false>, dim3(grid), dim3(256), 0, pgv_stream(stream), iters, sink);
  PGV_CHECK_LAUNCH("probe_mfma");
  return PGV_OK;
}

int pgv_probe_read(const float* buf, int64_t n, float* sink, void* stream) {
  PGV_CHECK_ARG(buf && sink && n >= 4 && (reinterpret_cast<uintptr_t>(buf) & 15) == 0, "pgv_probe_read: bad argument");
  hipLaunchKernelGGL(probe_read_kernel<8>, dim3(256 * 8), dim3(256), 0, pgv_stream(stream),
                     reinterpret_cast<const f32x4*>(buf), n / 4, sink);
  PGV_CHECK_LAUNCH("probe_read");
  return PGV_OK;
}

}  // extern "C"
