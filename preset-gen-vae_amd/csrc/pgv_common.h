// Shared helpers for the gfx950 kernels. wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pgv_hip.h"
static_assert((PGV_CLS_COPIES & (PGV_CLS_COPIES - 1)) == 0, "copy index = workgroup & (copies - 1)");

#define PGV_WAVE 64

void pgv_set_error(const char* fmt, ...);
int pgv_kernel_policy();

#define PGV_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      pgv_set_error(__VA_ARGS__);       \
      return PGV_E_INVALID;             \
    }                                   \
  } while (0)

#define PGV_CHECK_LAUNCH(name)                                                  \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      pgv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return PGV_E_LAUNCH;                                                      \
    }                                                                           \
  } while (0)

static inline hipStream_t pgv_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t pgv_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float pgv_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). Result valid in thread 0 (and all of wave 0).
__device__ __forceinline__ float pgv_block_sum(float v, float* smem /* >= 16 floats */) {
  v = pgv_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (wave == 0) {
    r = lane < nw ? smem[lane] : 0.f;
    r = pgv_wave_sum(r);
  }
  return r;
}

__device__ __forceinline__ double pgv_wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// float64 block sum (per-channel statistics: the sums cancel heavily downstream, see DESIGN.md numerics).
__device__ __forceinline__ double pgv_block_sum_d(double v, double* smem /* >= 16 doubles */) {
  v = pgv_wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (wave == 0) {
    r = lane < nw ? smem[lane] : 0.0;
    r = pgv_wave_sum_d(r);
  }
  return r;
}

// pgv_bn_finalize's arithmetic for channel c, evaluated by the kernel that CONSUMES the BatchNorm (pgv_bn_src,
// pgv_conv_*_bn): identical float64 expressions in every workgroup; `writer` (one workgroup) also stores the vectors the
// backward pass reads and updates the running statistics.
__device__ __forceinline__ void pgv_bn_finalize_dev(const pgv_bn_src& s, int C, int c, bool writer, float& sc, float& sh) {
  const double inv_n = 1.0 / (double)s.n;
  double sum = s.stats[c], sq = s.stats[C + c];
  for (int r = 1; r < s.stats_copies; ++r) sum += s.stats[r * 2 * C + c], sq += s.stats[r * 2 * C + C + c];   // (PGV_STATS_COPIES)
  const double mean = sum * inv_n;
  double var = sq * inv_n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double rstd = 1.0 / sqrt(var + (double)s.eps);
  const double g = s.gamma ? (double)s.gamma[c] : 1.0, bt = s.beta ? (double)s.beta[c] : 0.0;
  sc = (float)(g * rstd);
  sh = (float)(bt - mean * g * rstd);
  if (writer) {
    const double unbias = s.n > 1 ? (double)s.n / (double)(s.n - 1) : 1.0;
    s.scale[c] = sc;
    s.shift[c] = sh;
    if (s.mean) s.mean[c] = (float)mean;
    if (s.rstd) s.rstd[c] = (float)rstd;
    if (s.running_mean) s.running_mean[c] = (float)((1.0 - s.momentum) * s.running_mean[c] + s.momentum * mean);
    if (s.running_var) s.running_var[c] = (float)((1.0 - s.momentum) * s.running_var[c] + s.momentum * var * unbias);
    if (c == 0 && s.num_batches_tracked) *s.num_batches_tracked += 1;
  }
}
inline pgv_bn_src pgv_no_bn() {
  pgv_bn_src z = {};
  return z;
}

// N block-wide sums at once for blockDim.x == 256: DPP row reductions (no ds_bpermute), v_readlane across the four rows of
// a wave, ONE barrier across the waves.  pgv_block_sum costs two barriers and six LDS-latency shuffles per value - for
// kernels that end with several reductions (class sums, loss, bias gradient) that tail was longer than the streaming
// loop.  Results valid in threads 0..N-1: thread i holds sum i.  smem: >= 4*N floats.
template <int CTRL>
__device__ __forceinline__ float pgv_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ float pgv_block_sums(const float (&v)[N], float* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float t = v[i];
    t += pgv_dpp<0xB1>(t);   // quad_perm [1,0,3,2]
    t += pgv_dpp<0x4E>(t);   // quad_perm [2,3,0,1]
    t += pgv_dpp<0x141>(t);  // row_half_mirror
    t += pgv_dpp<0x140>(t);  // row_mirror: every lane of a 16-lane row holds the row's sum
    const int ti = __float_as_int(t);   // (v_readlane is an integer builtin: no value conversion)
    const float w = (__int_as_float(__builtin_amdgcn_readlane(ti, 0)) + __int_as_float(__builtin_amdgcn_readlane(ti, 16))) +
                    (__int_as_float(__builtin_amdgcn_readlane(ti, 32)) + __int_as_float(__builtin_amdgcn_readlane(ti, 48)));
    if (lane == 0) smem[wave * N + i] = w;
  }
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < N) r = smem[threadIdx.x] + smem[N + threadIdx.x] + smem[2 * N + threadIdx.x] + smem[3 * N + threadIdx.x];
  return r;
}

// Branch-free form of pgv_act for unrolled epilogues: bit-identical results, parameters derived once from (act, slope).
struct pgv_act_params {
  float ns, lo, hi;
};
__device__ __forceinline__ pgv_act_params pgv_act_setup(int act, float slope) {
  pgv_act_params p;
  p.ns = act == PGV_ACT_LEAKY_RELU ? slope : 1.0f;
  p.lo = act == PGV_ACT_HARDTANH ? -1.0f : -__builtin_inff();
  p.hi = act == PGV_ACT_HARDTANH ? 1.0f : __builtin_inff();
  return p;
}
__device__ __forceinline__ float pgv_act_apply(float y, const pgv_act_params& p) {
  const float r = y > 0.f ? y : p.ns * y;
  return fminf(p.hi, fmaxf(p.lo, r));
}
// The same with torch's NaN behaviour at a clamp: Hardtanh(NaN) = NaN (fminf / fmaxf return the other operand).  For the
// output layer's kernels: a NaN there must reach the reconstruction loss, where the reference's harness looks for it
// (utils/exception.py:13-23, train.py:245).  Behind a LeakyReLU the plain form leaves NaN as -inf: still not finite.
__device__ __forceinline__ float pgv_act_apply_nan(float y, const pgv_act_params& p) {
  const float r = pgv_act_apply(y, p);
  return y != y ? y : r;
}

// Operand precision of a product (PGV_COMPUTE_BF16): kernels without a bf16 MFMA loop round their operands to bfloat16
// (RNE) and keep multiplying on the fp32 pipe - products of bf16 values are exact in fp32, so the result differs from
// the bf16 matrix cores only by the fp32 summation order.
__device__ __forceinline__ float pgv_opnd(float x, bool bf16) { return bf16 ? (float)(__bf16)x : x; }

__device__ __forceinline__ float pgv_act(float y, int act, float slope) {
  if (act == PGV_ACT_LEAKY_RELU) return y > 0.f ? y : slope * y;
  if (act == PGV_ACT_HARDTANH) return y != y ? y : fminf(1.f, fmaxf(-1.f, y));   // (NaN stays NaN, as torch)
  return y;
}
