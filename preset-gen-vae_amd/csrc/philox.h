// Counter-based random numbers of the train step (Philox4x32-10): shared by the Dropout / eps kernels (eltwise.hip) and
// the BatchNorm-backward pass that regenerates a Dropout mask on the way (bn.hip).
#pragma once
#include "pgv_common.h"

namespace {

// ---- Philox4x32-10 -------------------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ U4 philox4x32_10(uint64_t counter, uint64_t stream_id, uint64_t key) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = (uint32_t)stream_id,
           c3 = (uint32_t)(stream_id >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                   n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }  // [0,1)

// keep mask (pre-scaled) of the four elements 4q .. 4q+3 of a Dropout draw
__device__ __forceinline__ void dropout_mask4(uint64_t off, uint64_t q, uint64_t stream_id, uint64_t seed, float p,
                                              float keep_scale, float (&m)[4]) {
  const U4 r = philox4x32_10(off + q, stream_id, seed);
  m[0] = u01(r.x) >= p ? keep_scale : 0.f, m[1] = u01(r.y) >= p ? keep_scale : 0.f;
  m[2] = u01(r.z) >= p ? keep_scale : 0.f, m[3] = u01(r.w) >= p ? keep_scale : 0.f;
}

}  // namespace
