// Shared by the deep-layer kernels (conv_deep.hip: fp32 LDS images; conv_deep_bf16.hip: bf16-native images).
#pragma once
#include "conv_tile.h"

namespace {

// XCD-aware (channel block, sample group) of a workgroup.  Workgroups are dealt round-robin over the 8 XCDs, each with
// its own L2: with blockIdx = mb * groups + grp every XCD works on ALL channel blocks at once and streams the whole
// weight tensor (up to 8 MB against a 4 MB L2) again and again.  Here the workgroups of one XCD share a channel block
// whenever the block count divides 8 (its weight slice, 1 MB or less, then stays in that XCD's L2 for all sample
// groups); a speed matter only.
__device__ __forceinline__ void deep_block(int nmb, int groups, int& mb, int& grp) {
  const int b = (int)blockIdx.x;
  if (nmb <= 8 && 8 % nmb == 0 && (nmb * groups) % 8 == 0) {
    const int x = b & 7, q = b >> 3, per = 8 / nmb;   // per = XCDs per channel block
    mb = x % nmb;
    grp = q * per + x / nmb;
  } else if (nmb % 8 == 0) {   // more channel blocks than XCDs: nmb / 8 of them per XCD, one after the other
    const int x = b & 7, q = b >> 3;
    mb = x * (nmb / 8) + q / groups;
    grp = q % groups;
  } else {
    mb = b / groups;
    grp = b - mb * groups;
  }
}

}  // namespace
