// Tuned gfx950 convolution kernels (dispatch).  Returns 1 when a tuned kernel handled the call.
#include "conv_kernels.h"

int pgv_conv_wgrad_tuned(const pgv_conv_desc*, const float*, const float*, const float*, const float*, const float*,
                         const float*, float*, void*, int64_t, hipStream_t) {
  return 0;
}
int64_t pgv_conv_wgrad_tuned_workspace(const pgv_conv_desc*) { return 0; }
