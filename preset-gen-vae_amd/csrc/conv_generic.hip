// Generic (any shape) convolution kernels: one thread per output element / one block per weight slice.
// They are the always-correct fallback for shapes the tuned gfx950 kernels (conv_direct.hip, conv_mfma.hip)
// do not cover, and the in-library cross-check used by the tests (pgv_set_kernel_policy(1)).
//
// Semantics follow torch's Conv2d / ConvTranspose2d as called by the reference's layer.Conv2D / layer.TConv2D
// (model/layer.py:19-20, :38-40): zero padding, cross-correlation, weight [Cs][Cb][kh][kw].
#include "pgv_common.h"
#include "conv_kernels.h"

namespace {

__global__ void conv_down_generic(pgv_conv_desc d, const float* __restrict__ big,
                                  const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                  const float* __restrict__ w, const float* __restrict__ bias, int act,
                                  float slope, float* __restrict__ small_out, int64_t total) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ow = idx % d.Ws;
  int64_t t = idx / d.Ws;
  const int oh = t % d.Hs;
  t /= d.Hs;
  const int cs = t % d.Cs;
  const int b = t / d.Cs;
  float acc = bias ? bias[cs] : 0.f;
  const bool bf = (d.flags & PGV_COMPUTE_BF16) != 0;
  const int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;
  for (int cb = 0; cb < d.Cb; ++cb) {
    const float sc = in_scale ? in_scale[cb] : 1.f, sh = in_shift ? in_shift[cb] : 0.f;
    const float* plane = big + ((int64_t)b * d.Cb + cb) * d.Hb * d.Wb;
    const float* wk = w + ((int64_t)cs * d.Cb + cb) * d.kh * d.kw;
    for (int kh = 0; kh < d.kh; ++kh) {
      const int ih = ih0 + kh;
      if (ih < 0 || ih >= d.Hb) continue;
      for (int kw = 0; kw < d.kw; ++kw) {
        const int iw = iw0 + kw;
        if (iw < 0 || iw >= d.Wb) continue;
        acc = fmaf(pgv_opnd(fmaf(plane[ih * d.Wb + iw], sc, sh), bf), pgv_opnd(wk[kh * d.kw + kw], bf), acc);
      }
    }
  }
  small_out[idx] = pgv_act(acc, act, slope);
}

__global__ void conv_up_generic(pgv_conv_desc d, const float* __restrict__ small_in,
                                const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                const float* __restrict__ w, const float* __restrict__ bias, int act, float slope,
                                float* __restrict__ big_out, int64_t total) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int iw = idx % d.Wb;
  int64_t t = idx / d.Wb;
  const int ih = t % d.Hb;
  t /= d.Hb;
  const int cb = t % d.Cb;
  const int b = t / d.Cb;
  float acc = bias ? bias[cb] : 0.f;
  const bool bf = (d.flags & PGV_COMPUTE_BF16) != 0;
  for (int cs = 0; cs < d.Cs; ++cs) {
    const float sc = in_scale ? in_scale[cs] : 1.f, sh = in_shift ? in_shift[cs] : 0.f;
    const float* plane = small_in + ((int64_t)b * d.Cs + cs) * d.Hs * d.Ws;
    const float* wk = w + ((int64_t)cs * d.Cb + cb) * d.kh * d.kw;
    for (int kh = 0; kh < d.kh; ++kh) {
      const int th = ih + d.pad - kh;
      if (th < 0 || th % d.stride) continue;
      const int oh = th / d.stride;
      if (oh >= d.Hs) continue;
      for (int kw = 0; kw < d.kw; ++kw) {
        const int tw = iw + d.pad - kw;
        if (tw < 0 || tw % d.stride) continue;
        const int ow = tw / d.stride;
        if (ow >= d.Ws) continue;
        acc = fmaf(pgv_opnd(fmaf(plane[oh * d.Ws + ow], sc, sh), bf), pgv_opnd(wk[kh * d.kw + kw], bf), acc);
      }
    }
  }
  big_out[idx] = pgv_act(acc, act, slope);
}

// One block per (cs, cb, split): threads stride over the (b, oh, ow) pixels of the split and keep kh*kw
// partial sums in registers; block reduction, then one atomicAdd per tap (gw pre-zeroed).
template <int KK>
__global__ void conv_wgrad_generic(pgv_conv_desc d, const float* __restrict__ big,
                                   const float* __restrict__ big_scale, const float* __restrict__ big_shift,
                                   const float* __restrict__ small_in, const float* __restrict__ small_scale,
                                   const float* __restrict__ small_shift, float* __restrict__ gw, int nsplit) {
  __shared__ float red[16];
  const int pair = blockIdx.x;
  const int cs = pair / d.Cb, cb = pair % d.Cb;
  const int split = blockIdx.y;
  const float bsc = big_scale ? big_scale[cb] : 1.f, bsh = big_shift ? big_shift[cb] : 0.f;
  const float ssc = small_scale ? small_scale[cs] : 1.f, ssh = small_shift ? small_shift[cs] : 0.f;
  float acc[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) acc[i] = 0.f;
  const int64_t npix = (int64_t)d.B * d.Hs * d.Ws;
  const int64_t per = (npix + nsplit - 1) / nsplit;
  const int64_t p0 = split * per, p1 = min(npix, p0 + per);
  const int kk = d.kh * d.kw;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const int ow = p % d.Ws;
    int64_t t = p / d.Ws;
    const int oh = t % d.Hs;
    const int b = t / d.Hs;
    const bool bf = (d.flags & PGV_COMPUTE_BF16) != 0;
    const float g = pgv_opnd(fmaf(small_in[(((int64_t)b * d.Cs + cs) * d.Hs + oh) * d.Ws + ow], ssc, ssh), bf);
    const float* plane = big + ((int64_t)b * d.Cb + cb) * d.Hb * d.Wb;
    const int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      if (k < kk) {
        const int ih = ih0 + k / d.kw, iw = iw0 + k % d.kw;
        if (ih >= 0 && ih < d.Hb && iw >= 0 && iw < d.Wb)
          acc[k] = fmaf(g, pgv_opnd(fmaf(plane[ih * d.Wb + iw], bsc, bsh), bf), acc[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    if (k < kk) {
      float s = pgv_block_sum(acc[k], red);
      if (threadIdx.x == 0) atomicAdd(&gw[(int64_t)pair * kk + k], s);
    }
  }
}

}  // namespace

int pgv_conv_down_generic(const pgv_conv_desc* d, const float* big, const float* in_scale,
                          const float* in_shift, const float* w, const float* bias, int act, float slope,
                          float* small_out, hipStream_t st) {
  const int64_t total = (int64_t)d->B * d->Cs * d->Hs * d->Ws;
  if (total == 0) return PGV_OK;
  hipLaunchKernelGGL(conv_down_generic, dim3((unsigned)pgv_cdiv(total, 256)), dim3(256), 0, st, *d, big, in_scale,
                     in_shift, w, bias, act, slope, small_out, total);
  PGV_CHECK_LAUNCH("conv_down_generic");
  return PGV_OK;
}

int pgv_conv_up_generic(const pgv_conv_desc* d, const float* small_in, const float* in_scale,
                        const float* in_shift, const float* w, const float* bias, int act, float slope,
                        float* big_out, hipStream_t st) {
  const int64_t total = (int64_t)d->B * d->Cb * d->Hb * d->Wb;
  if (total == 0) return PGV_OK;
  hipLaunchKernelGGL(conv_up_generic, dim3((unsigned)pgv_cdiv(total, 256)), dim3(256), 0, st, *d, small_in,
                     in_scale, in_shift, w, bias, act, slope, big_out, total);
  PGV_CHECK_LAUNCH("conv_up_generic");
  return PGV_OK;
}

int pgv_conv_wgrad_generic(const pgv_conv_desc* d, const float* big, const float* big_scale,
                           const float* big_shift, const float* small_in, const float* small_scale,
                           const float* small_shift, float* gw, hipStream_t st) {
  const int kk = d->kh * d->kw;
  PGV_CHECK_ARG(kk <= 25, "conv_wgrad: kernel %dx%d larger than 5x5 unsupported", d->kh, d->kw);
  const int64_t nw = (int64_t)d->Cs * d->Cb * kk;
  hipError_t e = (d->flags & PGV_PREZEROED) ? hipSuccess : hipMemsetAsync(gw, 0, nw * sizeof(float), st);
  if (e != hipSuccess) {
    pgv_set_error("conv_wgrad: memset failed: %s", hipGetErrorString(e));
    return PGV_E_LAUNCH;
  }
  const int64_t npix = (int64_t)d->B * d->Hs * d->Ws;
  if (npix == 0) return PGV_OK;
  const int64_t pairs = (int64_t)d->Cs * d->Cb;
  // Aim at ~4096 blocks; each split should still hold a few hundred pixels per thread-block pass.
  int nsplit = (int)max((int64_t)1, min(pgv_cdiv(4096, pairs), pgv_cdiv(npix, 1024)));
  dim3 grid((unsigned)pairs, (unsigned)nsplit);
  if (kk <= 1)
    hipLaunchKernelGGL(conv_wgrad_generic<1>, grid, dim3(256), 0, st, *d, big, big_scale, big_shift, small_in,
                       small_scale, small_shift, gw, nsplit);
  else if (kk <= 16)
    hipLaunchKernelGGL(conv_wgrad_generic<16>, grid, dim3(256), 0, st, *d, big, big_scale, big_shift, small_in,
                       small_scale, small_shift, gw, nsplit);
  else
    hipLaunchKernelGGL(conv_wgrad_generic<25>, grid, dim3(256), 0, st, *d, big, big_scale, big_shift, small_in,
                       small_scale, small_shift, gw, nsplit);
  PGV_CHECK_LAUNCH("conv_wgrad_generic");
  return PGV_OK;
}
