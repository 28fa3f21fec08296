"""Tensor-level wrappers over the C ABI (``include/pgv_hip.h``): torch supplies device memory and the current
HIP stream, every arithmetic step runs in the hand-written gfx950 kernels.  No wrapper has a CPU path: tensors must
be fp32, contiguous and on a ROCm device, otherwise a ``RuntimeError`` is raised."""
import ctypes

import torch

from . import _lib
from ._lib import PGV_ACT_HARDTANH, PGV_ACT_LEAKY_RELU, PGV_ACT_NONE, ConvDesc  # noqa: F401


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("preset-gen-vae_amd ops need tensors on a ROCm device (no CPU fallback exists)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("expected a contiguous (NCHW) tensor")


def _chk64(*tensors):
    """Per-channel reduction buffers (BN statistics / BN-backward projections) are float64 on the device."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
            raise RuntimeError("expected a contiguous float64 tensor on a ROCm device for reduction buffers")


def _p(t):
    return None if t is None else t.data_ptr()


PGV_PREZEROED, PGV_COMPUTE_BF16, PGV_STATS_COPIES = 1, 2, 4
CLS_COPIES = 8   # PGV_CLS_COPIES: class sums (pgv_bwd_fuse.cls) are kept as this many partial copies, one per XCD
_COMPUTE_FLAGS = 0


def set_compute_dtype(dtype):
    """'fp32' (default: exact fp32 products on the f32 matrix cores) or 'bf16' (PGV_COMPUTE_BF16: operands of every
    convolution / linear product rounded to bfloat16, bf16 matrix cores with fp32 accumulation; storage stays fp32)."""
    global _COMPUTE_FLAGS
    if dtype in ('fp32', 'f32', torch.float32):
        _COMPUTE_FLAGS = 0
    elif dtype in ('bf16', torch.bfloat16):
        _COMPUTE_FLAGS = PGV_COMPUTE_BF16
    else:
        raise ValueError(f"unknown compute dtype {dtype!r}")


def compute_dtype():
    return 'bf16' if _COMPUTE_FLAGS & PGV_COMPUTE_BF16 else 'fp32'


PGV_COMPUTE_F32_SPLIT = 8
# The product's default form of fp32 products: what ``config.train.fp32_products = None`` / ``VAETrainStep()`` run and what
# bench.py times on its headline line (one default for library, training path and benchmark).
DEFAULT_FP32_PRODUCTS = 'bf16x6'
_F32_SPLIT = DEFAULT_FP32_PRODUCTS == 'bf16x6'


def set_fp32_products(mode):
    """How fp32-mode products are evaluated by the layers that have a kernel for both: 'bf16x6' (the default,
    ``DEFAULT_FP32_PRODUCTS``; PGV_COMPUTE_F32_SPLIT: every operand as three exact bfloat16 terms, six bf16 matrix
    instructions per product with fp32 accumulation - fp32-level error) or 'native'
    (v_mfma_f32_16x16x4_f32 everywhere).  ``None`` restores the default.  No effect in bf16 mode.  Reachable from the training
    path as ``config.train.fp32_products`` (model/build.py) and ``VAETrainStep(fp32_products=...)``; bench.py times the
    default and reports the other form under 'extra'."""
    global _F32_SPLIT
    if mode is None:
        mode = DEFAULT_FP32_PRODUCTS
    if mode not in ('native', 'bf16x6'):
        raise ValueError(f"unknown fp32 product mode {mode!r}")
    _F32_SPLIT = mode == 'bf16x6'


def fp32_products():
    return 'bf16x6' if _F32_SPLIT else 'native'


def compute_mode():
    """(compute dtype, fp32 product form): what a weight shadow was laid out for (model/layer.py keeps it next to the
    shadows it saves for the backward pass)."""
    return (compute_dtype(), fp32_products())


def _flags():
    """Compute-mode bits of every descriptor built now."""
    return _COMPUTE_FLAGS | (PGV_COMPUTE_F32_SPLIT if (_F32_SPLIT and not (_COMPUTE_FLAGS & PGV_COMPUTE_BF16)) else 0)


class ConvGeom:
    """Geometry of one strided convolution between a big [B,Cb,Hb,Wb] and a small [B,Cs,Hs,Ws] tensor."""

    def __init__(self, Cb, Cs, k, stride, pad, Hb, Wb):
        self.Cb, self.Cs, self.k, self.stride, self.pad, self.Hb, self.Wb = Cb, Cs, k, stride, pad, Hb, Wb
        self.Hs = (Hb + 2 * pad - k) // stride + 1
        self.Ws = (Wb + 2 * pad - k) // stride + 1
        self._descs = {}
        self._shadow_bytes = {}

    def desc(self, B, flags=0, w_shadow=None):
        """``w_shadow``: the tensor ``conv_weight_shadow`` returned for the call's weight (``pgv_conv_desc.w_shadow``)."""
        flags |= _flags()
        if w_shadow is not None:   # per call: the descriptor carries a pointer
            return ConvDesc(B, self.Cb, self.Hb, self.Wb, self.Cs, self.Hs, self.Ws, self.k, self.k, self.stride,
                            self.pad, flags, w_shadow.data_ptr())
        d = self._descs.get((B, flags))
        if d is None:
            d = ConvDesc(B, self.Cb, self.Hb, self.Wb, self.Cs, self.Hs, self.Ws, self.k, self.k, self.stride,
                         self.pad, flags, None)
            self._descs[(B, flags)] = d
        return d

    def shadow_bytes(self):
        """Bytes of the weight shadow of this layer in the current compute mode (0: none)."""
        fl = _flags()
        if fl == 0:
            return 0
        n = self._shadow_bytes.get(fl)
        if n is None:
            n = self._shadow_bytes[fl] = int(_lib.load().pgv_conv_weight_shadow_bytes(ctypes.byref(self.desc(1))))
        return n


def _fuse_arg(bwd_fuse, out):
    """``bwd_fuse`` = (a, coef, gbias, act, slope[, cls[, gbias_copies]]): the call's product is the gradient of the BatchNorm output of
    the next-lower block; its BatchNorm + activation backward ``act'(a) * (coef[0]*g + coef[1]*a + coef[2])`` is applied
    before the store, ``gbias`` (caller-cleared, may be None) receives the bias gradient and ``cls`` (optional,
    caller-cleared [C*4]) the sums of the result by (row parity, column parity) class (``pgv_bwd_fuse``)."""
    if bwd_fuse is None:
        return None
    a, coef, gbias, act, slope = bwd_fuse[:5]
    cls = bwd_fuse[5] if len(bwd_fuse) > 5 else None
    gbias_copies = int(bwd_fuse[6]) if len(bwd_fuse) > 6 else 0
    _chk(a, coef, gbias, cls)
    if gbias_copies not in (0, CLS_COPIES) or (gbias_copies and (gbias is None or gbias.numel() != CLS_COPIES * a.shape[1])):
        raise ValueError("bwd_fuse: gbias_copies must be 0 or CLS_COPIES, with gbias holding CLS_COPIES * C floats")
    if a.shape != out.shape:
        raise ValueError("bwd_fuse: saved activation and output shapes differ")
    if coef.numel() != 3 * a.shape[1] or (gbias is not None and not gbias_copies and gbias.numel() != a.shape[1]):
        raise ValueError("bwd_fuse: coef must hold 3*C floats and gbias C floats")
    if cls is not None and cls.numel() != CLS_COPIES * 4 * a.shape[1]:
        raise ValueError("bwd_fuse: cls must hold CLS_COPIES * 4 * C floats")
    return _lib.BwdFuse(_p(a), _p(coef), _p(gbias), int(act), float(slope), _p(cls), gbias_copies)


def bn_src(stats, n, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, scale, shift, mean,
           rstd, stats_copies=1):
    """``pgv_bn_src``: a BatchNorm whose statistics have been accumulated and whose finalize step (``bn_finalize`` with
    these arguments) is left to the kernel that consumes it (``conv_down`` / ``conv_up`` / ``dropout_fwd`` with
    ``in_bn=``).  Keeps the tensors alive; ``scale`` / ``shift`` / ``mean`` / ``rstd`` are outputs.  ``stats_copies`` =
    CLS_COPIES: ``stats`` is that many partial copies of [2C] (a forward conv called with ``stats_copies=True``)."""
    if stats.numel() < 2 * scale.numel() * max(1, int(stats_copies)):
        raise ValueError("bn_src: stats holds fewer than stats_copies * 2C values")
    _chk64(stats)
    _chk(gamma, beta, running_mean, running_var, scale, shift, mean, rstd)
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise ValueError("num_batches_tracked must be an int64 device tensor")
    s = _lib.BnSrc(_p(stats), int(n), _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
                   _p(running_var), _p(num_batches_tracked), _p(scale), _p(shift), _p(mean), _p(rstd), int(stats_copies))
    s._keep = (stats, gamma, beta, running_mean, running_var, num_batches_tracked, scale, shift, mean, rstd)
    return s


def bn_src_finalize(src):
    """The finalize step of a ``bn_src`` as a launch of its own (no consumer kernel to ride in)."""
    C = src._keep[6].numel()
    _lib.check(_lib.load().pgv_bn_finalize_src(ctypes.byref(src), C, _stream()), "pgv_bn_finalize_src")


def conv_weight_shadow(geom, w):
    """The bf16 shadow of weight ``w`` of layer ``geom`` (``pgv_conv_weight_shadow``) for ``w_shadow=`` of the conv calls
    that multiply by ``w`` until it changes, or None: fp32 mode, or the layer has no bf16-native kernels."""
    n = geom.shadow_bytes()
    if n == 0:
        return None
    _chk(w)
    sh = torch.empty(n, device=w.device, dtype=torch.uint8)
    _lib.check(_lib.load().pgv_conv_weight_shadow(ctypes.byref(geom.desc(1)), _p(w), _p(sh), _stream()),
               "pgv_conv_weight_shadow")
    return sh


def conv_weight_shadows(pairs):
    """``conv_weight_shadow`` for the (geom, w) pairs of a conv stack in ONE launch (``pgv_conv_weight_shadows``): a list with
    the shadow of every pair, None where the layer has none."""
    todo = [(i, g, w) for i, (g, w) in enumerate(pairs) if g.shadow_bytes() > 0]
    out = [None] * len(pairs)
    if not todo:
        return out
    if len(todo) > 8:
        for i, g, w in todo:
            out[i] = conv_weight_shadow(g, w)
        return out
    sizes = [(g.shadow_bytes() + 255) // 256 * 256 for _, g, _ in todo]
    buf = torch.empty(sum(sizes), device=todo[0][2].device, dtype=torch.uint8)
    n, off = len(todo), 0
    descs = (ctypes.POINTER(ConvDesc) * n)()
    ws, shs = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    for k, (i, g, w) in enumerate(todo):
        _chk(w)
        out[i] = buf[off:off + g.shadow_bytes()]
        descs[k] = ctypes.pointer(g.desc(1))
        ws[k], shs[k] = w.data_ptr(), out[i].data_ptr()
        off += sizes[k]
    _lib.check(_lib.load().pgv_conv_weight_shadows(n, descs, ws, shs, _stream()), "pgv_conv_weight_shadows")
    return out


def conv_down(geom, big, w, bias, act, slope, in_scale=None, in_shift=None, stats=None, out=None, prezeroed=False,
              bwd_fuse=None, in_bn=None, stats_copies=False, w_shadow=None):
    """``prezeroed``: ``stats`` already holds zeros (PGV_PREZEROED) - the call accumulates without clearing it.
    ``in_bn`` (a ``bn_src``): the input's BatchNorm, finalized by this call (``pgv_conv_down_bn``).  ``stats_copies``:
    ``stats`` is CLS_COPIES zeroed partial copies of [2C] (PGV_STATS_COPIES)."""
    B = big.shape[0]
    fl = int(prezeroed) | (PGV_STATS_COPIES if stats_copies else 0)
    if out is None:
        out = torch.empty((B, geom.Cs, geom.Hs, geom.Ws), device=big.device, dtype=torch.float32)
    _chk(big, w, bias, in_scale, in_shift, out)
    _chk64(stats)
    lib = _lib.load()
    if in_bn is not None:
        if bwd_fuse is not None or in_scale is not None:
            raise ValueError("conv_down: in_bn excludes in_scale / bwd_fuse")
        _lib.check(lib.pgv_conv_down_bn(ctypes.byref(geom.desc(B, fl, w_shadow)), _p(big), ctypes.byref(in_bn), _p(w),
                                        _p(bias), act, slope, _p(out), _p(stats), _stream()), "pgv_conv_down_bn")
        return out
    f = _fuse_arg(bwd_fuse, out)
    _lib.check(lib.pgv_conv_down_fused(ctypes.byref(geom.desc(B, fl, w_shadow)), _p(big), _p(in_scale),
                                       _p(in_shift), _p(w), _p(bias), act, slope, _p(out), _p(stats),
                                       None if f is None else ctypes.byref(f), _stream()), "pgv_conv_down")
    return out


def conv_up(geom, small, w, bias, act, slope, in_scale=None, in_shift=None, stats=None, out=None, prezeroed=False,
            bwd_fuse=None, in_bn=None, stats_copies=False, w_shadow=None):
    B = small.shape[0]
    fl = int(prezeroed) | (PGV_STATS_COPIES if stats_copies else 0)
    if out is None:
        out = torch.empty((B, geom.Cb, geom.Hb, geom.Wb), device=small.device, dtype=torch.float32)
    _chk(small, w, bias, in_scale, in_shift, out)
    _chk64(stats)
    lib = _lib.load()
    if in_bn is not None:
        if bwd_fuse is not None or in_scale is not None:
            raise ValueError("conv_up: in_bn excludes in_scale / bwd_fuse")
        _lib.check(lib.pgv_conv_up_bn(ctypes.byref(geom.desc(B, fl, w_shadow)), _p(small), ctypes.byref(in_bn), _p(w),
                                      _p(bias), act, slope, _p(out), _p(stats), _stream()), "pgv_conv_up_bn")
        return out
    f = _fuse_arg(bwd_fuse, out)
    _lib.check(lib.pgv_conv_up_fused(ctypes.byref(geom.desc(B, fl, w_shadow)), _p(small), _p(in_scale), _p(in_shift),
                                     _p(w), _p(bias), act, slope, _p(out), _p(stats),
                                     None if f is None else ctypes.byref(f), _stream()), "pgv_conv_up")
    return out


def conv_up_sq(geom, small, w, bias, act, slope, target, scale, gbias, loss_acc, cls, in_scale=None, in_shift=None,
               in_bn=None):
    """``conv_up`` of an output block + the squared-error criterion against ``target`` where the output is produced
    (``pgv_conv_up_sqerr``; upstream gradient of the criterion = 1): returns (out, g_y) and adds into ``gbias`` / ``loss_acc``
    / ``cls`` - or None when the shape / mode has no fused kernel (nothing was launched: the caller takes ``conv_up`` +
    ``sqerr_act_bwd``)."""
    B = small.shape[0]
    out = torch.empty((B, geom.Cb, geom.Hb, geom.Wb), device=small.device, dtype=torch.float32)
    g_y = torch.empty_like(out)
    _chk(small, w, bias, in_scale, in_shift, out, target, gbias, loss_acc, cls)
    fused = ctypes.c_int(0)
    _lib.check(_lib.load().pgv_conv_up_sqerr(ctypes.byref(geom.desc(B, 0, None)), _p(small),
                                             None if in_bn is None else ctypes.byref(in_bn), _p(in_scale), _p(in_shift),
                                             _p(w), _p(bias), act, slope, _p(out), _p(target), scale, _p(g_y), _p(gbias),
                                             _p(loss_acc), _p(cls), ctypes.byref(fused), _stream()), "pgv_conv_up_sqerr")
    return (out, g_y) if fused.value else None


_ws_cache = {}
_ws_retired = []   # superseded workspaces: a captured hipGraph may still write its partial gradients into them


def _workspace(device, nbytes):
    """Weight-gradient workspace of the CURRENT stream on ``device`` (pgv_hip.h: private to a call until it has completed
    on its stream - calls on different streams must not share one).  A buffer that is outgrown is kept alive instead of
    being returned to the allocator: graphs captured earlier replay into its address."""
    if nbytes <= 0:
        return None
    key = (device.index, _stream())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _ws_retired.append(ws)
        ws = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _ws_cache[key] = ws
    return ws


def coef_scratch(geom, lower_is_big):
    """float64 elements of a ``coef_req`` scratch (``pgv_coef_req.scratch``): the tap sums of the output gradient (as
    partial copies when its channels are few) + 1."""
    c_gy = geom.Cs if lower_is_big else geom.Cb
    return max(c_gy * geom.k * geom.k, 1024) + 1


def conv_wgrad(geom, big, small, gw, big_scale=None, big_shift=None, small_scale=None, small_shift=None,
               prezeroed=False, coef_req=None, bias_finish=None):
    """``prezeroed``: ``gw`` already holds zeros (e.g. a slice of the zero_grad'ed flat gradient buffer).
    ``coef_req``: also the BatchNorm-backward coefficients of the block below (``pgv_conv_wgrad_coef``) - a dict with the
    fields of ``pgv_coef_req``; ``scratch`` is zeroed float64 of ``coef_scratch(geom, lower_is_big)`` elements; ``cls_copies``
    (optional) = CLS_COPIES when ``cls`` of a Conv2d consumer is a bias gradient kept as partial copies.
    ``bias_finish`` = (copies [CLS_COPIES * C], gbias [C], accumulate): also this block's bias gradient from the partial
    copies its producer kept (``pgv_bias_req``), in the reduce launch of the weight gradient."""
    B = big.shape[0]
    _chk(big, small, gw, big_scale, big_shift, small_scale, small_shift)
    lib = _lib.load()
    d = geom.desc(B, int(prezeroed))
    nbytes = lib.pgv_conv_wgrad_workspace(ctypes.byref(d))
    ws = _workspace(big.device, nbytes)
    if coef_req is None and bias_finish is None:
        _lib.check(lib.pgv_conv_wgrad(ctypes.byref(d), _p(big), _p(big_scale), _p(big_shift), _p(small),
                                      _p(small_scale), _p(small_shift), _p(gw), _p(ws), nbytes, _stream()),
                   "pgv_conv_wgrad")
        return gw
    req = bias = None
    if coef_req is not None:
        r = coef_req
        lower_is_big = bool(r['lower_is_big'])
        _chk(r['cls'], r['w'], r['scale'], r['shift'], r['mean'], r['rstd'], r['coef'], r.get('ggamma'), r.get('gbeta'))
        _chk64(r['scratch'])
        c_low = geom.Cb if lower_is_big else geom.Cs
        if r['scratch'].numel() < coef_scratch(geom, lower_is_big):
            raise ValueError("conv_wgrad: coef_req scratch needs coef_scratch(geom, lower_is_big) doubles")
        if r['coef'].numel() < 3 * c_low:
            raise ValueError("conv_wgrad: coef_req coef needs 3*C_lower floats")
        req = _lib.CoefReq(int(lower_is_big), _p(r['cls']), _p(r['w']), _p(r['scale']), _p(r['shift']), _p(r['mean']),
                           _p(r['rstd']), int(r['n']), _p(r['coef']), _p(r.get('ggamma')), _p(r.get('gbeta')),
                           _p(r['scratch']), int(r.get('cls_copies', 0)))
    if bias_finish is not None:
        copies, gb, acc = bias_finish
        _chk(copies, gb)
        if copies.numel() != CLS_COPIES * gb.numel():
            raise ValueError("conv_wgrad: bias_finish copies must hold CLS_COPIES * C floats")
        bias = _lib.BiasReq(_p(copies), _p(gb), gb.numel(), int(bool(acc)))
    _lib.check(lib.pgv_conv_wgrad_ex(ctypes.byref(d), _p(big), _p(big_scale), _p(big_shift), _p(small), _p(small_scale),
                                     _p(small_shift), _p(gw), _p(ws), nbytes, None if req is None else ctypes.byref(req),
                                     None if bias is None else ctypes.byref(bias), _stream()), "pgv_conv_wgrad_ex")
    return gw


def bn_stats(a, stats):
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // max(1, B * C)
    _chk(a)
    _chk64(stats)
    _lib.check(_lib.load().pgv_bn_stats(_p(a), B, C, HW, _p(stats), _stream()), "pgv_bn_stats")


def bn_finalize(stats, n, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, rstd,
                num_batches_tracked=None):
    C = stats.numel() // 2
    _chk64(stats)
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise ValueError("num_batches_tracked must be an int64 device tensor")
    _lib.check(_lib.load().pgv_bn_finalize(_p(stats), C, n, _p(gamma), _p(beta), eps, momentum, _p(running_mean),
                                           _p(running_var), _p(num_batches_tracked), _p(scale), _p(shift), _p(mean),
                                           _p(rstd), _stream()), "pgv_bn_finalize")


def bn1d_fwd(x, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, y, scale, mean, rstd):
    """Train-mode nn.BatchNorm1d forward over x[B, C] in one launch (``pgv_bn1d_fwd``)."""
    B, C = x.shape
    _chk(x, gamma, beta, running_mean, running_var, y, scale, mean, rstd)
    _lib.check(_lib.load().pgv_bn1d_fwd(_p(x), B, C, _p(gamma), _p(beta), eps, momentum, _p(running_mean),
                                        _p(running_var), _p(num_batches_tracked), _p(y), _p(scale), _p(mean), _p(rstd),
                                        _stream()), "pgv_bn1d_fwd")
    return y


def bn1d_bwd(g, x, scale, mean, rstd, gx, ggamma, gbeta):
    """Train-mode nn.BatchNorm1d backward over [B, C] in one launch (``pgv_bn1d_bwd``)."""
    B, C = x.shape
    _chk(g, x, scale, mean, rstd, gx, ggamma, gbeta)
    _lib.check(_lib.load().pgv_bn1d_bwd(_p(g), _p(x), _p(scale), _p(mean), _p(rstd), B, C, _p(gx), _p(ggamma),
                                        _p(gbeta), _stream()), "pgv_bn1d_bwd")
    return gx


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, scale, shift):
    C = running_mean.numel()
    _chk(gamma, beta, running_mean, running_var, scale, shift)
    _lib.check(_lib.load().pgv_bn_eval_affine(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, C, _p(scale),
                                              _p(shift), _stream()), "pgv_bn_eval_affine")


def affine_nchw(a, scale, shift, out=None):
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // max(1, B * C)
    if out is None:
        out = torch.empty_like(a)
    _chk(a, scale, shift, out)
    _lib.check(_lib.load().pgv_affine_nchw(_p(a), _p(scale), _p(shift), B, C, HW, _p(out), _stream()),
               "pgv_affine_nchw")
    return out


def conv_tap_sums(geom, gy, gy_is_big, T=None, prezeroed=False, cls=None):
    """T[c][kh][kw] (float64) = sums of ``gy`` over the positions each kernel tap pairs with the inside of the other
    tensor of ``geom`` (``pgv_conv_tap_sums``).  ``cls``: class sums of ``gy`` (``conv_class_sums`` or the bias gradient
    of the block that owns a small-side ``gy``): only the border rows / columns of ``gy`` are read then."""
    B = gy.shape[0]
    C = geom.Cb if gy_is_big else geom.Cs
    if T is None:
        T = torch.empty(C * geom.k * geom.k, device=gy.device, dtype=torch.float64)
        prezeroed = False
    _chk(gy, cls)
    _chk64(T)
    m = geom.stride if gy_is_big else 1
    if cls is not None and cls.numel() != C * m * m * (CLS_COPIES if gy_is_big else 1):
        raise ValueError("conv_tap_sums: cls must hold [CLS_COPIES,] C * m * m floats")
    _lib.check(_lib.load().pgv_conv_tap_sums(ctypes.byref(geom.desc(B)), int(gy_is_big), _p(gy), _p(cls), _p(T),
                                             PGV_PREZEROED if prezeroed else 0, _stream()), "pgv_conv_tap_sums")
    return T


def conv_class_sums(geom, gy, gy_is_big, cls=None, prezeroed=False):
    """Per-channel sums of ``gy`` by (row mod m, column mod m) class, m = stride for the big tensor, 1 for the small one
    (``pgv_conv_class_sums``)."""
    B = gy.shape[0]
    C = geom.Cb if gy_is_big else geom.Cs
    m = geom.stride if gy_is_big else 1
    if cls is None:
        cls = torch.empty(C * m * m * (CLS_COPIES if gy_is_big else 1), device=gy.device, dtype=torch.float32)
        prezeroed = False
    _chk(gy, cls)
    _lib.check(_lib.load().pgv_conv_class_sums(ctypes.byref(geom.desc(B)), int(gy_is_big), _p(gy), _p(cls),
                                               PGV_PREZEROED if prezeroed else 0, _stream()), "pgv_conv_class_sums")
    return cls


def bn_bwd_coef(geom, B, lower_is_big, w, gw, T, scale, shift, mean, rstd, n, coef, ggamma=None, gbeta=None):
    """BatchNorm-backward coefficients of the block below ``geom``'s block from that block's weights, weight gradient
    and tap sums (``pgv_bn_bwd_coef``); also writes the BatchNorm parameter gradients."""
    _chk(w, gw, scale, shift, mean, rstd, coef, ggamma, gbeta)
    _chk64(T)
    _lib.check(_lib.load().pgv_bn_bwd_coef(ctypes.byref(geom.desc(B)), int(lower_is_big), _p(w), _p(gw), _p(T),
                                           _p(scale), _p(shift), _p(mean), _p(rstd), int(n), _p(coef), _p(ggamma),
                                           _p(gbeta), _stream()), "pgv_bn_bwd_coef")
    return coef


def bn_bwd_coef_from_gy(geom, lower_is_big, gy, cls, T, w, gw, scale, shift, mean, rstd, n, coef, ggamma=None, gbeta=None,
                        prezeroed=False):
    """Tap sums of ``gy`` (border form, from its class sums ``cls``) + BatchNorm-backward coefficients in one launch
    where possible (``pgv_bn_bwd_coef_from_gy``).  ``T``: float64 scratch of C_gy*k*k + 1 elements."""
    B = gy.shape[0]
    _chk(gy, cls, w, gw, scale, shift, mean, rstd, coef, ggamma, gbeta)
    _chk64(T)
    C = geom.Cs if lower_is_big else geom.Cb
    if T.numel() < C * geom.k * geom.k + 1:
        raise ValueError("bn_bwd_coef_from_gy: T needs C*k*k + 1 doubles")
    _lib.check(_lib.load().pgv_bn_bwd_coef_from_gy(ctypes.byref(geom.desc(B)), int(lower_is_big), _p(gy), _p(cls), _p(T),
                                                   _p(w), _p(gw), _p(scale), _p(shift), _p(mean), _p(rstd), int(n),
                                                   _p(coef), _p(ggamma), _p(gbeta), PGV_PREZEROED if prezeroed else 0,
                                                   _stream()), "pgv_bn_bwd_coef_from_gy")
    return coef


def act_bwd_coef(g, a, coef, act, slope, g_y, gbias, prezeroed=False):
    """g_y = act'(a) * (coef[0]*g + coef[1]*a + coef[2]), gbias (+)= sum g_y (``pgv_act_bwd_coef``)."""
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // max(1, B * C)
    _chk(g, a, coef, g_y, gbias)
    _lib.check(_lib.load().pgv_act_bwd_coef(_p(g), _p(a), _p(coef), B, C, HW, act, slope, _p(g_y), _p(gbias),
                                            PGV_PREZEROED if prezeroed else 0, _stream()), "pgv_act_bwd_coef")


def bn_bwd_reduce(g_o, a, mean, rstd, red, prezeroed=False):
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // max(1, B * C)
    _chk(g_o, a, mean, rstd)
    _chk64(red)
    _lib.check(_lib.load().pgv_bn_bwd_reduce(_p(g_o), _p(a), _p(mean), _p(rstd), B, C, HW, _p(red), int(prezeroed),
                                             _stream()), "pgv_bn_bwd_reduce")


def act_bn_bwd(g_o, a, scale, mean, rstd, red, act, slope, g_y, gbias, ggamma=None, gbeta=None, prezeroed=False):
    """``prezeroed`` refers to ``gbias``; ``ggamma`` / ``gbeta`` (train-mode BN only) are plain float32 stores."""
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // max(1, B * C)
    _chk(g_o, a, scale, mean, rstd, g_y, gbias, ggamma, gbeta)
    _chk64(red)
    _lib.check(_lib.load().pgv_act_bn_bwd(_p(g_o), _p(a), _p(scale), _p(mean), _p(rstd), _p(red), B, C, HW, act, slope,
                                          _p(g_y), _p(gbias), _p(ggamma), _p(gbeta), int(prezeroed), _stream()),
               "pgv_act_bn_bwd")


def bn_act_bwd_fusable(B, C, HW):
    """Whether ``bn_act_bwd_fused`` serves this shape (else: ``bn_bwd_reduce`` + ``act_bn_bwd``)."""
    return bool(_lib.load().pgv_bn_act_bwd_fusable(int(B), int(C), int(HW)))


def bn_act_bwd_fused(g_o, a, scale, mean, rstd, act, slope, g_y, gbias, ggamma=None, gbeta=None, prezeroed=False):
    """``bn_bwd_reduce`` + ``act_bn_bwd`` of a train-mode BatchNorm as one launch (``pgv_bn_act_bwd_fused``, small planes)."""
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // max(1, B * C)
    _chk(g_o, a, scale, mean, rstd, g_y, gbias, ggamma, gbeta)
    _lib.check(_lib.load().pgv_bn_act_bwd_fused(_p(g_o), _p(a), _p(scale), _p(mean), _p(rstd), B, C, HW, act, slope, _p(g_y),
                                                _p(gbias), _p(ggamma), _p(gbeta), int(prezeroed), _stream()),
               "pgv_bn_act_bwd_fused")


def sqerr_act_bwd(a, x, g_loss, scale, act, slope, g_y, gbias, prezeroed=False, loss_acc=None, cls=None):
    """g_y = act'(a) * 2 scale g_loss (a - x), gbias += sum over (batch, pixels): squared-error criterion + output
    activation of a block without BatchNorm, backward in one pass.  ``loss_acc`` (zeroed scalar) += the criterion.
    ``cls`` (zeroed [CLS_COPIES * 4], single-channel tensors): += the sums of g_y by (row parity, column parity) class, spread
    over partial copies."""
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // max(1, B * C)
    _chk(a, x, g_loss, g_y, gbias, loss_acc, cls)
    if cls is not None:
        _lib.check(_lib.load().pgv_sqerr_act_bwd_cls(_p(a), _p(x), _p(g_loss), scale, B, C, HW, a.shape[-1], act, slope,
                                                     _p(g_y), _p(gbias), _p(loss_acc), _p(cls),
                                                     PGV_PREZEROED if prezeroed else 0, _stream()),
                   "pgv_sqerr_act_bwd_cls")
        return
    _lib.check(_lib.load().pgv_sqerr_act_bwd(_p(a), _p(x), _p(g_loss), scale, B, C, HW, act, slope, _p(g_y), _p(gbias),
                                             _p(loss_acc), PGV_PREZEROED if prezeroed else 0, _stream()),
               "pgv_sqerr_act_bwd")


def gemm(M, N, K, A, sam, sak, Bm, sbk, sbn, C, ldc, bias_n=None, prezeroed=False):
    """``prezeroed``: C holds zeros (PGV_PREZEROED) - a split-K product accumulates into it without a clearing launch."""
    _chk(A, Bm, C, bias_n)
    _lib.check(_lib.load().pgv_gemm(M, N, K, _p(A), sam, sak, _p(Bm), sbk, sbn, _p(C), ldc, _p(bias_n),
                                    _COMPUTE_FLAGS | (PGV_PREZEROED if prezeroed else 0), None, 0, _stream()),
               "pgv_gemm")
    return C


def linear_fwd(x, w, bias, out=None):
    """y[M,N] = x[M,K] @ w[N,K]^T + bias  (nn.Linear).  ``out``: a ZEROED [M,N] buffer to accumulate into."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=torch.float32) if out is None else out
    return gemm(M, N, K, x, K, 1, w, 1, K, y, N, bias, prezeroed=out is not None)


def linear_dgrad(gy, w, out=None):
    """gx[M,K] = gy[M,N] @ w[N,K].  ``out``: a ZEROED [M,K] buffer to accumulate into."""
    M, N = gy.shape
    K = w.shape[1]
    gx = torch.empty((M, K), device=gy.device, dtype=torch.float32) if out is None else out
    return gemm(M, K, N, gy, N, 1, w, K, 1, gx, K, prezeroed=out is not None)


def linear_wgrad(gy, x, gw, prezeroed=False):
    """gw[N,K] = gy[M,N]^T @ x[M,K]."""
    M, N = gy.shape
    K = x.shape[1]
    return gemm(N, K, M, gy, 1, N, x, K, 1, gw, K, prezeroed=prezeroed)


def colsum(x, out, prezeroed=False):
    M, N = x.shape
    _chk(x, out)
    _lib.check(_lib.load().pgv_colsum(_p(x), M, N, N, _p(out), PGV_PREZEROED if prezeroed else 0, _stream()),
               "pgv_colsum")
    return out


def dropout_mask(rng_state, stream_id, p, n, device):
    mask = torch.empty(n, device=device, dtype=torch.float32)
    _lib.check(_lib.load().pgv_dropout_mask(rng_state.data_ptr(), stream_id, p, n, _p(mask), _stream()),
               "pgv_dropout_mask")
    return mask


def dropout_apply(rng_state, stream_id, p, x):
    """(x * mask, mask) with the keep mask drawn on the fly (one pass instead of mask generation + multiply)."""
    _chk(x)
    y, mask = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.load().pgv_dropout_apply(rng_state.data_ptr(), stream_id, p, x.numel(), _p(x), _p(y), _p(mask),
                                             _stream()), "pgv_dropout_apply")
    return y, mask


def dropout_fwd(rng_state, stream_id, p, x, scale=None, shift=None, in_bn=None):
    """nn.Dropout forward without a stored mask (``pgv_dropout_fwd``): returns (y, saved_state) - ``saved_state`` is the
    copy of the generator state ``dropout_bwd`` regenerates the mask from.  ``scale`` / ``shift`` ([C], x is [B, C, ...]):
    a per-channel affine applied on the way in; ``in_bn`` (a ``bn_src``) instead: the BatchNorm that affine comes from,
    finalized by this call."""
    _chk(x, scale, shift)
    B = x.shape[0]
    C = x.shape[1] if (scale is not None or in_bn is not None) else 1
    HW = x.numel() // max(1, B * C)
    y = torch.empty_like(x)
    saved = torch.empty(2, device=x.device, dtype=torch.int64)
    if in_bn is not None:
        _lib.check(_lib.load().pgv_dropout_fwd_bn(rng_state.data_ptr(), stream_id, p, _p(x), B, C, max(1, HW),
                                                  ctypes.byref(in_bn), _p(y), saved.data_ptr(), _stream()),
                   "pgv_dropout_fwd_bn")
        return y, saved
    _lib.check(_lib.load().pgv_dropout_fwd(rng_state.data_ptr(), stream_id, p, _p(x), B, C, max(1, HW), _p(scale),
                                           _p(shift), _p(y), saved.data_ptr(), _stream()), "pgv_dropout_fwd")
    return y, saved


def dropout_bwd_bn_reduce(saved_state, stream_id, p, g_d, a, mean, rstd, red, prezeroed=False):
    """``dropout_bwd`` of ``g_d`` (shaped like ``a``: [B, C, H, W]) plus ``bn_bwd_reduce`` of the result against ``a`` in the
    same pass (``pgv_dropout_bwd_bn_reduce``): returns gx, accumulates ``red``."""
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // max(1, B * C)
    _chk(g_d, a, mean, rstd)
    _chk64(red)
    gx = torch.empty_like(a)
    _lib.check(_lib.load().pgv_dropout_bwd_bn_reduce(saved_state.data_ptr(), stream_id, p, _p(g_d), _p(a), _p(mean),
                                                     _p(rstd), B, C, HW, _p(gx), _p(red), int(prezeroed), _stream()),
               "pgv_dropout_bwd_bn_reduce")
    return gx


def dropout_bwd(saved_state, stream_id, p, gy, colsum=None, prezeroed=False):
    """gx = gy * mask (regenerated); ``colsum`` ([N], gy is [M, N]): also colsum (+)= gx.sum(0) in the same pass
    (``pgv_dropout_bwd_colsum``; ``prezeroed``: it already holds zeros)."""
    _chk(gy, colsum)
    gx = torch.empty_like(gy)
    if colsum is not None:
        M, N = gy.shape
        _lib.check(_lib.load().pgv_dropout_bwd_colsum(saved_state.data_ptr(), stream_id, p, M, N, _p(gy), _p(gx),
                                                      _p(colsum), PGV_PREZEROED if prezeroed else 0, _stream()),
                   "pgv_dropout_bwd_colsum")
        return gx
    _lib.check(_lib.load().pgv_dropout_bwd(saved_state.data_ptr(), stream_id, p, gy.numel(), _p(gy), _p(gx), _stream()),
               "pgv_dropout_bwd")
    return gx


def normal(rng_state, stream_id, shape, device):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    _lib.check(_lib.load().pgv_normal(rng_state.data_ptr(), stream_id, out.numel(), _p(out), _stream()), "pgv_normal")
    return out


def rng_advance(rng_state, inc):
    _lib.check(_lib.load().pgv_rng_advance(rng_state.data_ptr(), inc, _stream()), "pgv_rng_advance")


def mul(x, m, out=None):
    if out is None:
        out = torch.empty_like(x)
    _chk(x, m, out)
    _lib.check(_lib.load().pgv_mul(_p(x), _p(m), x.numel(), _p(out), _stream()), "pgv_mul")
    return out


def reparam_kl_fwd(ml, eps, kl_scale, want_z=True):
    B, _, D = ml.shape
    _chk(ml, eps)
    z = torch.empty((B, D), device=ml.device, dtype=torch.float32) if want_z else None
    kl = torch.empty((), device=ml.device, dtype=torch.float32)
    _lib.check(_lib.load().pgv_reparam_kl_fwd(_p(ml), _p(eps), B, D, kl_scale, _p(z), _p(kl), _stream()),
               "pgv_reparam_kl_fwd")
    return z, kl


def reparam_kl_fwd_rng(ml, rng_state, stream_id, kl_scale, kl=None):
    """``reparam_kl_fwd`` with eps drawn inside the kernel (``pgv_reparam_kl_fwd_rng``): returns (z, kl, eps).  ``kl``: a
    ZEROED 0-d / 1-element buffer to accumulate the Dkl term into (else one is cleared here)."""
    B, _, D = ml.shape
    _chk(ml, kl)
    z = torch.empty((B, D), device=ml.device, dtype=torch.float32)
    eps = torch.empty((B, D), device=ml.device, dtype=torch.float32)
    out = torch.empty((), device=ml.device, dtype=torch.float32) if kl is None else kl
    _lib.check(_lib.load().pgv_reparam_kl_fwd_rng(_p(ml), rng_state.data_ptr(), stream_id, B, D, kl_scale, _p(z), _p(eps),
                                                  _p(out), PGV_PREZEROED if kl is not None else 0, _stream()),
               "pgv_reparam_kl_fwd_rng")
    return z, out.reshape(()), eps


def bn1d_reparam_fwd(x, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, rng_state, stream_id,
                     kl_scale, kl=None):
    """Train-mode nn.BatchNorm1d over x[B, 2D] + reparameterisation + Dkl in one launch (``pgv_bn1d_reparam_fwd``):
    returns (y, scale, mean, rstd, z, kl, eps)."""
    B, C = x.shape
    D = C // 2
    _chk(x, gamma, beta, running_mean, running_var, kl)
    dev = x.device
    y = torch.empty_like(x)
    vec = torch.empty(3 * C, device=dev, dtype=torch.float32)
    z = torch.empty((B, D), device=dev, dtype=torch.float32)
    e = torch.empty((B, D), device=dev, dtype=torch.float32)
    out = torch.empty((), device=dev, dtype=torch.float32) if kl is None else kl
    _lib.check(_lib.load().pgv_bn1d_reparam_fwd(_p(x), B, D, _p(gamma), _p(beta), eps, momentum, _p(running_mean),
                                                _p(running_var), _p(num_batches_tracked), _p(y), _p(vec[:C]),
                                                _p(vec[C:2 * C]), _p(vec[2 * C:]), rng_state.data_ptr(), stream_id,
                                                kl_scale, _p(z), _p(e), _p(out),
                                                PGV_PREZEROED if kl is not None else 0, _stream()),
               "pgv_bn1d_reparam_fwd")
    return y, vec[:C], vec[C:2 * C], vec[2 * C:], z, out.reshape(()), e


def bn1d_reparam_bwd(g_z, g_kl, g_y, y, eps, x, scale, mean, rstd, kl_scale, gx, ggamma, gbeta, colsum=None,
                     colsum_accumulate=False):
    B, C = x.shape
    _chk(g_z, g_kl, g_y, y, eps, x, scale, mean, rstd, gx, ggamma, gbeta, colsum)
    _lib.check(_lib.load().pgv_bn1d_reparam_bwd(_p(g_z), _p(g_kl), _p(g_y), _p(y), _p(eps), _p(x), _p(scale), _p(mean),
                                                _p(rstd), B, C // 2, kl_scale, _p(gx), _p(ggamma), _p(gbeta), _p(colsum),
                                                PGV_PREZEROED if colsum_accumulate else 0, _stream()),
               "pgv_bn1d_reparam_bwd")
    return gx


def reparam_kl_bwd(ml, eps, g_z, g_kl, kl_scale):
    B, _, D = ml.shape
    _chk(ml, eps, g_z, g_kl)
    g_ml = torch.empty_like(ml)
    _lib.check(_lib.load().pgv_reparam_kl_bwd(_p(ml), _p(eps), _p(g_z), _p(g_kl), B, D, kl_scale, _p(g_ml), _stream()),
               "pgv_reparam_kl_bwd")
    return g_ml


def sqerr_fwd(xhat, x, scale):
    _chk(xhat, x)
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().pgv_sqerr_fwd(_p(xhat), _p(x), x.numel(), scale, _p(loss), _stream()), "pgv_sqerr_fwd")
    return loss


def sqerr_bwd(xhat, x, g_loss, scale, hardtanh=False):
    _chk(xhat, x, g_loss)
    g = torch.empty_like(xhat)
    _lib.check(_lib.load().pgv_sqerr_bwd(_p(xhat), _p(x), _p(g_loss), x.numel(), scale, int(hardtanh), _p(g),
                                         _stream()), "pgv_sqerr_bwd")
    return g


def adam_step(p, g, m, v, hyper, beta1, beta2, eps, weight_decay):
    _chk(p, g, m, v, hyper)
    _lib.check(_lib.load().pgv_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), beta1, beta2, eps,
                                         weight_decay, _stream()), "pgv_adam_step")


STFT_DB, STFT_LINEAR, STFT_COMPLEX = 0, 1, 2


def stft_mel(wav, hop, n_frames, window, norm, mel_csr, n_mels, floor_lin, affine_a, affine_b, out=None, mode=STFT_DB):
    """``mode``: STFT_DB (clamped dB + affine), STFT_LINEAR (normalised amplitudes), STFT_COMPLEX (complex64
    [B, 513, n_frames] un-normalised one-sided STFT; no mel projection) - ``pgv_stft`` of include/pgv_hip.h."""
    B, n = wav.shape
    _chk(wav, window)
    rows = n_mels if n_mels > 0 else 513
    if mode == STFT_COMPLEX:
        if n_mels > 0 or out is not None:
            raise ValueError("stft_mel: the complex STFT has no mel projection / caller-provided output")
        res = torch.empty((B, rows, n_frames), device=wav.device, dtype=torch.complex64)
        out = torch.view_as_real(res)
    elif out is None:
        res = out = torch.empty((B, rows, n_frames), device=wav.device, dtype=torch.float32)
    else:
        _chk(out)
        if out.numel() != B * rows * n_frames:
            raise ValueError(f"stft_mel: out has {out.numel()} elements, expected {B}x{rows}x{n_frames}")
        res = out
    rp, col, val = mel_csr if mel_csr is not None else (None, None, None)
    _lib.check(_lib.load().pgv_stft(_p(wav), B, n, 1024, hop, n_frames, _p(window), norm,
                                    None if rp is None else rp.data_ptr(), None if col is None else col.data_ptr(),
                                    _p(val), n_mels, mode, floor_lin, affine_a, affine_b, out.data_ptr(), _stream()),
               "pgv_stft")
    return res


def fill(t, v):
    _chk(t)
    _lib.check(_lib.load().pgv_fill(_p(t), t.numel(), v, _stream()), "pgv_fill")


def copy(src, dst):
    _chk(src, dst)
    _lib.check(_lib.load().pgv_copy(_p(src), _p(dst), src.numel(), _stream()), "pgv_copy")


# ---- preset-parameter losses and metrics -------------------------------------------------------------
PARAMS_CCE, PARAMS_CCE_SOFTMAX, PARAMS_BCE = _lib.PGV_PARAMS_CCE, _lib.PGV_PARAMS_CCE_SOFTMAX, _lib.PGV_PARAMS_BCE
_PARAMS_WS = {}


def params_tables(device, num_idx, cat_groups, rules):
    """``pgv_params_tables`` of a PresetIndexesHelper on ``device``: ``num_idx`` the numerical learnable columns,
    ``cat_groups`` the one-hot groups (lists of columns), ``rules`` = [(trigger column, [numerical columns], [first
    columns of groups]), ...] (data/preset.py:259-281).  The returned structure keeps its device tensors alive."""
    if len(rules) > 32:
        raise ValueError("at most 32 useless-parameter rules")
    G, K = len(cat_groups), max((len(g) for g in cat_groups), default=1)
    i32 = dict(dtype=torch.int32, device=device)
    cat = torch.full((max(G, 1), K), -1, dtype=torch.int32)
    for gi, g in enumerate(cat_groups):
        cat[gi, :len(g)] = torch.tensor(list(g), dtype=torch.int32)
    first_to_group = {g[0]: gi for gi, g in enumerate(cat_groups)}
    num_pos = {c: i for i, c in enumerate(num_idx)}
    num_rules, cat_rules = [0] * max(len(num_idx), 1), [0] * max(G, 1)
    for r, (_, nums, cats) in enumerate(rules):
        for n in nums:
            if n in num_pos:
                num_rules[num_pos[n]] |= 1 << r
        for c in cats:
            if c in first_to_group:
                cat_rules[first_to_group[c]] |= 1 << r
    keep = (torch.tensor(list(num_idx) or [0], **i32), torch.tensor(num_rules, dtype=torch.int64).to(**i32),
            cat.to(device), torch.tensor(cat_rules, dtype=torch.int64).to(**i32),
            torch.tensor([r[0] for r in rules] or [0], **i32))
    t = _lib.ParamsTables(len(num_idx), _p(keep[0]), _p(keep[1]), G, K, _p(keep[2]), _p(keep[3]), len(rules), _p(keep[4]))
    t._keep = keep
    return t


def params_loss(u_out, u_in, tables, mode, softmax_t, normalize, cat_factor, want_grad=True):
    """``pgv_params_loss``: (loss 0-d tensor, d loss / d u_out or None) of SynthParamsLoss in one launch."""
    _chk(u_out, u_in)
    if u_out.shape != u_in.shape or u_out.dim() != 2:
        raise ValueError("params_loss: u_out and u_in must be [B, L] tensors of the same shape")
    B, L = u_in.shape
    key = (u_in.device, _stream())
    need = 8 * (1 + min(B, 1024))
    ws = _PARAMS_WS.get(key)
    if ws is None or ws.numel() < need:
        # zeroed once: the arrival counter in its first word is left at zero by every call.  A superseded buffer stays
        # alive (a captured graph may still replay into it).
        _PARAMS_WS.setdefault('_old', []).append(ws)
        ws = torch.zeros(max(need, 8 * 1025), dtype=torch.uint8, device=u_in.device)
        _PARAMS_WS[key] = ws
    loss = torch.empty((), device=u_in.device, dtype=torch.float32)
    grad = torch.empty_like(u_out) if want_grad else None
    _lib.check(_lib.load().pgv_params_loss(_p(u_out), _p(u_in), B, L, ctypes.byref(tables), int(mode), float(softmax_t),
                                           int(bool(normalize)), float(cat_factor), _p(loss), _p(grad), _p(ws), ws.numel(),
                                           _stream()), "pgv_params_loss")
    return loss, grad


def params_item_tables(device, items):
    """Item tables of ``pgv_params_columns``: ``items`` = [(kind, column or list of columns, cardinal), ...]."""
    kind, first, length, card, idx = [], [], [], [], []
    for k, cols, c in items:
        kind.append(int(k))
        if isinstance(cols, (list, tuple)):
            first.append(len(idx))
            length.append(len(cols))
            idx += [int(v) for v in cols]
        else:
            first.append(int(cols))
            length.append(1)
        card.append(float(c))
    i32 = dict(dtype=torch.int32, device=device)
    return (torch.tensor(kind or [0], **i32), torch.tensor(first or [0], **i32), torch.tensor(length or [0], **i32),
            torch.tensor(card or [0.0], dtype=torch.float32, device=device), torch.tensor(idx or [0], **i32), len(items))


def params_columns(u_out, u_in, item_tables, want_cols=True, want_match=True):
    """``pgv_params_columns``: (in_cols [B, n], out_cols [B, n], match [n]) for the items of ``params_item_tables``."""
    _chk(u_out, u_in)
    kind, first, length, card, idx, n = item_tables
    B, L = u_in.shape
    f32 = dict(device=u_in.device, dtype=torch.float32)
    in_cols = torch.empty((B, n), **f32) if want_cols else None
    out_cols = torch.empty((B, n), **f32) if want_cols else None
    match = torch.empty((n,), **f32) if want_match else None
    _lib.check(_lib.load().pgv_params_columns(_p(u_out), _p(u_in), B, L, n, _p(kind), _p(first), _p(length), _p(card),
                                              _p(idx), _p(in_cols), _p(out_cols), _p(match), _stream()),
               "pgv_params_columns")
    return in_cols, out_cols, match
