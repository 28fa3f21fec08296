"""Conv building blocks of the spectrogram VAE, MI355X-native.

Mirrors the surface of the reference's ``model/layer.py`` (``Conv2D`` :10-26, ``TConv2D`` :29-46: conv -> activation
-> BatchNorm2d, BN *after* the activation, ``batch_norm=None`` drops it) with the same sub-module names, so
state-dict keys are identical (``...enc2conv.weight``, ``...enc2bn.running_mean``, ``...dec2tconv.weight``).
``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.BatchNorm2d`` objects are kept purely as *parameter containers*
(torch-default initialisation, reference key names); their ``forward`` is never called.  All arithmetic runs in the
HIP kernels behind ``include/pgv_hip.h`` through :class:`ConvStackFn`:

* forward of block l:  a_l = act(conv(o_{l-1}) + b)  with the producer's BatchNorm folded into the consumer's load
  (o_{l-1} = a_{l-1}*scale+shift inside the image, 0 in the zero padding), BN batch statistics accumulated in the
  conv epilogue, ``pgv_bn_finalize`` turning them into (scale, shift, mean, rstd) and updating the running stats;
* backward: ``pgv_bn_bwd_reduce`` -> ``pgv_act_bn_bwd`` (also yields the bias gradient) -> ``pgv_conv_wgrad`` and
  the transposed convolution for the input gradient.
"""
import torch
import torch.nn as nn

from .. import ops
from ..ops import PGV_ACT_HARDTANH, PGV_ACT_LEAKY_RELU, PGV_ACT_NONE


def _act_code(activation):
    """Map the reference's activation modules (encoder.py:239-240, decoder.py:98,203-204) to kernel epilogues."""
    if activation is None or isinstance(activation, nn.Identity):
        return PGV_ACT_NONE, 0.0
    if isinstance(activation, nn.LeakyReLU):
        return PGV_ACT_LEAKY_RELU, float(activation.negative_slope)
    if isinstance(activation, nn.ReLU):
        return PGV_ACT_LEAKY_RELU, 0.0
    if isinstance(activation, nn.Hardtanh) and activation.min_val == -1.0 and activation.max_val == 1.0:
        return PGV_ACT_HARDTANH, 0.0
    raise NotImplementedError(f"activation {activation} has no HIP epilogue")


def _one(v):
    if isinstance(v, (list, tuple)):
        if len(set(v)) != 1:
            raise NotImplementedError(f"anisotropic kernel/stride/padding {v} not implemented")
        return int(v[0])
    return int(v)


class _Block:
    """Host-side description of one conv(+act)(+BN) block, bound to its parameter-container modules."""

    def __init__(self, conv, act_module, bn):
        self.conv, self.bn = conv, bn
        self.up = isinstance(conv, nn.ConvTranspose2d)
        self.act, self.slope = _act_code(act_module)
        self.k, self.stride, self.pad = _one(conv.kernel_size), _one(conv.stride), _one(conv.padding)
        if _one(conv.dilation) != 1 or conv.groups != 1 or conv.padding_mode != 'zeros':
            raise NotImplementedError("dilation/groups/non-zero padding modes are not implemented")
        self.out_pad = tuple(conv.output_padding) if self.up else (0, 0)
        if self.up:  # ConvTranspose2d weight [Cin, Cout, kh, kw] = [Cs][Cb][kh][kw]
            self.Cs, self.Cb = conv.in_channels, conv.out_channels
        else:        # Conv2d weight [Cout, Cin, kh, kw] = [Cs][Cb][kh][kw]
            self.Cs, self.Cb = conv.out_channels, conv.in_channels
        self.c_out = conv.out_channels
        self._geoms = {}

    def geom(self, H_in, W_in):
        g = self._geoms.get((H_in, W_in))
        if g is None:
            if self.up:
                Hb = (H_in - 1) * self.stride - 2 * self.pad + self.k + self.out_pad[0]
                Wb = (W_in - 1) * self.stride - 2 * self.pad + self.k + self.out_pad[1]
            else:
                Hb, Wb = H_in, W_in
            g = ops.ConvGeom(self.Cb, self.Cs, self.k, self.stride, self.pad, Hb, Wb)
            if self.up and (g.Hs, g.Ws) != (H_in, W_in):
                raise ValueError("output_padding inconsistent with stride")
            self._geoms[(H_in, W_in)] = g
        return g

    def params(self):
        ps = [self.conv.weight, self.conv.bias]
        if self.bn is not None:
            ps += [self.bn.weight, self.bn.bias]
        return ps


def _grad_dest(param, accumulated=False):
    """Where a parameter gradient is written.  Flat-buffer mode (optim.FlatParams): straight into the parameter's
    slice of the flat gradient buffer and autograd gets ``None``; otherwise a fresh tensor that autograd accumulates
    as usual.  ``accumulated=True`` is for kernels that add into their output with atomics: a third value tells
    whether the destination already holds zeros (PGV_PREZEROED) - true for the flat buffer, which
    ``FusedAdam.zero_grad`` clears with one fill at the start of every step (train.py:208)."""
    view = getattr(param, '_pgv_grad_view', None)
    if getattr(param, '_pgv_shared', False):
        view = None     # parameter of a stack applied several times per forward: autograd sums the applications
    if view is not None:
        if param.grad is not view:
            param.grad = view
        if not accumulated:
            return view, None
        flat = param._pgv_flat
        clean = flat.grad_zeroed and getattr(param, '_pgv_zero_gen', -1) != flat.zero_gen
        param._pgv_zero_gen = flat.zero_gen  # a second backward before the next zero_grad() must clear for itself
        return view, None, clean
    t = torch.empty_like(param)
    return (t, t, False) if accumulated else (t, t)


def _step_zeros(param, n, dtype, tag, device):
    """``n`` zeroed accumulators for one application of a stack: from the optimizer's step scratch (cleared by the fill of
    ``zero_grad()``) when there is one and this is its first use in the step, else a fresh ``torch.zeros``."""
    flat = getattr(param, '_pgv_flat', None)
    if flat is not None and not getattr(param, '_pgv_shared', False):
        t = flat.step_scratch((id(param), tag), n, dtype)
        if t is not None:
            return t
    return torch.zeros(n, device=device, dtype=dtype)


# 'fused': BatchNorm + activation backward of every block that has a consumer inside its stack rides in that consumer's
# input-gradient epilogue (no pass over the gradient); 'passes': the reduce + apply passes for every block (A/B aid).
BN_BACKWARD_MODE = 'fused'
PASSFREE_MIN_PLANE = 1024   # H*W of the block's output from which the pass-free backward is used
BF16_PASSFREE_MIN_N = 1 << 17   # ... and, in bf16 operand mode, B*H*W from which it is used (see ConvStackFn.backward)


def set_bn_backward_mode(mode):
    global BN_BACKWARD_MODE
    if mode not in ('fused', 'passes'):
        raise ValueError(f"unknown BatchNorm backward mode {mode!r}")
    BN_BACKWARD_MODE = mode


_IDENTITY_COEF = {}


def _identity_coef(C, device):
    """(1, 0, 0) coefficients of pgv_bwd_fuse for a block without BatchNorm: one constant tensor per (device, C) instead
    of two fills and a concatenation per step.  Read-only by contract."""
    key = (str(device), C)
    t = _IDENTITY_COEF.get(key)
    if t is None:
        t = _IDENTITY_COEF[key] = torch.cat([torch.ones(C, device=device), torch.zeros(2 * C, device=device)])
    return t


# Set by parallel.GradAllReduce: called with a parameter right after the kernel writing its gradient was launched on
# the current stream (gradient-ready notification for bucketed all-reduce overlap).
GRAD_READY_HOOK = None


def _grad_done(*params):
    """Parameters of a stack that is applied several times per forward (``_pgv_shared``: stacked spectrogram channels)
    are NOT announced: their gradient is the sum over the applications, accumulated by autograd after this kernel, so
    their bucket is only complete when backward has returned - ``GradAllReduce.wait()`` flushes it then."""
    if GRAD_READY_HOOK is not None:
        for p in params:
            if p is not None and not getattr(p, '_pgv_shared', False):
                GRAD_READY_HOOK(p)


# True only while VAETrainStep evaluates the model for a step whose total it assembles itself (total = 1 * recons + beta *
# latent, x_out feeds nothing else): the 'unit' form of the fused reconstruction criterion (ConvStackFn.forward) is taken only
# then - a model used by hand, with any loss built on its outputs, never sees it.
UNIT_RECONS_GRADIENT = False


class ConvStackFn(torch.autograd.Function):
    """A chain of conv blocks evaluated with folded BatchNorm between consecutive blocks."""

    @staticmethod
    def forward(ctx, x, blocks, training, sq_target, sq_scale, out_dropout, *params):
        """``sq_target`` / ``sq_scale`` (optional): also return ``sq_scale * sum((out - sq_target)^2)`` - the
        reconstruction criterion evaluated where the output is produced, so that its backward can be fused with the
        output block's (``pgv_sqerr_act_bwd``).  A NEGATIVE ``sq_scale`` asks for the deferred form (scale = -sq_scale):
        no forward pass over the tensors at all - the returned scalar starts at zero and receives its value from the
        backward kernel, which reads both tensors anyway; only for callers that always run backward before they look
        at the value (VAETrainStep).
        ``out_dropout`` = (rng, p, stream_id) (optional): nn.Dropout on the output (encoder.py:85), in the pass that
        applies the last block's BatchNorm; backward regenerates the mask (``pgv_dropout_fwd`` / ``_bwd``)."""
        x = x.contiguous()
        B = x.shape[0]
        dev = x.device
        # ``sq_scale`` = (scale, 'unit'): the caller PROMISES that the criterion's value enters its total with a gradient of
        # exactly 1 and that the output receives no other gradient (VAETrainStep: total = recons + beta * latent).  The
        # criterion and the output activation's backward then ride in the output layer's forward kernel where it has one
        # (ops.conv_up_sq: 73 + 53 -> 96 us on the 8 -> 1 channel layer), and backward starts from the stored gradient.
        sq_unit = False
        if isinstance(sq_scale, tuple):
            sq_scale, tag = sq_scale
            sq_unit = tag == 'unit'
        ctx.sq_fwd = None
        fwd_sq = (sq_target is not None and sq_unit and sq_scale < 0 and out_dropout is None and blocks[-1].up and
                  blocks[-1].bn is None and blocks[-1].c_out == 1)
        sq_loss_acc = None
        cur, cur_scale, cur_shift = x, None, None
        pending = None
        saved = []
        pi = 0
        # BatchNorm statistics of all blocks in one arena cleared by ONE fill (PGV_PREZEROED): a memset node per block
        # costs ~5 us of dependent-launch latency each
        # (as CLS_COPIES partial copies each: the kernels' end-of-kernel atomics then stay inside their XCD's L2 - 256
        # workgroups finishing together on one copy cost the transposed-conv kernels 7-8 us per launch)
        # (every plane size: the deep-layer kernels spread them too - 17x23: 10 / 18 us per forward launch went into 256
        # same-address float64 atomics per channel; families without copies add into copy 0)
        def stat_copies(blk, h, w):
            go = blk.geom(h, w)
            ho, wo = (go.Hb, go.Wb) if blk.up else (go.Hs, go.Ws)
            return ops.CLS_COPIES, ho, wo
        n_stats, hh, ww = 0, x.shape[2], x.shape[3]
        for blk in blocks:
            sc_b, hh, ww = stat_copies(blk, hh, ww)
            if blk.bn is not None and training:
                n_stats += sc_b * 2 * blk.c_out
        arena = _step_zeros(params[0], n_stats, torch.float64, 'stats', dev) if n_stats else None
        a_off = 0
        # bf16 operand mode: the weights of the layers with bf16-native kernels rounded once per step into the layouts those
        # kernels stream (None otherwise) - one launch for the whole stack; the backward pass multiplies by the same weights
        pairs, hh, ww, pj = [], x.shape[2], x.shape[3], 0
        for blk in blocks:
            go = blk.geom(hh, ww)
            pairs.append((go, params[pj]))
            hh, ww = (go.Hb, go.Wb) if blk.up else (go.Hs, go.Ws)
            pj += 2 + (2 if blk.bn is not None else 0)
        shadows = ops.conv_weight_shadows(pairs)   # (all None in plain fp32 mode, without a launch)
        for bi, blk in enumerate(blocks):
            w, b = params[pi], params[pi + 1]
            pi += 2
            g = blk.geom(cur.shape[2], cur.shape[3])
            has_bn = blk.bn is not None
            gamma = beta = None
            if has_bn:
                gamma, beta = params[pi], params[pi + 1]
                pi += 2
            C = blk.c_out
            stats = None
            if has_bn and training:
                SC = stat_copies(blk, cur.shape[2], cur.shape[3])[0]
                stats = arena[a_off:a_off + SC * 2 * C]
                a_off += SC * 2 * C
            fn = ops.conv_up if blk.up else ops.conv_down
            w_sh = shadows[bi]
            a = None
            if fwd_sq and bi == len(blocks) - 1 and w_sh is None:
                gen0 = getattr(b, '_pgv_zero_gen', -1)
                gb, gb_ret, gb_zero = _grad_dest(b, accumulated=True)
                if not gb_zero:
                    gb.zero_()
                sq_loss_acc = _step_zeros(params[0], 1, torch.float32, 'sqloss', dev)
                sq_cls = _step_zeros(params[0], ops.CLS_COPIES * 4, torch.float32, 'sqcls', dev)
                res = ops.conv_up_sq(g, cur, w, b, blk.act, blk.slope, sq_target.contiguous(), -float(sq_scale), gb,
                                     sq_loss_acc, sq_cls, in_scale=None if pending is not None else cur_scale,
                                     in_shift=None if pending is not None else cur_shift, in_bn=pending)
                if res is not None:
                    a, g_y0 = res
                    ctx.sq_fwd = (g_y0, gb, gb_ret, sq_cls)
                    pending = None
                elif gb_zero:
                    b._pgv_zero_gen = gen0     # (nothing was launched: backward finds the bias gradient's slice as clean as it was)
            # (pending: the producer's train-mode BatchNorm, finalized by this kernel in its prologue - ops.bn_src)
            if a is not None:
                pass
            elif pending is not None:
                a = fn(g, cur, w, b, blk.act, blk.slope, stats=stats, prezeroed=stats is not None, in_bn=pending,
                       stats_copies=stats is not None and SC > 1, w_shadow=w_sh)
                pending = None
            else:
                a = fn(g, cur, w, b, blk.act, blk.slope, in_scale=cur_scale, in_shift=cur_shift, stats=stats,
                       prezeroed=stats is not None, stats_copies=stats is not None and SC > 1, w_shadow=w_sh)
            scale = shift = mean = rstd = None
            if has_bn:
                vec = torch.empty(4 * C, device=dev, dtype=torch.float32)
                scale, shift, mean, rstd = vec[:C], vec[C:2 * C], vec[2 * C:3 * C], vec[3 * C:]
                if training:
                    n = B * a.shape[2] * a.shape[3]
                    if n <= 1:
                        raise ValueError("Expected more than 1 value per channel when training")
                    bn = blk.bn
                    mom = bn.momentum if bn.momentum is not None else 0.1
                    track = bn.track_running_stats and bn.running_mean is not None
                    # not a launch: the finalize arithmetic rides in the prologue of the kernel that applies this
                    # BatchNorm - the next block's, or the output pass below
                    pending = ops.bn_src(stats, n, gamma, beta, bn.eps, mom, bn.running_mean if track else None,
                                         bn.running_var if track else None,
                                         bn.num_batches_tracked if track else None, scale, shift, mean, rstd,
                                         stats_copies=SC)
                else:
                    ops.bn_eval_affine(gamma, beta, blk.bn.running_mean, blk.bn.running_var, blk.bn.eps, scale, shift)
                    mean = rstd = None
            saved.append((cur, cur_scale, cur_shift, a, scale, mean, rstd, g, (w_sh, ops.compute_mode())))
            cur, cur_scale, cur_shift = a, scale, shift
        ctx.drop = None
        if out_dropout is not None:
            rng, p, stream_id = out_dropout
            if pending is not None:
                out, drop_state = rng.dropout_nomask(p, cur, stream_id, in_bn=pending)
            else:
                out, drop_state = rng.dropout_nomask(p, cur, stream_id, cur_scale, cur_shift)
            ctx.drop = (drop_state, stream_id, float(p))
        else:
            if pending is not None:
                ops.bn_src_finalize(pending)
            # (a stack whose last block has no BatchNorm returns that block's activation: as a VIEW, so that the tensor
            # object ctx.saved holds is not the output itself - output -> grad_fn -> ctx -> output would be a reference
            # cycle that keeps the step's activations alive until a garbage collection)
            out = ops.affine_nchw(cur, cur_scale, cur_shift) if cur_scale is not None else cur.view_as(cur)
        ctx.blocks, ctx.saved, ctx.params = blocks, saved, params
        ctx.sq = None
        if sq_target is None:
            return out
        sq_target = sq_target.contiguous()
        ctx.set_materialize_grads(False)
        if sq_scale < 0:
            ctx.sq = (sq_target, -float(sq_scale))
            ctx.sq_deferred = sq_loss_acc if sq_loss_acc is not None else _step_zeros(params[0], 1, torch.float32, 'sqloss', dev)
            return out, ctx.sq_deferred.reshape(())
        ctx.sq = (sq_target, float(sq_scale))
        ctx.sq_deferred = None
        return out, ops.sqerr_fwd(out, sq_target, float(sq_scale))

    @staticmethod
    def backward(ctx, g_out, g_loss=None):
        blocks, saved, params = ctx.blocks, ctx.saved, ctx.params
        fused_sq = False
        sq_fwd = ctx.sq_fwd
        if sq_fwd is not None:
            if g_out is not None:
                raise RuntimeError("ConvStackFn: the stack was run with the 'unit' promise (the criterion is the only consumer of "
                                   "its output) and the output received a gradient of its own")
            if g_loss is None:
                return (None,) * (6 + len(params))
            fused_sq = True               # (the gradient of the output block's pre-activation exists since forward)
        elif ctx.sq is not None:
            last = blocks[-1]
            if g_loss is not None and g_out is None and last.bn is None:
                fused_sq = True          # criterion + output activation backward in one pass, g_out never exists
            elif g_loss is not None:
                g_sq = ops.sqerr_bwd(saved[-1][3], ctx.sq[0], g_loss.contiguous(), ctx.sq[1])
                g_out = g_sq if g_out is None else g_out + g_sq
            if ctx.sq_deferred is not None and not fused_sq:   # the deferred value has no fused kernel to come from
                ctx.sq_deferred.add_(ops.sqerr_fwd(saved[-1][3], ctx.sq[0], ctx.sq[1]))
            if g_out is None and not fused_sq:
                return (None,) * (6 + len(params))
        g_o = g_out.contiguous() if g_out is not None else None
        # (nn.Dropout on the stack's output: its backward pass is the first thing to happen to g_out - as a pass of its
        # own, or, when the top block has a train-mode BatchNorm, inside that block's reduce pass below)
        drop_pending = ctx.drop is not None and g_o is not None
        dev = saved[-1][3].device
        nb = len(blocks)
        B = saved[-1][3].shape[0]
        grads = [None] * len(params)
        # Block li < nb-1 normally gets the pass-free backward: the input-gradient kernel of block li+1 applies block
        # li's BatchNorm + activation backward in its epilogue (pgv_bwd_fuse) with coefficients derived from block
        # li+1's weight gradient (pgv_bn_bwd_coef) - the gradient of block li's BatchNorm output is never stored.
        # Only the top block of a stack (its gradient arrives from outside) runs the reduce + apply passes.
        # (In bf16 operand mode the identity yields sum g*bf16(o) instead of sum g*o: a coherent error of relative size
        # 2^-9/sqrt(n) in every element of g_y, n = elements per channel, that the next weight gradient - a heavily
        # cancelling sum - amplifies; measured 6x the bf16 oracle's self-distance on enc2conv.weight at B = 3, n = 4455.
        # The mode therefore uses the pass-free backward only from n = 2^17 on - training batches - where the term is at
        # the fp32 rounding level.)
        passfree = [False] * nb
        if BN_BACKWARD_MODE == 'fused':
            bf16 = ops.compute_dtype() != 'fp32'
            for li in range(nb - 1):
                if bf16 and saved[li][3].numel() // saved[li][3].shape[1] < BF16_PASSFREE_MIN_N:
                    continue
                up = blocks[li + 1]
                gsz = saved[li + 1][7]
                hy, wy = (gsz.Hb, gsz.Wb) if up.up else (gsz.Hs, gsz.Ws)
                # (only where the consumer's input-gradient kernel has the fused epilogue to gain from - the large planes:
                # on the deep layers' small planes the coefficient / tap-sum launches cost what the reduce pass did)
                passfree[li] = (up.k <= 5 and hy <= 1024 and wy <= 1024 and
                                saved[li][3].shape[2] * saved[li][3].shape[3] >= PASSFREE_MIN_PLANE)
        n_red = sum(2 * blk.c_out for li, (blk, sv) in enumerate(zip(blocks, saved))
                    if blk.bn is not None and sv[5] is not None and not passfree[li])
        arena = _step_zeros(params[0], n_red, torch.float64, 'red', dev) if n_red else None  # BN-backward projections
        def tap_slot(li):   # float64 scratch of the coefficient request of block li (ops.conv_wgrad, coef_req)
            return ops.coef_scratch(saved[li + 1][7], not blocks[li + 1].up)
        n_tap = sum(tap_slot(li) for li in range(nb - 1)
                    if passfree[li] and blocks[li].bn is not None and saved[li][5] is not None)
        tap_arena = _step_zeros(params[0], n_tap, torch.float64, 'tap', dev) if n_tap else None
        a_off = t_off = 0
        pis = []   # index of every block's first parameter
        pi = 0
        for blk in blocks:
            pis.append(pi)
            pi += 4 if blk.bn is not None else 2
        g_y_fused = None   # g_y of the current block when the block above produced it in its input-gradient epilogue
        gb_cur = None      # the current block's bias gradient (complete when g_y is)
        cls_cur = None     # sums of the current block's g_y by (row, column) parity class, when its producer kept them

        def wants_cls(i):
            """Will block i's g_y be the output gradient of a stride-2 ConvTranspose2d whose tap sums are needed?"""
            return (i > 0 and passfree[i - 1] and blocks[i].up and blocks[i].stride == 2 and
                    blocks[i - 1].bn is not None and saved[i - 1][5] is not None)

        n_cls = sum(ops.CLS_COPIES * 4 * blocks[i].c_out for i in range(nb) if wants_cls(i))   # (partial copies per XCD)
        cls_arena = _step_zeros(params[0], n_cls, torch.float32, 'cls', dev) if n_cls else None
        c_off = 0
        # bias gradients that a fused input gradient produces: as per-XCD partial copies too (the same-address atomics of
        # 256 workgroups finishing together cost those kernels 3-6 us each), added up by the reduce launch of the block's
        # own weight gradient (ops.conv_wgrad, bias_finish)
        n_gbc = sum(ops.CLS_COPIES * blocks[i].c_out for i in range(nb - 1) if passfree[i])
        gbc_arena = _step_zeros(params[0], n_gbc, torch.float32, 'gbc', dev) if n_gbc else None
        b_off = 0
        bias_pending = None      # (copies, destination, accumulate) of the current block's bias gradient, if it came as copies
        gb_is_copies = False
        for li in range(nb - 1, -1, -1):
            blk = blocks[li]
            inp, in_scale, in_shift, a, scale, mean, rstd, geom, (w_sh, sh_mode) = saved[li]
            if sh_mode != ops.compute_mode():
                w_sh = None   # (the operand mode changed between forward and backward: the saved shadow has the other
                #               mode's layout - without one the call computes from w itself)
            has_bn = blk.bn is not None
            pi = pis[li]
            w = params[pi]
            C = blk.c_out
            if g_y_fused is not None:
                # bias / BatchNorm parameter gradients were written on the way (gb_cur: this block's bias gradient)
                g_y, g_y_fused = g_y_fused, None
            else:
                red = ggamma = gbeta = None
                fused_bn = False   # BatchNorm-backward reduce + apply as one launch (small planes: ops.bn_act_bwd_fused)
                if drop_pending and not (has_bn and mean is not None):
                    g_o = ops.dropout_bwd(ctx.drop[0], ctx.drop[1], ctx.drop[2], g_o)
                    drop_pending = False
                if has_bn and mean is not None:
                    red = arena[a_off:a_off + 2 * C]
                    a_off += 2 * C
                    if drop_pending:   # Dropout backward + BatchNorm-backward reduce as one pass over the gradient
                        g_o = ops.dropout_bwd_bn_reduce(ctx.drop[0], ctx.drop[1], ctx.drop[2], g_o.reshape(a.shape), a,
                                                        mean, rstd, red, prezeroed=True)
                        drop_pending = False
                    elif not (fused_sq and li == nb - 1) and ops.bn_act_bwd_fusable(a.shape[0], C, a.shape[2] * a.shape[3]):
                        fused_bn = True
                    else:
                        ops.bn_bwd_reduce(g_o, a, mean, rstd, red, prezeroed=True)
                    ggamma, grads[pi + 2] = _grad_dest(params[pi + 2])   # written by act_bn_bwd below
                    gbeta, grads[pi + 3] = _grad_dest(params[pi + 3])
                # (eval-mode BN: gamma/beta gradients are not produced)
                # g_y overwrites g_o unless g_o is the caller's tensor (top block)
                cls_cur = None
                if sq_fwd is not None and li == nb - 1:
                    # (criterion + activation backward happened in the forward kernel: g_y, the bias gradient and the class sums
                    # are there)
                    g_y, gb, gb_ret, cls_f = sq_fwd
                    if wants_cls(li) and C == 1 and a.shape[2] * a.shape[3] >= 16384:
                        cls_cur = cls_f
                        c_off += ops.CLS_COPIES * 4       # (its slice of the arena stays unused)
                else:
                    gb, gb_ret, gb_zero = _grad_dest(params[pi + 1], accumulated=True)
                if sq_fwd is not None and li == nb - 1:
                    pass
                elif fused_sq and li == nb - 1:
                    g_y = torch.empty_like(a)
                    if wants_cls(li) and C == 1 and a.shape[2] * a.shape[3] >= 16384:
                        cls_cur = cls_arena[c_off:c_off + ops.CLS_COPIES * 4]
                        c_off += ops.CLS_COPIES * 4
                    ops.sqerr_act_bwd(a, ctx.sq[0], g_loss.contiguous(), ctx.sq[1], blk.act, blk.slope, g_y, gb,
                                      prezeroed=gb_zero, loss_acc=ctx.sq_deferred, cls=cls_cur)
                else:
                    g_y = g_o if li != nb - 1 else torch.empty_like(g_o)
                    if fused_bn:
                        ops.bn_act_bwd_fused(g_o.reshape(a.shape), a, scale, mean, rstd, blk.act, blk.slope, g_y, gb,
                                             ggamma=ggamma, gbeta=gbeta, prezeroed=gb_zero)
                    else:
                        ops.act_bn_bwd(g_o, a, scale if has_bn else None, mean, rstd, red, blk.act, blk.slope, g_y, gb,
                                       ggamma=ggamma, gbeta=gbeta, prezeroed=gb_zero)
                grads[pi + 1] = gb_ret
                gb_cur = gb
                if ggamma is not None:
                    _grad_done(params[pi + 2], params[pi + 3])
            gw, gw_ret, gw_zero = _grad_dest(w, accumulated=True)
            grads[pi] = gw_ret
            need_dx = li > 0 or ctx.needs_input_grad[0]
            fuse = None
            coef_req = None
            if need_dx and li > 0 and passfree[li - 1]:
                low = blocks[li - 1]
                _, _, _, a_low, _, mean_low, rstd_low, _, _ = saved[li - 1]
                pl = pis[li - 1]
                Cl = low.c_out
                if low.bn is not None and mean_low is not None:
                    # train-mode BatchNorm: coefficients from W * gW of this block and the tap sums of g_y, computed by
                    # this block's weight-gradient call (before the weight gradient is announced: a gradient exchange
                    # rewrites it)
                    T = tap_arena[t_off:t_off + tap_slot(li - 1)]
                    t_off += tap_slot(li - 1)
                    # class sums of g_y: for a Conv2d consumer simply this block's bias gradient; for a ConvTranspose2d
                    # one the sums by row / column parity class
                    cls_copies = 0
                    if not blk.up:
                        cls = gb_cur.reshape(-1)
                        cls_copies = ops.CLS_COPIES if gb_is_copies else 0
                    elif cls_cur is not None:
                        cls = cls_cur
                    else:
                        cls = ops.conv_class_sums(geom, g_y, True)
                    coef = torch.empty(3 * Cl, device=dev, dtype=torch.float32)
                    gg_low, grads[pl + 2] = _grad_dest(params[pl + 2])
                    gbt_low, grads[pl + 3] = _grad_dest(params[pl + 3])
                    coef_req = dict(lower_is_big=not blk.up, cls=cls, w=w, scale=in_scale, shift=in_shift, mean=mean_low,
                                    rstd=rstd_low, n=a_low.numel() // Cl, coef=coef, ggamma=gg_low, gbeta=gbt_low,
                                    scratch=T, cls_copies=cls_copies)
                elif low.bn is not None:   # eval-mode BatchNorm: g_a = scale * g
                    coef = torch.cat([in_scale, torch.zeros(2 * Cl, device=dev, dtype=torch.float32)])
                else:                      # no BatchNorm (the first encoder block): activation backward only
                    coef = _identity_coef(Cl, dev)
                gb_low, grads[pl + 1], gb_zero = _grad_dest(params[pl + 1], accumulated=True)
                gb_copies = gbc_arena[b_off:b_off + ops.CLS_COPIES * Cl]
                b_off += ops.CLS_COPIES * Cl
                bias_next = (gb_copies, gb_low.reshape(-1), False)
                cls_low = None
                if wants_cls(li - 1):
                    cls_low = cls_arena[c_off:c_off + ops.CLS_COPIES * 4 * Cl]
                    c_off += ops.CLS_COPIES * 4 * Cl
                fuse = (a_low, coef, gb_copies, low.act, low.slope, cls_low, ops.CLS_COPIES)
            if blk.up:   # ConvTranspose2d: big = g_y, small = block input (folded BN of the producer)
                ops.conv_wgrad(geom, g_y, inp, gw, small_scale=in_scale, small_shift=in_shift, prezeroed=gw_zero,
                               coef_req=coef_req, bias_finish=bias_pending)
            else:        # Conv2d: big = block input, small = g_y
                ops.conv_wgrad(geom, inp, g_y, gw, big_scale=in_scale, big_shift=in_shift, prezeroed=gw_zero,
                               coef_req=coef_req, bias_finish=bias_pending)
            bias_pending, gb_is_copies = None, False
            # (announced only now: the coefficient arithmetic reads this block's bias and weight gradients, which a
            # gradient exchange rewrites in place)
            _grad_done(params[pi + 1], w)
            if need_dx:
                if blk.up:
                    g_o = ops.conv_down(geom, g_y, w, None, PGV_ACT_NONE, 0.0, bwd_fuse=fuse, w_shadow=w_sh)
                else:
                    g_o = ops.conv_up(geom, g_y, w, None, PGV_ACT_NONE, 0.0, bwd_fuse=fuse, w_shadow=w_sh)
                if fuse is not None:
                    g_y_fused, g_o = g_o, None
                    gb_cur, cls_cur = fuse[2], fuse[5]
                    bias_pending, gb_is_copies = bias_next, True
                    low = blocks[li - 1]
                    pl = pis[li - 1]
                    if low.bn is not None and saved[li - 1][5] is not None:
                        _grad_done(params[pl + 2], params[pl + 3])
            else:
                g_o = None
        return (g_o, None, None, None, None, None) + tuple(grads)


def run_stack(x, blocks, training, sq_target=None, sq_scale=None, out_dropout=None):
    """``sq_target`` given: returns (output, sq_scale * sum((output - sq_target)^2)); ``out_dropout`` = (rng, p,
    stream_id): nn.Dropout on the output - see ConvStackFn.forward."""
    params = []
    for blk in blocks:
        params += blk.params()
    return ConvStackFn.apply(x, tuple(blocks), bool(training), sq_target, sq_scale, out_dropout, *params)


class _ConvBlockBase(nn.Sequential):
    """Shared body of Conv2D / TConv2D: children keep the reference names, forward goes through the HIP stack."""

    def _finish(self, conv_name, conv, activation, name_prefix, batch_norm):
        if batch_norm == 'before':
            raise NotImplementedError("batch_norm='before' is unused by speccnn8l1_bn and not implemented")
        self.add_module(name_prefix + conv_name, conv)
        self.add_module(name_prefix + 'act', activation)
        bn = None
        if batch_norm == 'after':
            bn = nn.BatchNorm2d(conv.out_channels)
            self.add_module(name_prefix + 'bn', bn)
        elif batch_norm is not None:
            raise ValueError(f"batch_norm={batch_norm!r}")
        self._pgv_block = _Block(conv, activation, bn)

    def pgv_blocks(self):
        return [self._pgv_block]

    def forward(self, x):
        return run_stack(x, [self._pgv_block], self.training)


class Conv2D(_ConvBlockBase):
    """conv -> activation -> BatchNorm2d (reference model/layer.py:10-26)."""

    def __init__(self, in_ch, out_ch, kernel_size, stride, padding, dilation,
                 padding_mode='zeros', activation=None, name_prefix='', batch_norm='after'):
        super().__init__()
        activation = nn.ReLU() if activation is None else activation
        conv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, padding, dilation, padding_mode=padding_mode)
        self._finish('conv', conv, activation, name_prefix, batch_norm)


class TConv2D(_ConvBlockBase):
    """transposed conv -> activation -> BatchNorm2d (reference model/layer.py:29-46)."""

    def __init__(self, in_ch, out_ch, kernel_size, stride, padding, output_padding=0, dilation=1,
                 padding_mode='zeros', activation=None, name_prefix='', batch_norm='after'):
        super().__init__()
        activation = nn.ReLU() if activation is None else activation
        conv = nn.ConvTranspose2d(in_ch, out_ch, kernel_size, stride, padding, output_padding,
                                  dilation=dilation, padding_mode=padding_mode)
        self._finish('tconv', conv, activation, name_prefix, batch_norm)


def _small_zeros(param, shape, tag):
    """A zeroed [shape] activation buffer from the optimizer's step scratch (cleared by zero_grad's one fill) for a
    split-K product to accumulate into, or None: small tensors only, first use in the step only."""
    n = 1
    for v in shape:
        n *= int(v)
    flat = getattr(param, '_pgv_flat', None)
    if flat is None or n > 1 << 18 or getattr(param, '_pgv_shared', False):
        return None
    t = flat.step_scratch((id(param), tag), n, torch.float32)
    # (.data: same memory, but not a view of the scratch buffer as far as autograd is concerned - in-place torch ops on
    # other slices of the scratch would otherwise invalidate every tensor saved from this one)
    return None if t is None else t.view(*shape).data


class LinearFn(torch.autograd.Function):
    """nn.Linear on the f32-MFMA GEMM (encoder.py:85, decoder.py:64).  The long-K products (encoder forward, decoder
    input gradient: K = 25 024) are split over K and accumulate with atomics: into a zeroed slice of the step scratch when
    there is one (no clearing launch), as the weight gradient does into the zero_grad'ed flat gradient."""

    @staticmethod
    def forward(ctx, x, w, b, out_dropout=None, bias_grad_elsewhere=False):
        """``out_dropout`` = (rng, p, stream_id): nn.Dropout on the output (decoder.py:64-65) as part of this function -
        its backward pass then also sums the columns of the gradient for the bias.  ``bias_grad_elsewhere``: the
        consumer's backward kernel writes the bias gradient (EncoderHeadFn)."""
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.params = (w, b)
        ctx.bias_elsewhere = bool(bias_grad_elsewhere)
        ctx.drop = None
        # (the output is handed to autograd consumers that keep it only until the end of the step)
        out = _small_zeros(w, (x.shape[0], w.shape[0]), 'lin_y') if x.shape[1] >= 4096 else None
        y = ops.linear_fwd(x, w, b, out=out)
        if out_dropout is not None:
            rng, p, stream_id = out_dropout
            y, saved = rng.dropout_nomask(p, y, stream_id)
            ctx.drop = (saved, stream_id, float(p))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        wp, bp = ctx.params
        gy = gy.contiguous()
        bias_done, gb_ret, gb_zero = ctx.bias_elsewhere, None, False
        if ctx.drop is not None:
            gb = None
            if bp is not None and not bias_done:
                gb, gb_ret, gb_zero = _grad_dest(bp, accumulated=True)
                bias_done = True
            gy = ops.dropout_bwd(ctx.drop[0], ctx.drop[1], ctx.drop[2], gy, colsum=gb, prezeroed=bias_done and gb_zero)
        gx = None
        if ctx.needs_input_grad[0]:
            out = _small_zeros(wp, (gy.shape[0], w.shape[1]), 'lin_gx') if gy.shape[1] >= 4096 else None
            gx = ops.linear_dgrad(gy, w, out=out)
        gw, gw_ret, gw_zero = _grad_dest(wp, accumulated=True)
        ops.linear_wgrad(gy, x, gw, prezeroed=gw_zero)
        if not bias_done:
            if bp is not None:
                gb, gb_ret, gb_zero = _grad_dest(bp, accumulated=True)
                ops.colsum(gy, gb, prezeroed=gb_zero)
        _grad_done(wp, bp)
        return gx, gw_ret, gb_ret, None, None


class EncoderHeadFn(torch.autograd.Function):
    """The encoder's output BatchNorm1d (train mode, encoder.py:86-87) + reparameterisation + Dkl (VAE.py:49-56,
    loss.py:57-66) as one launch per direction (``pgv_bn1d_reparam_fwd`` / ``_bwd``); the backward launch also writes
    the bias gradient of the Linear in front (``lin_bias``: that layer is called with ``bias_grad_elsewhere``).
    Returns (z_mu_logvar [B, 2D], z [B, D], Dkl)."""

    @staticmethod
    def forward(ctx, x, bn, gamma, beta, rng, kl_scale, kl_buf, lin_bias):
        x = x.contiguous()
        B = x.shape[0]
        if B <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        track = bn.track_running_stats and bn.running_mean is not None
        mom = bn.momentum if bn.momentum is not None else 0.1
        if track and (bn.num_batches_tracked.dtype != torch.int64 or not bn.num_batches_tracked.is_cuda):
            raise ValueError("num_batches_tracked must be an int64 device tensor")
        y, scale, mean, rstd, z, kl, eps = rng.bn1d_reparam(
            x, gamma, beta, bn.eps, mom, bn.running_mean if track else None, bn.running_var if track else None,
            bn.num_batches_tracked if track else None, kl_scale, kl_buf)
        # (save_for_backward, not an attribute: y is an OUTPUT of this function - kept as an attribute it would close a
        # reference cycle output -> grad_fn -> ctx -> output and hold the whole step's activations until a garbage
        # collection: eager steps grew by 780 MB each)
        ctx.save_for_backward(x, y, eps, scale, mean, rstd)
        ctx.params = (gamma, beta, lin_bias)
        ctx.kl_scale = kl_scale
        ctx.set_materialize_grads(False)
        return y, z, kl

    @staticmethod
    def backward(ctx, g_y, g_z, g_kl):
        x, y, eps, scale, mean, rstd = ctx.saved_tensors
        gamma, beta, lin_bias = ctx.params
        c = lambda t: None if t is None else t.contiguous()   # noqa: E731
        gx = torch.empty_like(x)
        ggamma, gg_ret = _grad_dest(gamma)
        gbeta, gbt_ret = _grad_dest(beta)
        gb = gb_ret = None
        gb_zero = False
        if lin_bias is not None:
            gb, gb_ret, gb_zero = _grad_dest(lin_bias, accumulated=True)
        ops.bn1d_reparam_bwd(c(g_z), c(g_kl), c(g_y), y, eps, x, scale, mean, rstd, ctx.kl_scale, gx, ggamma, gbeta,
                             colsum=gb, colsum_accumulate=False)
        _grad_done(gamma, beta)
        return gx, None, gg_ret, gbt_ret, None, None, None, gb_ret


class MaskMulFn(torch.autograd.Function):
    """y = x * mask with a pre-scaled keep mask (nn.Dropout train mode: encoder.py:85, decoder.py:65)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return ops.mul(x.contiguous(), mask)

    @staticmethod
    def backward(ctx, gy):
        (mask,) = ctx.saved_tensors
        return ops.mul(gy.contiguous(), mask), None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout (train mode) with the mask drawn from the on-device Philox stream inside the forward kernel."""

    @staticmethod
    def forward(ctx, x, rng, p, stream_id=0):
        # no stored mask: backward regenerates it from the generator state of this draw (two words)
        y, saved = rng.dropout_nomask(p, x.contiguous(), stream_id)
        ctx.drop = (saved, stream_id, float(p))
        return y

    @staticmethod
    def backward(ctx, gy):
        saved, stream_id, p = ctx.drop
        return ops.dropout_bwd(saved, stream_id, p, gy.contiguous()), None, None, None


class BatchNorm1dFn(torch.autograd.Function):
    """nn.BatchNorm1d over [B, C] (encoder.py:86-87).  Train mode: one launch per direction (``pgv_bn1d_fwd`` /
    ``pgv_bn1d_bwd``); eval mode: the folded running-statistics affine."""

    @staticmethod
    def forward(ctx, x, bn, training, gamma, beta):
        x = x.contiguous()
        B, C = x.shape
        dev = x.device
        vec = torch.empty(3 * C, device=dev, dtype=torch.float32)
        scale, mean, rstd = vec[:C], vec[C:2 * C], vec[2 * C:]
        ctx.params = (gamma, beta)
        if training:
            if B <= 1:
                raise ValueError("Expected more than 1 value per channel when training")
            track = bn.track_running_stats and bn.running_mean is not None
            mom = bn.momentum if bn.momentum is not None else 0.1
            y = torch.empty_like(x)
            if track and (bn.num_batches_tracked.dtype != torch.int64 or not bn.num_batches_tracked.is_cuda):
                raise ValueError("num_batches_tracked must be an int64 device tensor")
            ops.bn1d_fwd(x, gamma, beta, bn.eps, mom, bn.running_mean if track else None,
                         bn.running_var if track else None, bn.num_batches_tracked if track else None, y, scale, mean,
                         rstd)
            ctx.saved = (x, scale, mean, rstd)
            return y
        shift = torch.empty(C, device=dev, dtype=torch.float32)
        ops.bn_eval_affine(gamma, beta, bn.running_mean, bn.running_var, bn.eps, scale, shift)
        ctx.saved = (x, scale, None, None)
        return ops.affine_nchw(x.view(B, C, 1), scale, shift).view(B, C)

    @staticmethod
    def backward(ctx, g):
        x, scale, mean, rstd = ctx.saved
        gamma, beta = ctx.params
        B, C = x.shape
        g = g.contiguous()
        gx = torch.empty_like(x)
        if mean is not None:
            ggamma, gg_ret = _grad_dest(gamma)
            gbeta, gb_ret = _grad_dest(beta)
            ops.bn1d_bwd(g, x, scale, mean, rstd, gx, ggamma, gbeta)
            _grad_done(gamma, beta)
            return gx, None, None, gg_ret, gb_ret
        # eval mode: g_x = scale * g
        x3, g3 = x.view(B, C, 1), g.view(B, C, 1)
        ops.act_bn_bwd(g3, x3, scale, None, None, None, PGV_ACT_NONE, 0.0, gx.view(B, C, 1), None)
        return gx, None, None, None, None
