"""On-device counter-based RNG (Philox4x32-10 in ``csrc/eltwise.hip``) for the Dropout masks and the
reparameterisation noise (reference: ``nn.Dropout`` encoder.py:85 / decoder.py:65, ``Normal(...).sample()`` VAE.py:54-55).

The reference draws from torch's device generator; its bit stream cannot be reproduced, only its distribution
(SURVEY.md §3.2), so parity tests inject masks / eps and the product path uses this stream.  The state lives in device
memory ({seed, offset} as two 64-bit words) and is advanced by a kernel, so a captured hipGraph keeps drawing fresh
numbers on every replay."""
import torch

from . import ops


# Philox stream ids of the consumers: one counter space each, so draws at the same offset are independent
STREAM_DROPOUT, STREAM_EPS, STREAM_ENC_DROPOUT, STREAM_DEC_DROPOUT = 0, 1, 2, 3


class DeviceRNG:
    """{seed, offset} in device memory.  Every consumer draws on its own Philox stream id; the offset advances after a
    draw - immediately, or, between ``begin()`` and ``flush()``, ONCE for all draws of a forward pass (one tiny
    dependent launch per step instead of one per draw)."""

    def __init__(self, device, seed=None):
        if seed is None:
            seed = torch.initial_seed()
        self.state = torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        self._deferred = False
        self._pending = 0
        # hand_over: flush() does not launch the advance but leaves the increment in ``owed`` for a caller that folds it
        # into a launch of its own (VAETrainStep: the optimizer's step-counter launch) and collects it with take_owed()
        self.hand_over = False
        self.owed = 0

    def take_owed(self):
        inc, self.owed = self.owed, 0
        return inc

    def begin(self):
        self._deferred, self._pending = True, 0

    def flush(self):
        if self._pending:
            if self.hand_over:
                self.owed += self._pending
            else:
                ops.rng_advance(self.state, self._pending)
        self._deferred, self._pending = False, 0

    def _advance(self, n):
        inc = (n + 3) // 4
        if self._deferred:
            self._pending = max(self._pending, inc)
        else:
            ops.rng_advance(self.state, inc)

    def dropout_mask(self, p, shape, stream_id=STREAM_DROPOUT):
        n = 1
        for s in shape:
            n *= int(s)
        mask = ops.dropout_mask(self.state, stream_id, float(p), n, self.state.device)
        self._advance(n)
        return mask.view(*shape)

    def dropout(self, p, x, stream_id=STREAM_DROPOUT):
        """nn.Dropout forward on ``x`` in one pass: returns (x * mask, mask); the same draw as ``dropout_mask``."""
        y, mask = ops.dropout_apply(self.state, stream_id, float(p), x)
        self._advance(x.numel())
        return y, mask

    def dropout_nomask(self, p, x, stream_id=STREAM_DROPOUT, scale=None, shift=None, in_bn=None):
        """nn.Dropout forward on ``x`` (after an optional per-channel affine, given as vectors or as the BatchNorm to
        finalize - ``ops.bn_src``) without a stored mask: returns (y, saved_state) for ``ops.dropout_bwd``; the same
        draw as ``dropout_mask``."""
        y, saved = ops.dropout_fwd(self.state, stream_id, float(p), x, scale, shift, in_bn)
        self._advance(x.numel())
        return y, saved

    def reparam_kl(self, ml, kl_scale, kl=None, stream_id=STREAM_EPS):
        """z = mu + sigma * eps with eps drawn in the same launch (the draw ``normal((B, D))`` makes), plus the Dkl term:
        returns (z, kl, eps)."""
        z, kl, eps = ops.reparam_kl_fwd_rng(ml, self.state, stream_id, kl_scale, kl)
        self._advance(eps.numel())
        return z, kl, eps

    def bn1d_reparam(self, x, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, kl_scale, kl=None,
                     stream_id=STREAM_EPS):
        """Train-mode BatchNorm1d over x[B, 2D], then ``reparam_kl`` on the result, as one launch
        (``ops.bn1d_reparam_fwd``): returns (y, scale, mean, rstd, z, kl, eps)."""
        out = ops.bn1d_reparam_fwd(x, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                                   self.state, stream_id, kl_scale, kl)
        self._advance(out[-1].numel())
        return out

    def normal(self, shape, stream_id=STREAM_EPS):
        out = ops.normal(self.state, stream_id, tuple(int(s) for s in shape), self.state.device)
        self._advance(out.numel())
        return out


def device_rng(module, device):
    """RNG attached to ``module`` (created lazily on ``device``; seeded from ``torch.initial_seed()``).  A module whose
    ``_rng`` attribute is set (encoder / decoder inside a VAE) draws from that shared generator instead."""
    shared = getattr(module, '_rng', None)
    if shared is not None and shared.state.device == device:
        return shared
    rng = getattr(module, '_pgv_rng_obj', None)
    if rng is None or rng.state.device != device:
        rng = DeviceRNG(device)
        object.__setattr__(module, '_pgv_rng_obj', rng)
    return rng
