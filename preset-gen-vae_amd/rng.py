"""On-device counter-based RNG (Philox4x32-10 in ``csrc/eltwise.hip``) for the Dropout masks and the
reparameterisation noise (reference: ``nn.Dropout`` encoder.py:85 / decoder.py:65, ``Normal(...).sample()`` VAE.py:54-55).

The reference draws from torch's device generator; its bit stream cannot be reproduced, only its distribution
(SURVEY.md §3.2), so parity tests inject masks / eps and the product path uses this stream.  The state lives in device
memory ({seed, offset} as two 64-bit words) and is advanced by a kernel, so a captured hipGraph keeps drawing fresh
numbers on every replay."""
import torch

from . import ops


class DeviceRNG:
    def __init__(self, device, seed=None):
        if seed is None:
            seed = torch.initial_seed()
        self.state = torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        self._calls = 0

    def _advance(self, n):
        ops.rng_advance(self.state, (n + 3) // 4)

    def dropout_mask(self, p, shape):
        n = 1
        for s in shape:
            n *= int(s)
        mask = ops.dropout_mask(self.state, 0, float(p), n, self.state.device)
        self._advance(n)
        return mask.view(*shape)

    def dropout(self, p, x):
        """nn.Dropout forward on ``x`` in one pass: returns (x * mask, mask); the same draw as ``dropout_mask``."""
        y, mask = ops.dropout_apply(self.state, 0, float(p), x)
        self._advance(x.numel())
        return y, mask

    def normal(self, shape):
        out = ops.normal(self.state, 1, tuple(int(s) for s in shape), self.state.device)
        self._advance(out.numel())
        return out


def device_rng(module, device):
    """RNG attached to ``module`` (created lazily on ``device``; seeded from ``torch.initial_seed()``)."""
    rng = getattr(module, '_pgv_rng_obj', None)
    if rng is None or rng.state.device != device:
        rng = DeviceRNG(device)
        object.__setattr__(module, '_pgv_rng_obj', rng)
    return rng
