"""Shared test helpers: golden loading, deterministic inputs identical to tests/golden/make_goldens.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth_input(B, H=257, W=347, dtype=torch.float64):
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    h = torch.arange(H, dtype=torch.float64).view(1, 1, H, 1)
    w = torch.arange(W, dtype=torch.float64).view(1, 1, 1, W)
    x = torch.sin(0.05 * h * (1 + 0.3 * b) + 0.5 * b) * torch.cos(0.021 * w + 0.1 * b) * torch.exp(-w / 300.0)
    x = x + 0.3 * torch.sin(0.37 * h + 0.11 * w * (1 + b))
    return torch.clamp(x * 1.2 - 0.2, -1.0, 1.0).to(dtype)


def synth_vec(shape, a, ph, dtype=torch.float64):
    n = int(np.prod(shape))
    return torch.sin(torch.arange(n, dtype=torch.float64) * a + ph).reshape(shape).to(dtype)


def unpack_mask(g, which, p=0.3, dtype=torch.float64):
    shape = tuple(int(v) for v in g[f'in/{which}_mask_shape'])
    bits = np.unpackbits(g[f'in/{which}_mask_bits'])[:int(np.prod(shape))]
    return torch.tensor(bits.reshape(shape), dtype=dtype) / (1.0 - p)


def golden_state_dict(g, template_keys_shapes, dtype=torch.float64):
    from oracle import vae_oracle as vo
    return vo.closed_form_state_dict(template_keys_shapes, seed=int(g['meta/seed']), dtype=dtype)


def check_big(name, t, g, prefix, rtol, atol=1e-12):
    """Compare a big tensor with its golden checksum + strided sample."""
    t = t.detach().double().reshape(-1).cpu()
    cs = g[prefix + '/checksum']
    idx = torch.tensor(g[prefix + '/sample_idx'])
    sample = torch.tensor(g[prefix + '/sample'])
    scale = max(float(cs[2]), 1e-30)
    got = t[idx]
    err = (got - sample).abs().max().item()
    assert err <= rtol * scale + atol, f"{name}: sample max err {err:.3e} vs scale {scale:.3e}"
    assert abs(t.abs().sum().item() - cs[1]) <= rtol * max(cs[1], 1e-30) * 10 + atol * t.numel(), \
        f"{name}: abs-sum {t.abs().sum().item():.9e} vs {cs[1]:.9e}"
    assert abs(t.abs().max().item() - cs[2]) <= rtol * scale * 10 + atol, f"{name}: max-abs mismatch"


def rel_l2(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
