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


def _vo():
    from oracle import vae_oracle
    return vae_oracle


def param_shapes(arch, dim_z, output_bn):
    """Reference state-dict template (key order as the reference registers them), from the layer tables."""
    enc_rows, dec_rows, cnn_in = _vo().arch_tables(arch)
    tpl = {}

    def bn(prefix, c):
        tpl[prefix + '.weight'] = (c,)
        tpl[prefix + '.bias'] = (c,)
        tpl[prefix + '.running_mean'] = (c,)
        tpl[prefix + '.running_var'] = (c,)
        tpl[prefix + '.num_batches_tracked'] = ()

    for i, (name, ci, co, k, s, p, has_bn) in enumerate(enc_rows):
        if arch == 'speccnn8l1_bn' and name == 'enc7':
            base = 'encoder.features_mixer_cnn.0.'
        elif arch == 'speccnn8l1_bn' and name == 'enc8':
            base = 'encoder.features_mixer_cnn.1.'
        else:
            base = f'encoder.single_ch_cnn.enc_nn.{i}.'
        tpl[base + name + 'conv.weight'] = (co, ci, k, k)
        tpl[base + name + 'conv.bias'] = (co,)
        if has_bn:
            bn(base + name + 'bn', co)
    feat = enc_rows[-1][2] * (3 * 4 if arch == 'speccnn8l1_bn' else 17 * 23)
    tpl['encoder.mlp.1.weight'] = (2 * dim_z, feat)
    tpl['encoder.mlp.1.bias'] = (2 * dim_z,)
    if output_bn:
        bn('encoder.mlp.lat_in_regularization', 2 * dim_z)
    tpl['decoder.mlp.0.weight'] = (int(np.prod(cnn_in)), dim_z)
    tpl['decoder.mlp.0.bias'] = (int(np.prod(cnn_in)),)
    j = 0
    for (name, ci, co, k, s, p, op, has_bn) in dec_rows:
        if name == 'dec1':
            base = 'decoder.features_unmixer_cnn.'
        else:
            base = f'decoder.single_ch_cnn.dec_nn.{j}.'
            j += 1
        tpl[base + name + 'tconv.weight'] = (ci, co, k, k)
        tpl[base + name + 'tconv.bias'] = (co,)
        if has_bn:
            bn(base + name + 'bn', co)
    tpl[f'decoder.single_ch_cnn.dec_nn.{j}.weight'] = (8, 1, 5, 5)
    tpl[f'decoder.single_ch_cnn.dec_nn.{j}.bias'] = (1,)
    return tpl




def stack_channels(x1, n_ch):
    """Extra spectrogram channels of the stacked-input cases (same formula as tests/golden/make_goldens.py)."""
    return torch.cat([x1] + [torch.roll(x1, shifts=37 * c, dims=3) * (1.0 - 0.2 * c) for c in range(1, n_ch)], dim=1)


def template_from_meta(g):
    """State-dict template (reference registration order) stored in a golden file as meta/keys + meta/shapes."""
    tpl = {}
    for k, sh in zip(g['meta/keys'], g['meta/shapes']):
        tpl[str(k)] = tuple(int(v) for v in str(sh).split()) if str(sh) else ()
    return tpl


class MiniPresetIndexesHelper:
    """Duck-typed PresetIndexesHelper (data/preset.py) of a 14-column learnable representation used by the f4 goldens:
    numerical columns 0-3 and 13 (13 = a categorical VST parameter learned as numerical), one-hot groups 4-6, 7-10
    (a numerical VST parameter learned as categorical) and 11-12; columns 0 and 1 act as 'operator volumes' whose
    zero value makes other parameters useless (same mechanism as the Dexed rule of data/preset.py:259-281)."""
    learnable_preset_size = 14
    vst_param_cardinals = {0: -1, 1: 5, 2: 3, 3: -1, 4: 3, 5: 4, 6: 2, 7: 4}
    num_idx_learned_as_num = {0: 0, 1: 1, 2: 2, 3: 3}
    num_idx_learned_as_cat = {5: [7, 8, 9, 10]}
    cat_idx_learned_as_cat = {4: [4, 5, 6], 6: [11, 12]}
    cat_idx_learned_as_num = {7: 13}
    useless_rules = [(0, [2, 3], [7]), (1, [], [11])]

    def get_numerical_learnable_indexes(self):
        return [0, 1, 2, 3, 13]

    def get_categorical_learnable_indexes(self):
        return [[4, 5, 6], [7, 8, 9, 10], [11, 12]]

    def get_useless_learned_params_indexes(self, preset_GT):
        nums, cats = [], []
        for trig, n, c in self.useless_rules:
            if preset_GT[trig].item() < 1e-3:
                nums += n
                cats += c
        return nums, cats


class RandomPresetIndexesHelper(MiniPresetIndexesHelper):
    """A seeded Dexed-sized learnable representation (numerical columns, one-hot groups of 2..32 classes, six 'operator
    volume' rules that make numerical columns and groups useless, a few columns no term reads) for the f4 kernels."""

    def __init__(self, seed=0, n_num=60, n_groups=40, n_rules=6, n_unused=5):
        import random
        rnd = random.Random(seed)
        sizes = [rnd.choice([2, 3, 4, 5, 8, 12, 32]) for _ in range(n_groups)]
        cols = list(range(n_num + sum(sizes) + n_unused))
        rnd.shuffle(cols)
        self._num = sorted(cols[:n_num])
        pos, self._groups = n_num, []
        for s in sizes:
            self._groups.append(sorted(cols[pos:pos + s]))
            pos += s
        self.learnable_preset_size = len(cols)
        trig = self._num[:n_rules]
        others = self._num[n_rules:]
        self.useless_rules = []
        for r in range(n_rules):
            nums = rnd.sample(others, 6)
            cats = [g[0] for g in rnd.sample(self._groups, 5)]
            self.useless_rules.append((trig[r], nums, cats))
        n_as_num = n_num - 8
        self.num_idx_learned_as_num = {v: c for v, c in enumerate(self._num[:n_as_num])}
        self.cat_idx_learned_as_num = {1000 + i: c for i, c in enumerate(self._num[n_as_num:])}
        half = n_groups // 2
        self.num_idx_learned_as_cat = {2000 + i: g for i, g in enumerate(self._groups[:half])}
        self.cat_idx_learned_as_cat = {3000 + i: g for i, g in enumerate(self._groups[half:])}
        self.vst_param_cardinals = {v: rnd.choice([-1, 2, 5, 17, 100]) for v in self.num_idx_learned_as_num}
        self.vst_param_cardinals.update({v: rnd.choice([2, 3, 7]) for v in self.cat_idx_learned_as_num})
        self.vst_param_cardinals.update({v: len(g) for v, g in self.num_idx_learned_as_cat.items()})
        self.vst_param_cardinals.update({v: len(g) for v, g in self.cat_idx_learned_as_cat.items()})

    def get_numerical_learnable_indexes(self):
        return list(self._num)

    def get_categorical_learnable_indexes(self):
        return [list(g) for g in self._groups]

    def random_batch(self, B, seed=1):
        """(u_in one-hot targets with some zero 'volumes', u_out probabilities in (0, 1)) as float64 CPU tensors."""
        import torch
        gen = torch.Generator().manual_seed(seed)
        L = self.learnable_preset_size
        u_in = torch.rand((B, L), generator=gen, dtype=torch.float64)
        u_out = torch.rand((B, L), generator=gen, dtype=torch.float64) * 0.9 + 0.05
        for g in self._groups:
            cls = torch.randint(0, len(g), (B,), generator=gen)
            u_in[:, g] = 0.0
            u_in[torch.arange(B), torch.tensor(g)[cls]] = 1.0
        for trig, _, _ in self.useless_rules:
            off = torch.rand((B,), generator=gen) < 0.3
            u_in[off, trig] = 0.0
        return u_in, u_out
