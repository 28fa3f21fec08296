"""Preset-regression losses and metrics evaluated ON THE DEVICE without host synchronisation.

Mirrors of the reference's ``model/loss.py``:

* ``SynthParamsLoss`` (:72-183): MSE / L2 over the numerical learnable columns plus a categorical term per one-hot
  group (categorical cross-entropy, optionally with a temperature softmax, or binary cross-entropy / 8), useless
  parameters (e.g. a Dexed operator with zero output level) excluded per row;
* ``QuantizedNumericalParamsLoss`` (:187-261): numerical VST parameters after the synth's quantisation;
* ``CategoricalParamsAccuracy`` (:265-315).

The reference walks rows, parameters and groups in Python and calls ``.item()`` per group; here every group is a
padded row of one index matrix, so a call is a handful of gathers / masked reductions and returns device scalars
(``CategoricalParamsAccuracy`` with ``reduce=False`` is the only path that reads values back, once).  Unlike the
reference (loss.py:134-135) the inputs are not modified in place.

Useless-parameter rules: ``idx_helper.useless_rules`` = ``[(trigger_learn_idx, [num_learn_idx...],
[cat_first_learn_idx...]), ...]`` (a row's listed parameters are useless when ``u_in[row, trigger] < 1e-3``) if the
helper provides it, else the Dexed rule built from ``full_to_learnable`` exactly as ``data/preset.py:259-281`` does,
else none."""
import numpy as np
import torch
import torch.nn.functional as F

from . import loss as _loss


def _dexed_useless_rules(idx_helper):
    """data/preset.py:259-281: operator i (output-level VST index 31 + 22 i) at zero volume makes its own parameters
    (VST indexes 23-30, 32-43 + 22 i) useless."""
    f2l = idx_helper.full_to_learnable
    base = [23, 24, 25, 26, 27, 28, 29, 30, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43]
    rules = []
    for op_i, vol_idx in enumerate([31 + 22 * i for i in range(6)]):
        trig = f2l[vol_idx]
        if trig is None:
            continue
        if not isinstance(trig, int):
            raise NotImplementedError("Dexed Operator output volume learned as categorical")
        nums, cats = [], []
        for vst_idx in [idx + op_i * 22 for idx in base]:
            learn = f2l[vst_idx]
            if isinstance(learn, int):
                nums.append(learn)
            elif isinstance(learn, list):
                cats.append(learn[0])
        rules.append((trig, nums, cats))
    return rules


def _useless_rules(idx_helper):
    rules = getattr(idx_helper, 'useless_rules', None)
    if rules is not None:
        return list(rules)
    synth = getattr(idx_helper, '_synth', None)
    if synth is not None and getattr(synth, 'name', str(synth)) == 'DEXED':
        return _dexed_useless_rules(idx_helper)
    return []


class SynthParamsLoss:
    def __init__(self, idx_helper, normalize_losses, categorical_loss_factor=0.2, prevent_useless_params_loss=True,
                 cat_bce=True, cat_softmax=False, cat_softmax_t=0.1):
        if cat_bce and cat_softmax:
            raise ValueError("'cat_bce' (Binary Cross-Entropy) and 'cat_softmax' (implies Categorical Cross-Entropy)"
                             "cannot be both set to True")
        self.idx_helper = idx_helper
        self.normalize_losses = normalize_losses
        self.cat_bce, self.cat_softmax, self.cat_softmax_t = cat_bce, cat_softmax, cat_softmax_t
        self.cat_loss_factor = categorical_loss_factor
        self.prevent_useless_params_loss = prevent_useless_params_loss
        self.numerical_criterion = _loss.MSELoss(reduction='mean') if normalize_losses else _loss.L2Loss()
        self.num_indexes = list(idx_helper.get_numerical_learnable_indexes())
        self.cat_indexes = [list(g) for g in idx_helper.get_categorical_learnable_indexes()]
        self._rules = _useless_rules(idx_helper) if prevent_useless_params_loss else []
        self._tables = {}

    def _dev_tables(self, device, L):
        key = (str(device), L)
        t = self._tables.get(key)
        if t is not None:
            return t
        G = len(self.cat_indexes)
        K = max((len(g) for g in self.cat_indexes), default=1)
        idx = torch.zeros((G, K), dtype=torch.long)
        valid = torch.zeros((G, K), dtype=torch.bool)
        for gi, g in enumerate(self.cat_indexes):
            idx[gi, :len(g)] = torch.tensor(g, dtype=torch.long)
            valid[gi, :len(g)] = True
        first_to_group = {g[0]: gi for gi, g in enumerate(self.cat_indexes)}
        R = len(self._rules)
        trig = torch.tensor([r[0] for r in self._rules], dtype=torch.long)
        num_member = torch.zeros((R, L), dtype=torch.float32)
        cat_member = torch.zeros((R, max(G, 1)), dtype=torch.float32)
        for ri, (_, nums, cats) in enumerate(self._rules):
            for n in nums:
                num_member[ri, n] = 1.0
            for c in cats:
                if c in first_to_group:
                    cat_member[ri, first_to_group[c]] = 1.0
        t = {'num_idx': torch.tensor(self.num_indexes, dtype=torch.long, device=device),
             'cat_idx': idx.to(device), 'cat_valid': valid.to(device), 'trig': trig.to(device),
             'num_member': num_member.to(device), 'cat_member': cat_member.to(device)}
        self._tables[key] = t
        return t

    def __call__(self, u_out, u_in):
        """Categorical parameters must be one-hot encoded.  Returns a 0-d tensor on the inputs' device."""
        B, L = u_in.shape
        t = self._dev_tables(u_in.device, L)
        useless_num = useless_cat = None
        if self._rules:
            off = (u_in[:, t['trig']] < 1e-3).to(u_in.dtype)                 # [B, R]
            useless_num = (off @ t['num_member']) > 0                          # [B, L]
            useless_cat = (off @ t['cat_member']) > 0                          # [B, G]
        num_loss = 0.0
        if len(self.num_indexes) > 0:
            a, b = u_out[:, t['num_idx']], u_in[:, t['num_idx']]
            if useless_num is not None:                                        # loss.py:128-135 zeroes both sides
                keep = ~useless_num[:, t['num_idx']]
                a, b = a * keep, b * keep
            num_loss = self.numerical_criterion(a.contiguous(), b.contiguous())
        cat_loss = 0.0
        G = len(self.cat_indexes)
        if G > 0:
            q = u_out[:, t['cat_idx']]                                         # [B, G, K]
            p = u_in[:, t['cat_idx']]
            valid = t['cat_valid'].unsqueeze(0)                                # [1, G, K]
            row_ok = torch.ones((B, G), dtype=torch.bool, device=u_in.device) if useless_cat is None else ~useless_cat
            n_rows = row_ok.sum(dim=0).to(u_out.dtype)                         # useful rows per group
            if not self.cat_bce:
                if self.cat_softmax:
                    q = torch.softmax((q / self.cat_softmax_t).masked_fill(~valid, float('-inf')), dim=2)
                target = p.bool() & valid
                # one probability per row and group (one-hot target): -sum log q_target / useful rows (loss.py:166-171)
                logq = torch.log(torch.where(target, q, torch.ones_like(q))).sum(dim=2)        # [B, G]
                per_group = -(logq * row_ok).sum(dim=0) / n_rows
            else:
                bce = F.binary_cross_entropy(q, p, reduction='none')           # [B, G, K]
                k_g = t['cat_valid'].sum(dim=1).to(u_out.dtype)                                # group sizes
                per_group = (bce * valid * row_ok.unsqueeze(2)).sum(dim=(0, 2)) / (n_rows * k_g) / 8.0
            cat_loss = per_group.sum()
            if self.normalize_losses:
                cat_loss = cat_loss / G
        return num_loss + cat_loss * self.cat_loss_factor


class QuantizedNumericalParamsLoss:
    """loss.py:187-261 (detached: a metric, not differentiable)."""

    def __init__(self, idx_helper, numerical_loss=None, limited_vst_params_indexes=None):
        self.idx_helper = idx_helper
        self.numerical_loss = numerical_loss if numerical_loss is not None else torch.nn.MSELoss()
        for vst_idx in idx_helper.num_idx_learned_as_cat:
            assert idx_helper.vst_param_cardinals[vst_idx] > 0
        self.limited_vst_params_indexes = limited_vst_params_indexes
        lim = limited_vst_params_indexes
        self._as_num = [(v, l) for v, l in idx_helper.num_idx_learned_as_num.items() if lim is None or v in lim]
        self._as_cat = [(v, list(l)) for v, l in idx_helper.num_idx_learned_as_cat.items() if lim is None or v in lim]
        self.num_params_count = len(idx_helper.num_idx_learned_as_num) + len(idx_helper.num_idx_learned_as_cat)

    @torch.no_grad()
    def __call__(self, u_out, u_in):
        dev, dt = u_in.device, u_in.dtype
        cols_in, cols_out = [], []
        if self._as_num:
            idx = torch.tensor([l for _, l in self._as_num], dtype=torch.long, device=dev)
            card = torch.tensor([float(self.idx_helper.vst_param_cardinals[v]) for v, _ in self._as_num], device=dev,
                                dtype=dt)
            o = u_out[:, idx]
            quant = torch.round(o * (card - 1.0)) / (card - 1.0)
            cols_in.append(u_in[:, idx])
            cols_out.append(torch.where(card > 0, quant, o))                   # cardinal < 0: continuous parameter
        for _, learn in self._as_cat:
            idx = torch.tensor(learn, dtype=torch.long, device=dev)
            c = float(len(learn))
            cols_in.append((torch.argmax(u_in[:, idx], dim=-1).to(dt) / (c - 1.0)).unsqueeze(1))
            cols_out.append((torch.argmax(u_out[:, idx], dim=-1).to(dt) / (c - 1.0)).unsqueeze(1))
        n_used = sum(c.shape[1] for c in cols_in)
        if self.limited_vst_params_indexes is not None and n_used < self.num_params_count:
            # the reference pre-allocates num_params_count columns and leaves the unused ones at zero (loss.py:222-224)
            pad = torch.zeros((u_in.shape[0], self.num_params_count - n_used), device=dev, dtype=dt)
            cols_in.append(pad)
            cols_out.append(pad)
        return self.numerical_loss(torch.cat(cols_out, dim=1), torch.cat(cols_in, dim=1))


class CategoricalParamsAccuracy:
    """loss.py:265-315.  ``reduce=True`` returns a 0-d device tensor (no synchronisation); ``reduce=False`` a dict of
    Python floats keyed by VST parameter index, as the reference."""

    def __init__(self, idx_helper, reduce=True, percentage_output=True, limited_vst_params_indexes=None):
        self.idx_helper = idx_helper
        self.reduce = reduce
        self.percentage_output = percentage_output
        self.limited_vst_params_indexes = limited_vst_params_indexes

    @torch.no_grad()
    def __call__(self, u_out, u_in):
        lim = self.limited_vst_params_indexes
        keys, accs = [], []
        for vst_idx, learn_idx in self.idx_helper.cat_idx_learned_as_num.items():
            if lim is not None and vst_idx not in lim:
                continue
            card = float(self.idx_helper.vst_param_cardinals[vst_idx])
            tgt = torch.round(u_in[:, learn_idx] * (card - 1.0)).to(torch.int32)
            out = torch.round(u_out[:, learn_idx] * (card - 1.0)).to(torch.int32)
            keys.append(vst_idx)
            accs.append((tgt == out).to(torch.float32).mean())
        for vst_idx, learn_indexes in self.idx_helper.cat_idx_learned_as_cat.items():
            if lim is not None and vst_idx not in lim:
                continue
            idx = torch.tensor(list(learn_indexes), dtype=torch.long, device=u_in.device)
            keys.append(vst_idx)
            accs.append((torch.argmax(u_in[:, idx], dim=-1) == torch.argmax(u_out[:, idx], dim=-1))
                        .to(torch.float32).mean())
        acc = torch.stack(accs) * (100.0 if self.percentage_output else 1.0)
        if self.reduce:
            return acc.mean()
        vals = acc.cpu().numpy()
        return {k: float(v) for k, v in zip(keys, np.asarray(vals, dtype=np.float64))}
