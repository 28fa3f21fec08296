"""CPU restatement (torch CPU float64, plain loops) of the reference's preset-regression losses and metrics
(model/loss.py:72-315) and of the Dexed useless-parameter rule (data/preset.py:259-281) - SURVEY.md §8 f4.

TEST INFRASTRUCTURE ONLY.  Pinned by ``tests/golden/params_loss.npz`` (reference classes driven with the duck-typed
``tests/helpers.MiniPresetIndexesHelper``)."""
import numpy as np
import torch
import torch.nn.functional as F


def synth_params_loss(u_out, u_in, helper, normalize_losses, categorical_loss_factor=0.2,
                      prevent_useless_params_loss=True, cat_bce=True, cat_softmax=False, cat_softmax_t=0.1):
    """SynthParamsLoss.__call__ (model/loss.py:118-183), without the in-place mutation of the arguments."""
    B = u_in.shape[0]
    num_indexes = helper.get_numerical_learnable_indexes()
    cat_indexes = helper.get_categorical_learnable_indexes()
    useless_num, useless_cat = [[] for _ in range(B)], [[] for _ in range(B)]
    if prevent_useless_params_loss:                                        # loss.py:123-127
        for row in range(B):
            useless_num[row], useless_cat[row] = helper.get_useless_learned_params_indexes(u_in[row, :])
    num_loss = 0.0
    if len(num_indexes) > 0:                                               # loss.py:128-136
        keep = torch.ones_like(u_in)
        for row in range(B):
            for idx in num_indexes:
                if idx in useless_num[row]:
                    keep[row, idx] = 0.0
        a, b = (u_out * keep)[:, num_indexes], (u_in * keep)[:, num_indexes]
        if normalize_losses:
            num_loss = F.mse_loss(a, b, reduction='mean')
        else:
            num_loss = torch.sum((a - b) ** 2) / B                         # L2Loss (loss.py:15-43), batch-averaged
    cat_loss = 0.0
    for group in cat_indexes:                                              # loss.py:137-179
        rows = [r for r in range(B) if not (prevent_useless_params_loss and group[0] in useless_cat[r])]
        target, q = u_in[rows][:, group], u_out[rows][:, group]
        if not cat_bce:
            if cat_softmax:
                q = torch.softmax(q / cat_softmax_t, dim=1)
            cat_loss = cat_loss - torch.sum(torch.log(q[target.bool()])) / len(rows)
        else:
            cat_loss = cat_loss + F.binary_cross_entropy(q, target, reduction='mean') / 8.0
    if len(cat_indexes) > 0 and normalize_losses:
        cat_loss = cat_loss / len(cat_indexes)
    return num_loss + cat_loss * categorical_loss_factor


def quantized_numerical_params_loss(u_out, u_in, helper, limited_vst_params_indexes=None):
    """QuantizedNumericalParamsLoss.__call__ with nn.MSELoss (model/loss.py:213-261)."""
    n_cols = len(helper.num_idx_learned_as_num) + len(helper.num_idx_learned_as_cat)
    a = torch.zeros((u_in.shape[0], n_cols), dtype=u_in.dtype)
    b = torch.zeros_like(a)
    col = 0
    for vst_idx, learn_idx in helper.num_idx_learned_as_num.items():
        if limited_vst_params_indexes is not None and vst_idx not in limited_vst_params_indexes:
            continue
        a[:, col] = u_in[:, learn_idx]
        v = u_out[:, learn_idx].clone()
        card = helper.vst_param_cardinals[vst_idx]
        if card > 0:
            v = torch.round(v * (card - 1.0)) / (card - 1.0)
        b[:, col] = v
        col += 1
    for vst_idx, learn_indexes in helper.num_idx_learned_as_cat.items():
        if limited_vst_params_indexes is not None and vst_idx not in limited_vst_params_indexes:
            continue
        card = len(learn_indexes)
        a[:, col] = torch.argmax(u_in[:, learn_indexes], dim=-1).to(u_in.dtype) / (card - 1.0)
        b[:, col] = torch.argmax(u_out[:, learn_indexes], dim=-1).to(u_in.dtype) / (card - 1.0)
        col += 1
    return F.mse_loss(b, a)


def categorical_params_accuracy(u_out, u_in, helper, percentage_output=True, limited_vst_params_indexes=None):
    """CategoricalParamsAccuracy.__call__ (model/loss.py:281-315): dict keyed by VST parameter index."""
    acc = {}
    for vst_idx, learn_idx in helper.cat_idx_learned_as_num.items():
        if limited_vst_params_indexes is not None and vst_idx not in limited_vst_params_indexes:
            continue
        card = helper.vst_param_cardinals[vst_idx]
        t = torch.round(u_in[:, learn_idx] * (card - 1.0)).to(torch.int32)
        o = torch.round(u_out[:, learn_idx] * (card - 1.0)).to(torch.int32)
        acc[vst_idx] = (t == o).sum().item() / t.numel()
    for vst_idx, learn_indexes in helper.cat_idx_learned_as_cat.items():
        if limited_vst_params_indexes is not None and vst_idx not in limited_vst_params_indexes:
            continue
        t, o = torch.argmax(u_in[:, learn_indexes], dim=-1), torch.argmax(u_out[:, learn_indexes], dim=-1)
        acc[vst_idx] = (t == o).sum().item() / t.numel()
    if percentage_output:
        acc = {k: v * 100.0 for k, v in acc.items()}
    return acc


def dexed_useless_learned_params_indexes(full_to_learnable, preset_gt):
    """PresetIndexesHelper.get_useless_learned_params_indexes, Dexed branch (data/preset.py:259-281)."""
    base = [23, 24, 25, 26, 27, 28, 29, 30, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43]
    nums, cats = [], []
    for op_i, vol in enumerate([31 + 22 * i for i in range(6)]):
        learn_vol = full_to_learnable[vol]
        if isinstance(learn_vol, int) and float(preset_gt[learn_vol]) < 1e-3:
            for vst_idx in [i + op_i * 22 for i in base]:
                learn = full_to_learnable[vst_idx]
                if isinstance(learn, int):
                    nums.append(learn)
                elif isinstance(learn, list):
                    cats.append(learn[0])
    return nums, cats


def decode_full_to_learnable(coded):
    """Inverse of the golden file's integer coding: -1 = not learnable, v >= 0 numerical column, v <= -2 a 3-column
    one-hot group starting at -(v + 2)."""
    out = []
    for v in np.asarray(coded).tolist():
        out.append(None if v == -1 else (int(v) if v >= 0 else [-(v + 2), -(v + 2) + 1, -(v + 2) + 2]))
    return out
