"""Preset-regression losses and metrics evaluated ON THE DEVICE by the HIP kernels of ``csrc/params_loss.hip``, without
host synchronisation.

Mirrors of the reference's ``model/loss.py``:

* ``SynthParamsLoss`` (:72-183): MSE / L2 over the numerical learnable columns plus a categorical term per one-hot
  group (categorical cross-entropy, optionally with a temperature softmax, or binary cross-entropy / 8), useless
  parameters (e.g. a Dexed operator with zero output level) excluded per row;
* ``QuantizedNumericalParamsLoss`` (:187-261): numerical VST parameters after the synth's quantisation;
* ``CategoricalParamsAccuracy`` (:265-315).

The reference walks rows, parameters and groups in Python and calls ``.item()`` per group; here the helper's index
lists become device tables once (``ops.params_tables`` / ``ops.params_item_tables``) and a call is ONE launch:
``pgv_params_loss`` returns the loss and its gradient w.r.t. the network output (a workgroup per row, a thread per
numerical column / one-hot group, deterministic cross-workgroup sum), ``pgv_params_columns`` the column pairs and match
rates of the two metrics (+ ``pgv_sqerr_fwd`` for the quantised MSE).  ``CategoricalParamsAccuracy`` with
``reduce=False`` is the only path that reads values back, once.  Unlike the reference (loss.py:134-135) the inputs are
not modified in place.  Like every other op of the package there is no CPU path: tensors must live on a ROCm device.

Useless-parameter rules: ``idx_helper.useless_rules`` = ``[(trigger_learn_idx, [num_learn_idx...],
[cat_first_learn_idx...]), ...]`` (a row's listed parameters are useless when ``u_in[row, trigger] < 1e-3``) if the
helper provides it, else the Dexed rule built from ``full_to_learnable`` exactly as ``data/preset.py:259-281`` does,
else none."""
import numpy as np
import torch

from .. import ops


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


class _ParamsLossFn(torch.autograd.Function):
    """loss, d loss / d u_out from one ``pgv_params_loss`` launch; backward scales the stored gradient."""

    @staticmethod
    def forward(ctx, u_out, u_in, crit):
        loss, grad = ops.params_loss(u_out, u_in, crit._tables_on(u_in.device), crit._mode, crit.cat_softmax_t,
                                     crit.normalize_losses, crit.cat_loss_factor, want_grad=u_out.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return (grad * g if grad is not None else None), None, None


def _f32c(t):
    return t.detach().float().contiguous() if not (t.dtype == torch.float32 and t.is_contiguous()) else t


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
        self.num_indexes = list(idx_helper.get_numerical_learnable_indexes())
        self.cat_indexes = [list(g) for g in idx_helper.get_categorical_learnable_indexes()]
        self._rules = _useless_rules(idx_helper) if prevent_useless_params_loss else []
        self._mode = ops.PARAMS_BCE if cat_bce else (ops.PARAMS_CCE_SOFTMAX if cat_softmax else ops.PARAMS_CCE)
        self._tables = {}

    def _tables_on(self, device):
        t = self._tables.get(device)
        if t is None:
            t = self._tables[device] = ops.params_tables(device, self.num_indexes, self.cat_indexes, self._rules)
        return t

    def __call__(self, u_out, u_in):
        """Categorical parameters must be one-hot encoded.  Returns a 0-d tensor on the inputs' device."""
        u_in = _f32c(u_in)
        if u_out.dtype != torch.float32 or not u_out.is_contiguous():
            u_out = u_out.float().contiguous()
        return _ParamsLossFn.apply(u_out, u_in, self)


class QuantizedNumericalParamsLoss:
    """loss.py:187-261 (detached: a metric, not differentiable).  ``numerical_loss`` None = nn.MSELoss() evaluated by
    ``pgv_sqerr_fwd``; any other criterion is called on the two column matrices ``pgv_params_columns`` produced."""

    def __init__(self, idx_helper, numerical_loss=None, limited_vst_params_indexes=None):
        self.idx_helper = idx_helper
        if isinstance(numerical_loss, torch.nn.MSELoss) and numerical_loss.reduction == 'mean':
            numerical_loss = None
        self.numerical_loss = numerical_loss
        for vst_idx in idx_helper.num_idx_learned_as_cat:
            assert idx_helper.vst_param_cardinals[vst_idx] > 0
        self.limited_vst_params_indexes = limited_vst_params_indexes
        lim = limited_vst_params_indexes
        self._items = [(ops._lib.PGV_PARAMS_COL_QUANTIZED, l, idx_helper.vst_param_cardinals[v])
                       for v, l in idx_helper.num_idx_learned_as_num.items() if lim is None or v in lim]
        self._items += [(ops._lib.PGV_PARAMS_COL_ONEHOT_VALUE, list(l), len(l))
                        for v, l in idx_helper.num_idx_learned_as_cat.items() if lim is None or v in lim]
        self.num_params_count = len(idx_helper.num_idx_learned_as_num) + len(idx_helper.num_idx_learned_as_cat)
        self._tables = {}

    @torch.no_grad()
    def __call__(self, u_out, u_in):
        u_out, u_in = _f32c(u_out), _f32c(u_in)
        t = self._tables.get(u_in.device)
        if t is None:
            t = self._tables[u_in.device] = ops.params_item_tables(u_in.device, self._items)
        B, n_used = u_in.shape[0], len(self._items)
        # the reference pre-allocates num_params_count columns and leaves the unused ones at zero (loss.py:222-224)
        n_cols = self.num_params_count if self.limited_vst_params_indexes is not None else n_used
        if n_used == 0:
            cols_in = cols_out = torch.zeros((B, n_cols), device=u_in.device)
        else:
            cols_in, cols_out, _ = ops.params_columns(u_out, u_in, t, want_match=False)
        if self.numerical_loss is None:
            return ops.sqerr_fwd(cols_out, cols_in, 1.0 / (B * n_cols)) if n_used else cols_in.sum() / (B * n_cols)
        if n_cols > n_used:
            pad = torch.zeros((B, n_cols - n_used), device=u_in.device)
            cols_in, cols_out = torch.cat([cols_in, pad], dim=1), torch.cat([cols_out, pad], dim=1)
        return self.numerical_loss(cols_out, cols_in)


class CategoricalParamsAccuracy:
    """loss.py:265-315.  ``reduce=True`` returns a 0-d device tensor (no synchronisation); ``reduce=False`` a dict of
    Python floats keyed by VST parameter index, as the reference."""

    def __init__(self, idx_helper, reduce=True, percentage_output=True, limited_vst_params_indexes=None):
        self.idx_helper = idx_helper
        self.reduce = reduce
        self.percentage_output = percentage_output
        self.limited_vst_params_indexes = limited_vst_params_indexes
        lim = limited_vst_params_indexes
        self._keys, self._items = [], []
        for vst_idx, learn_idx in idx_helper.cat_idx_learned_as_num.items():
            if lim is None or vst_idx in lim:
                self._keys.append(vst_idx)
                self._items.append((ops._lib.PGV_PARAMS_COL_CLASS, learn_idx, idx_helper.vst_param_cardinals[vst_idx]))
        for vst_idx, learn_indexes in idx_helper.cat_idx_learned_as_cat.items():
            if lim is None or vst_idx in lim:
                self._keys.append(vst_idx)
                self._items.append((ops._lib.PGV_PARAMS_COL_ONEHOT_CLASS, list(learn_indexes), len(learn_indexes)))
        self._tables = {}

    @torch.no_grad()
    def __call__(self, u_out, u_in):
        u_out, u_in = _f32c(u_out), _f32c(u_in)
        t = self._tables.get(u_in.device)
        if t is None:
            t = self._tables[u_in.device] = ops.params_item_tables(u_in.device, self._items)
        _, _, match = ops.params_columns(u_out, u_in, t, want_cols=False)
        acc = match * (100.0 if self.percentage_output else 1.0)
        if self.reduce:
            return acc.mean()
        vals = acc.cpu().numpy()
        return {k: float(v) for k, v in zip(self._keys, np.asarray(vals, dtype=np.float64))}
