"""Flat parameter storage + fused Adam (reference: ``torch.optim.Adam(lr, betas, weight_decay)`` at
train.py:166-167 — L2-coupled decay applied to every parameter, eps 1e-8, no amsgrad).

``FlatParams`` moves every parameter of a module into ONE contiguous fp32 buffer (parameters become views, so
state-dict keys and shapes are unchanged) and gives every parameter a view into ONE flat gradient buffer that the
backward kernels write directly.  That makes the optimizer a single HBM-streaming kernel over 7 passes x numel x 4 B
(read p,g,m,v; write p,m,v) instead of ~60 small launches, and makes the data-parallel gradient exchange a handful of
contiguous RCCL all-reduces (``parallel.py``).  Parameters are laid out in *reverse* registration order — decoder
output layer first, encoder input layer last — which is the order backward produces gradients, so finished gradient
buckets are contiguous prefixes.
"""
import torch

from . import _lib, ops

_ALIGN = 64  # floats (256 B): 16-byte streaming accesses, and gradient slices start on a cache-line boundary - the
# weight-gradient kernels flush with float atomics, which run ~1.5x slower when their 64-byte segments straddle lines
_SCRATCH = 1 << 20   # floats of step scratch behind the flat gradient (4 MB: the z = 512 split-K outputs [256, 1024] fit too)


class FlatParams:
    def __init__(self, params):
        params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params_natural = list(uniq)        # registration order: the order torch.optim.Adam(model.parameters()) sees
        self.params = list(reversed(uniq))
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatParams needs fp32 parameters on one device")
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.flat_param = torch.zeros(off, device=dev, dtype=torch.float32)
        # gradient buffer + "step scratch": small accumulators that must start a step at zero (BatchNorm statistics /
        # backward projections of every conv stack) live behind the gradients, so the ONE fill of zero_grad() clears
        # them too instead of one fill node (~5 us of dependent-launch latency) per arena
        self._grad_and_scratch = torch.zeros(off + _SCRATCH, device=dev, dtype=torch.float32)
        self.flat_grad = self._grad_and_scratch[:off]
        self._scratch = self._grad_and_scratch[off:]
        self._scratch_slots = {}   # key -> (offset, n_floats): fixed addresses (a captured step replays them)
        self._scratch_used = 0
        self._scratch_gen = {}     # key -> zero_gen of its last hand-out
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_param[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + n].view(p.shape)
            gv = self.flat_grad[o:o + n].view(p.shape)
            p._pgv_grad_view = gv
            p._pgv_flat = self
            p.grad = gv
        # True between zero_grad() and the next optimizer step: the gradient kernels that accumulate with atomics may
        # then skip their own clearing pass (PGV_PREZEROED)
        self.grad_zeroed = False
        self.zero_gen = 0  # bumped by every zero_grad(): a slice is clean only for the first backward after it

    def step_scratch(self, key, n, dtype):
        """``n`` zeros of ``dtype`` that were cleared by this step's ``zero_grad()``, or None (no zero_grad since the last
        optimizer step, ``key`` already served in this step, or the scratch is full): the caller then allocates."""
        if not self.grad_zeroed or self._scratch_gen.get(key) == self.zero_gen:
            return None
        n_floats = n * (2 if dtype == torch.float64 else 1)
        slot = self._scratch_slots.get(key)
        if slot is None or slot[1] < n_floats:
            need = (n_floats + _ALIGN - 1) // _ALIGN * _ALIGN
            if self._scratch_used + need > self._scratch.numel():
                return None
            slot = (self._scratch_used, need)
            self._scratch_slots[key] = slot
            self._scratch_used += need
        self._scratch_gen[key] = self.zero_gen
        return self._scratch[slot[0]:slot[0] + n_floats].view(dtype)

    def bucket_ranges(self, n_buckets):
        """Split [0, numel) at parameter boundaries into ~equal contiguous ranges (gradient-ready order)."""
        target = self.numel / max(1, n_buckets)
        ranges, start = [], 0
        ends = [o + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN for p, o in zip(self.params, self.offsets)]
        for e in ends:
            if e - start >= target and len(ranges) < n_buckets - 1:
                ranges.append((start, e))
                start = e
        if start < self.numel:
            ranges.append((start, self.numel))
        return ranges

    def params_in_range(self, lo, hi):
        return [p for p, o in zip(self.params, self.offsets) if lo <= o < hi]


class FusedAdam(torch.optim.Optimizer):
    """Adam with coupled L2 over a :class:`FlatParams` buffer, one kernel per step.

    ``lr`` lives in device memory (``set_lr``), the bias corrections advance on device (``pgv_adam_tick``), so a
    captured step replays correctly.  ``grad_scale`` multiplies the gradient first (1/world_size after a sum
    all-reduce)."""

    def __init__(self, flat: FlatParams, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        self.flat = flat
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False)
        # parameters in REGISTRATION order, so that state_dict() indices mean what they mean in the reference's
        # torch.optim.Adam(extended_ae_model.parameters()) (train.py:166-167): checkpoints are interchangeable
        super().__init__(flat.params_natural, defaults)
        dev = flat.flat_param.device
        self.exp_avg = torch.zeros_like(flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(flat.flat_param)
        self.hyper = torch.tensor([lr, 1.0, 1.0, grad_scale], device=dev, dtype=torch.float32)
        self.pows = torch.tensor([1.0, 1.0, 0.0], device=dev, dtype=torch.float64)   # beta1^t, beta2^t, t
        self._lr = lr
        self._offset_of = {id(p): o for p, o in zip(flat.params, flat.offsets)}

    def set_lr(self, lr):
        """LR warm-up / ReduceLROnPlateau hook (train.py:195-197,296): writes the device-resident learning rate."""
        self._lr = float(lr)
        for g in self.param_groups:
            g['lr'] = float(lr)
        if self.hyper.is_cuda:
            ops.fill(self.hyper[0:1], float(lr))
        else:
            self.hyper[0] = float(lr)

    def sync_lr(self):
        """Pick up a learning rate that a torch scheduler (or train.py:195-197's warm-up loop) wrote into
        ``param_groups``.  ``step()`` does it on every eager step; a captured step calls it before each replay."""
        lr = self.param_groups[0]['lr']
        if lr != self._lr:
            self.set_lr(lr)

    def zero_grad(self, set_to_none=False):
        """``optimizer.zero_grad()`` of train.py:208: ONE fill of the flat gradient buffer.  The weight / bias gradient
        kernels accumulate with atomics; knowing the buffer is clean they skip their per-call memset nodes (~5 us of
        dependent-launch latency each, 16 per step)."""
        self.flat._grad_and_scratch.zero_()
        self.flat.grad_zeroed = True
        self.flat.zero_gen += 1
        return None

    @torch.no_grad()
    def step(self, closure=None, rng_advance=None, loss_total=None):
        """``rng_advance`` = (state tensor, increment) and ``loss_total`` = (a, b, weight_b, c or None, total[, finite]):
        bookkeeping of the train step that rides in the step-counter launch (``pgv_step_tick``): the generator offset, the
        reported total = a + b * weight_b (+ c) and the all-losses-finite flag."""
        g = self.param_groups[0]
        self.sync_lr()
        b1, b2 = g['betas']
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        if rng_advance is None and loss_total is None:
            _lib.check(lib.pgv_adam_tick(self.pows.data_ptr(), self.hyper.data_ptr(), b1, b2, st), "pgv_adam_tick")
        else:
            rs, inc = rng_advance if rng_advance is not None else (None, 0)
            la, lb, wb, lc, tot, fin = (tuple(loss_total) + (None,))[:6] if loss_total is not None else (None,) * 6
            ptr = lambda t: None if t is None else t.data_ptr()   # noqa: E731
            _lib.check(lib.pgv_step_tick(self.pows.data_ptr(), self.hyper.data_ptr(), b1, b2, ptr(rs), int(inc), ptr(la),
                                         ptr(lb), ptr(wb), ptr(lc), ptr(tot), ptr(fin), st), "pgv_step_tick")
        f = self.flat
        ops.adam_step(f.flat_param, f.flat_grad, self.exp_avg, self.exp_avg_sq, self.hyper, b1, b2, g['eps'],
                      g['weight_decay'])
        f.grad_zeroed = False
        return None

    # -- checkpoints (logs/logger.py:199-202 saves optimizer.state_dict(), train.py:177-179 restores it) ----------
    def step_count(self):
        """Number of updates applied so far: an explicit device-side counter next to the powers of beta (it advances on
        the device so that captured steps replay correctly; reading it synchronises - checkpoint time only).  The count
        is NOT recovered from beta1^t, which underflows to zero after ~7000 steps."""
        return int(round(float(self.pows[2].item())))

    def state_dict(self):
        """torch.optim.Adam's layout: ``state[i] = {'step', 'exp_avg', 'exp_avg_sq'}`` per parameter i of
        ``model.parameters()`` order plus ``param_groups`` - loadable by ``torch.optim.Adam.load_state_dict`` (and so by
        the reference), and what ``load_state_dict`` below accepts."""
        sd = super().state_dict()
        t = self.step_count()
        state = {}
        if t > 0:
            for i, p in enumerate(self.flat.params_natural):
                o, n = self._offset_of[id(p)], p.numel()
                state[i] = {'step': torch.tensor(float(t)),
                            'exp_avg': self.exp_avg[o:o + n].view(p.shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        sd['state'] = state
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        groups = state_dict['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.flat.params_natural):
            raise ValueError("optimizer state_dict does not match this model: expected ONE parameter group of "
                             f"{len(self.flat.params_natural)} parameters")
        if groups[0].get('amsgrad', False):
            raise ValueError("amsgrad state cannot be loaded (train.py:166-167 uses plain Adam)")
        g = self.param_groups[0]
        for k in ('betas', 'eps', 'weight_decay'):
            if k in groups[0]:
                g[k] = tuple(groups[0][k]) if k == 'betas' else groups[0][k]
        state = state_dict['state']
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, p in enumerate(self.flat.params_natural):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            o, n = self._offset_of[id(p)], p.numel()
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of parameter {i}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
            self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(st['step']))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one fused step counter only")
        t = steps.pop() if steps else 0
        b1, b2 = g['betas']
        self.pows.copy_(torch.tensor([b1 ** t, b2 ** t, float(t)], dtype=torch.float64))
        self.hyper[1:3].copy_(torch.tensor([1.0 - b1 ** t, 1.0 - b2 ** t], dtype=torch.float32))
        self.set_lr(groups[0]['lr'])

    def snapshot(self):
        """Copies of everything a step changes on the optimizer side (graph-capture warm-up, train_step.py)."""
        return (self.flat.flat_param.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.pows.clone(),
                self.hyper.clone())

    def restore(self, snap):
        for dst, src in zip((self.flat.flat_param, self.exp_avg, self.exp_avg_sq, self.pows, self.hyper), snap):
            dst.copy_(src)
