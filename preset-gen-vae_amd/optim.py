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
_SCRATCH = 1 << 16   # floats of step scratch behind the flat gradient (256 KB)


class FlatParams:
    def __init__(self, params):
        params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
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
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(flat.params, defaults)
        dev = flat.flat_param.device
        self.exp_avg = torch.zeros_like(flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(flat.flat_param)
        self.hyper = torch.tensor([lr, 1.0, 1.0, grad_scale], device=dev, dtype=torch.float32)
        self.pows = torch.ones(2, device=dev, dtype=torch.float64)
        self._lr = lr

    def set_lr(self, lr):
        """LR warm-up / ReduceLROnPlateau hook (train.py:195-197,296): writes the device-resident learning rate."""
        self._lr = float(lr)
        for g in self.param_groups:
            g['lr'] = float(lr)
        ops.fill(self.hyper[0:1], float(lr))

    def zero_grad(self, set_to_none=False):
        """``optimizer.zero_grad()`` of train.py:208: ONE fill of the flat gradient buffer.  The weight / bias gradient
        kernels accumulate with atomics; knowing the buffer is clean they skip their per-call memset nodes (~5 us of
        dependent-launch latency each, 16 per step)."""
        self.flat._grad_and_scratch.zero_()
        self.flat.grad_zeroed = True
        self.flat.zero_gen += 1
        return None

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        if g['lr'] != self._lr:  # a torch LR scheduler edited param_groups
            self.set_lr(g['lr'])
        b1, b2 = g['betas']
        lib = _lib.load()
        _lib.check(lib.pgv_adam_tick(self.pows.data_ptr(), self.hyper.data_ptr(), b1, b2,
                                     torch.cuda.current_stream().cuda_stream), "pgv_adam_tick")
        f = self.flat
        ops.adam_step(f.flat_param, f.flat_grad, self.exp_avg, self.exp_avg_sq, self.hyper, b1, b2, g['eps'],
                      g['weight_decay'])
        f.grad_zeroed = False
        return None
