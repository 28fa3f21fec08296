"""The VAE train step — harness counterpart of the reference's minibatch body (train.py:203-248):

    optimizer.zero_grad(); ae_out = model(x); recons = MSE(x_out, x); lat = latent_loss(...) * beta;
    (recons + lat [+ controls]).backward(); optimizer.step()

Everything between the input tensor and the updated parameters runs in the HIP kernels; this class only sequences
them, optionally capturing the whole step into one hipGraph (``torch.cuda.CUDAGraph`` records the launches our C ABI
makes on the capturing stream) so the ~100 launches replay without host involvement, and optionally exchanging
gradients across ranks (``parallel.GradAllReduce``).

Launch modes:
  eager, 1 rank      every kernel launched from Python;
  graph, 1 rank      one hipGraph = zero_grad + forward + losses + backward + Adam;
  eager, N ranks     bucketed all-reduce launched from gradient-ready hooks, overlapped with the rest of backward;
  graph, N ranks     ``graph_buckets=True`` (default): the step is cut at the gradient-bucket boundaries into k + 2
                     hipGraphs - [zero_grad + forward + backward until bucket 0 is complete], [... until bucket 1], ...,
                     [rest of backward], [Adam].  Between two replays the host enqueues that bucket's all-reduce on the
                     communication stream (an ordinary RCCL launch behind an event of the compute stream; no collective is
                     captured), so the exchange of bucket i overlaps the backward graphs that follow it AND the ~50
                     compute launches replay from the device: k + 2 graph launches and k collective launches per step
                     instead of ~100 kernel launches from Python.
                     ``graph_buckets=False``: two hipGraphs around the whole exchange ([zero_grad .. backward], all
                     buckets, [Adam]) - replay without overlap (round 3's form, kept for A/B timing)."""
import torch

from . import optim as optim_mod
from .model import loss as loss_mod


class VAETrainStep:
    def __init__(self, ae_model, lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-4, beta=0.2, normalize_losses=True,
                 reg_model=None, grad_sync=None, use_graph=False, controls_criterion=None, monitors=None,
                 graph_buckets=True, fp32_products=None, input_producer=None):
        """``input_producer``: callable(buffer) that WRITES the step's minibatch into ``buffer`` (the tensor passed to ``step``
        - in graph mode ``static_input``) as the first launches of the step: in graph mode they are captured with it, so a
        front-end that turns a resident raw-audio minibatch into spectrograms (``MelSpectrogram.batch(wav, out=buffer)``,
        BASELINE config 5) costs no launch of its own between two replays.  The producer must be capture-safe (no host
        synchronisation, its own inputs in buffers that outlive the step) and is called in the two warm-up steps as well.
        ``fp32_products``: 'bf16x6' / 'native' selects the form of the fp32 products for the whole process
        (``ops.set_fp32_products``; None keeps the current setting, whose default is 'bf16x6' - the mode bench.py times).
        ``controls_criterion``: callable(v_out, v_in) -> 0-d loss, the backprop criterion of the preset-regression
        output (train.py:108-116: ``model.params_loss.SynthParamsLoss``; default: MSE over all columns, the numeric branch
        of that loss on an all-numerical representation).  ``monitors``: {name: callable(v_out, v_in)} evaluated under
        ``no_grad`` on every minibatch BEFORE the controls criterion, as train.py:229-233 does with
        ``QuantizedNumericalParamsLoss`` / ``CategoricalParamsAccuracy``; their values come back as
        ``out['monitors'][name]`` (device tensors, no host synchronisation; captured with the rest of the step)."""
        if fp32_products is not None:
            from . import ops
            ops.set_fp32_products(fp32_products)
        self.model = ae_model
        self.reg_model = reg_model
        self.input_producer = input_producer
        self.beta = float(beta)
        params = list(ae_model.parameters()) + (list(reg_model.parameters()) if reg_model is not None else [])
        self.flat = optim_mod.FlatParams(params)
        world = 1
        self.grad_sync = None
        if grad_sync is not None:
            self.grad_sync = grad_sync(self.flat) if callable(grad_sync) else grad_sync
            world = self.grad_sync.world_size
            if not use_graph:
                self.grad_sync.install()      # gradient-ready hooks (the graph mode exchanges between its two graphs)
        self.optimizer = optim_mod.FusedAdam(self.flat, lr=lr, betas=betas, weight_decay=weight_decay,
                                             grad_scale=1.0 / world)
        if normalize_losses:   # train.py:103-106
            self.recons_criterion = loss_mod.MSELoss(reduction='mean')
        else:
            self.recons_criterion = loss_mod.L2Loss()
        # let the model evaluate this criterion inside the decoder's output stack (fused backward, VAE.py)
        vae = getattr(ae_model, 'ae_model', ae_model)
        if hasattr(vae, 'fuse_recons_criterion'):
            # (deferred: this class always runs backward before anybody reads the loss values)
            # (unit: total = 1 * recons + beta * latent below, and x_out feeds nothing else that is differentiated)
            vae.fuse_recons_criterion = ('mse_mean' if normalize_losses else 'l2_batch') + '+deferred+unit'
        self.controls_criterion = controls_criterion if controls_criterion is not None else \
            loss_mod.MSELoss(reduction='mean')
        self.monitors = dict(monitors or {})
        self.use_graph = use_graph
        self.graph_buckets = bool(graph_buckets)
        self._bucket_graphs = None     # [(hipGraph, [buckets complete when it has run])], bucket-graph mode
        self._const = {}
        self._graph = None
        self._graph_update = None
        self._static_x = None
        self._stage_x = None           # second input buffer of prefetch_input / step_prefetched
        self._stage_pending = False
        self._static_v = None
        self._out = None

    # -- one eager step --------------------------------------------------------------------------------------
    def _step_body(self, x, v_in=None, inject=None):
        out = self._forward_backward(x, v_in, inject, hooks=True)
        if self.grad_sync is not None:
            self.grad_sync.wait()
        self._optimizer_step(out)
        return out

    def _optimizer_step(self, out):
        """Adam, with the step's single-thread bookkeeping in its step-counter launch: the generator offset the forward
        left to us (``DeviceRNG.hand_over``) and the reported total loss."""
        rng = self._rng()
        inc = rng.take_owed() if rng is not None else 0
        self.optimizer.step(rng_advance=(rng.state, inc) if inc else None, loss_total=out.pop('_total_terms'))

    def _rng(self):
        vae = getattr(self.model, 'ae_model', self.model)
        return getattr(vae, '_pgv_rng_obj', None)

    def _forward_backward(self, x, v_in=None, inject=None, hooks=False):
        inject = inject or {}
        if self.input_producer is not None:
            self.input_producer(x)
        pending = self._rng()
        if pending is not None and pending.owed:
            # a forward whose optimizer step never came (gradient accumulation, a loss probe) left its generator advance
            # with us: apply it now, or this forward would draw the same Dropout masks and eps again
            from . import ops
            ops.rng_advance(pending.state, pending.take_owed())
        self.optimizer.zero_grad()
        if self.grad_sync is not None and hooks:
            self.grad_sync.start_step()
        rng = None
        if hasattr(getattr(self.model, 'ae_model', self.model), 'encoder'):
            from .rng import device_rng
            rng = device_rng(getattr(self.model, 'ae_model', self.model), x.device)
            rng.hand_over = True     # its advance rides in the optimizer's step-counter launch (_optimizer_step)
        from .model import layer as layer_mod
        layer_mod.UNIT_RECONS_GRADIENT = True      # (the roots and root gradients below keep the promise)
        try:
            z_mu_logvar, z0, zK, ladj, x_out = self.model(x, None, **inject)
        finally:
            layer_mod.UNIT_RECONS_GRADIENT = False
            if rng is not None:
                rng.hand_over = False
        recons = self.recons_criterion(x_out, x)
        lat = self.model.latent_loss(z_mu_logvar, z0, zK, ladj)
        # total = recons + lat * beta (+ controls); total.backward() (train.py:227,246-247).  The gradients of the
        # terms are the constants (1, beta, 1): they are handed to autograd as constant tensors, so neither the sum
        # nor its backward costs arithmetic launches on 0-d tensors (5 us of dependent-launch latency each);
        # the reported total is one fused multiply-add
        one, beta_t = self._constants(x.device)
        roots, root_grads = [recons, lat], [one, beta_t]
        cont, mon = None, None
        if self.reg_model is not None and v_in is not None:
            v_out = self.reg_model(zK)
            if self.monitors:                                    # train.py:229-233
                with torch.no_grad():
                    mon = {name: fn(v_out.detach(), v_in) for name, fn in self.monitors.items()}
            cont = self.controls_criterion(v_out, v_in)          # train.py:238-239
            roots.append(cont)
            root_grads.append(one)
        torch.autograd.backward(roots, root_grads)
        # total = recons + lat * beta (+ controls), evaluated by the optimizer's step-counter launch (after backward:
        # recons may be a deferred value that the backward kernels deliver)
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            total = torch.empty((), device=x.device, dtype=torch.float32)   # (written by the captured optimizer launch)
            finite = torch.empty((), device=x.device, dtype=torch.float32)
        else:   # NaN until _optimizer_step has run: a caller of _forward_backward alone never reads garbage
            total = torch.full((), float('nan'), device=x.device, dtype=torch.float32)
            finite = torch.zeros((), device=x.device, dtype=torch.float32)
        # 'finite': 1.0 when every loss term of the step is finite (pgv_step_tick) - what the reference's harness tests with
        # utils.exception.check_nan_values before it raises ModelConvergenceError (train.py:245), here without a launch or a
        # synchronisation of its own: a harness reads it when it logs (``if not out['finite']: raise ...``)
        terms = (recons.detach(), lat.detach(), beta_t, None if cont is None else cont.detach(), total, finite)
        return {'recons': recons.detach(), 'latent': lat.detach(), 'total': total, 'finite': finite, '_total_terms': terms,
                'controls': None if cont is None else cont.detach(), 'z_mu_logvar': z_mu_logvar.detach(),
                'x_out': x_out.detach(), 'monitors': mon}

    def _constants(self, device):
        """(1, beta) as device scalars.  beta is written IN PLACE when ``self.beta`` changed (``set_beta``, or a caller
        assigning the attribute as train.py:227's ``Sched/beta`` ramp would): a captured step reads the same address."""
        c = self._const.get(str(device))
        if c is None:
            c = self._const[str(device)] = [torch.ones((), device=device), torch.full((), self.beta, device=device),
                                            self.beta]
        if c[2] != self.beta:
            c[1].fill_(self.beta)
            c[2] = self.beta
        return c[0], c[1]

    def set_beta(self, beta):
        """KL weight schedule (config.py:114-117: beta ramps 0.1 -> 0.2 over 25 epochs, applied at train.py:227)."""
        self.beta = float(beta)
        for dev in list(self._const):
            self._constants(torch.device(dev))

    def set_lr(self, lr):
        """Learning-rate schedule (warm-up train.py:195-197, ReduceLROnPlateau train.py:296).  Editing
        ``optimizer.param_groups[...]['lr']`` as torch schedulers do works too: ``step`` syncs it before every launch."""
        self.optimizer.set_lr(lr)

    def step(self, x, v_in=None, inject=None):
        """One minibatch.  Returns the losses / outputs as device tensors.  They are buffers of the step - in graph mode
        the captured step's static buffers, in eager mode ``z_mu_logvar`` may live in the optimizer's step scratch (the
        encoder's split-K linear layer accumulates into a slice cleared by ``zero_grad``): read them before the next
        ``step`` overwrites them.  Eager mode returns tensors the caller owns (``_owned``)."""
        if not self.use_graph or inject:
            return self._owned(self._step_body(x, v_in, inject))
        if self._graph is None:
            self._capture(x, v_in)
        # schedules: the captured kernels read lr / beta from device memory; refresh those words when the host-side
        # values (param_groups['lr'] edited by a scheduler, self.beta) changed since the last launch
        self.optimizer.sync_lr()
        self._constants(self._static_x.device)
        # a loader that writes its minibatch straight into ``static_input`` (and passes that tensor) skips the copy
        if self.input_producer is None and x.data_ptr() != self._static_x.data_ptr():   # (a producer fills it inside the graph)
            self._static_x.copy_(x, non_blocking=True)
        if v_in is not None and v_in.data_ptr() != self._static_v.data_ptr():
            self._static_v.copy_(v_in, non_blocking=True)
        if self._bucket_graphs is not None:
            self.grad_sync.start_step()
            for graph, ready in self._bucket_graphs:
                if graph is not None:
                    graph.replay()
                for bi in ready:                 # behind an event of the compute stream, on the communication stream
                    self.grad_sync.launch_bucket(bi)
            self.grad_sync.wait()                # (buckets nobody announced + join of the two streams)
            self._graph_update.replay()
            return self._out
        self._graph.replay()
        if self._graph_update is not None:
            self.grad_sync.exchange()
            self._graph_update.replay()
        return self._out

    def _owned(self, out):
        """Eager mode hands out tensors the caller may keep across steps (epoch-level latent metrics keep a list of
        them): anything that lives in the optimizer's step scratch - ``z_mu_logvar`` when the encoder's split-K linear
        layer accumulated into a zero_grad'ed slice, the Dkl word - would be cleared by the next ``zero_grad`` and
        rewritten by the next step, so it is copied into a tensor of its own.  (Graph mode returns the captured step's
        static buffers by contract: see ``step``.)"""
        scratch = self.flat._grad_and_scratch.untyped_storage().data_ptr()
        for k, t in out.items():
            if torch.is_tensor(t) and t.untyped_storage().data_ptr() == scratch:
                out[k] = t.clone()
        return out

    def prefetch_input(self, host_x):
        """Start moving the NEXT minibatch to the device while the current step runs (graph mode, after the first step):
        the pinned host tensor is copied on a copy stream of its own into a second device buffer; ``step_prefetched()`` then
        moves it into the captured step's input buffer with one device-to-device copy (91 MB: ~35 us) in front of the
        replay.  The captured step reads its input until the end of backward (the first layer's weight gradient), so a
        loader that wrote into ``static_input`` directly would have to wait for the whole step - 1.7 ms of PCIe time in
        series with 1.9 ms of step; this way the two overlap.  One minibatch in flight at a time.

        ``host_x``: same shape and dtype as the captured input, pinned.  The caller must leave it untouched until the copy
        has been made: ``stage_ready`` (a ``torch.cuda.Event`` recorded behind the copy) tells - ``stage_ready.synchronize()``
        or ``.query()`` before the loader reuses the buffer; ``step_prefetched`` waits for it on the device."""
        if not self.use_graph:
            raise RuntimeError("prefetch_input / step_prefetched belong to the graph mode (use_graph=True): an eager step "
                               "reads the tensor it is given, there is no captured input buffer to stage into")
        if self.input_producer is not None:
            raise RuntimeError("prefetch_input: this step has an input_producer - the captured step writes its own input "
                               "buffer (stage the producer's source, e.g. the waveform buffer, instead)")
        if self._static_x is None:
            raise RuntimeError("prefetch_input: run one step first (the captured step's input buffer does not exist yet)")
        if tuple(host_x.shape) != tuple(self._static_x.shape) or host_x.dtype != self._static_x.dtype:
            raise ValueError(f"prefetch_input: expected a {tuple(self._static_x.shape)} {self._static_x.dtype} minibatch (the "
                             f"captured step's shape; a short last batch must be padded or stepped eagerly), got "
                             f"{tuple(host_x.shape)} {host_x.dtype}")
        if not host_x.is_cuda and not host_x.is_pinned():
            raise ValueError("prefetch_input: the host minibatch must be in pinned memory (torch.Tensor.pin_memory / "
                             "DataLoader(pin_memory=True)): a pageable source makes the copy synchronous - no overlap")
        if self._stage_x is None:
            self._stage_x = torch.empty_like(self._static_x)
            self._copy_stream = torch.cuda.Stream(device=self._static_x.device)
            self._stage_ready = torch.cuda.Event()
            self._stage_free = torch.cuda.Event()
            self._stage_free.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(self._stage_free)     # the previous minibatch has left the staging buffer
            self._stage_x.copy_(host_x, non_blocking=True)
            self._stage_ready.record(self._copy_stream)
        self._stage_pending = True

    def step_prefetched(self, v_in=None):
        """One step on the minibatch handed to ``prefetch_input``."""
        if not getattr(self, '_stage_pending', False):
            raise RuntimeError("step_prefetched: no minibatch was handed to prefetch_input")
        cur = torch.cuda.current_stream()
        cur.wait_event(self._stage_ready)
        self._static_x.copy_(self._stage_x, non_blocking=True)
        self._stage_free.record(cur)
        self._stage_pending = False
        return self.step(self._static_x, v_in)

    @property
    def stage_ready(self):
        """Event recorded behind the last ``prefetch_input`` copy (None before the first): the host buffer handed to
        ``prefetch_input`` may be reused once it has completed."""
        return getattr(self, '_stage_ready', None)

    @property
    def static_input(self):
        """The captured graph's input buffer (None before the first graph step): fill it in place - e.g. as the
        output of ``MelSpectrogram.batch`` or the destination of the host-to-device copy - and pass it to ``step``."""
        return self._static_x

    def _mutable_state(self):
        """Every tensor a step changes besides the flat parameter / Adam buffers: BatchNorm running statistics and
        counters, the device RNG states."""
        mods = [self.model] + ([self.reg_model] if self.reg_model is not None else [])
        ts = []
        for m in mods:
            ts += [b for b in m.buffers()]
            for sub in m.modules():
                rng = getattr(sub, '_pgv_rng_obj', None)
                if rng is not None:
                    ts.append(rng.state)
        return ts

    def _capture(self, x, v_in):
        self._static_x = x.clone()
        self._static_v = None if v_in is None else v_in.clone()
        # warm-up on a side stream (allocator pools, lazy module state), then capture.  The warm-up runs REAL steps
        # (on N ranks including the exchange); everything they changed - parameters, Adam moments and step count,
        # BatchNorm running statistics, RNG offsets - is put back afterwards, so that the first replay is step 1 of
        # the run exactly as in the eager mode and in the reference
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        from .rng import device_rng
        vae = getattr(self.model, 'ae_model', self.model)
        if hasattr(vae, 'encoder'):
            device_rng(vae, self._static_x.device)        # the generator exists before its state is snapshotted
        with torch.cuda.stream(s):
            snap_opt = self.optimizer.snapshot()
            state = self._mutable_state()
            snap_state = [t.clone() for t in state]
            for _ in range(2):
                self._step_body(self._static_x, self._static_v)
            self.optimizer.restore(snap_opt)
            for t, c in zip(state, snap_state):
                t.copy_(c)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        # (capture_error_mode 'thread_local': another thread of the process - a data-loader worker pinning memory, a logger
        # calling into the runtime - must not invalidate the capture)
        if self.grad_sync is None:
            with torch.cuda.graph(self._graph, capture_error_mode='thread_local'):
                self._out = self._step_body(self._static_x, self._static_v)
            return
        if self.graph_buckets:
            self._capture_bucket_graphs()
            return
        # N ranks: [zero_grad + forward + backward] | eager exchange | [Adam]; the capture itself leaves the parameters
        # untouched (captured work does not run), so every rank still holds identical replicas afterwards
        with torch.cuda.graph(self._graph, capture_error_mode='thread_local'):
            self._out = self._forward_backward(self._static_x, self._static_v)
        self._graph_update = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph_update, capture_error_mode='thread_local'):
            self._optimizer_step(self._out)

    def _capture_bucket_graphs(self):
        """Cut the captured step at the gradient-bucket boundaries.  The capture runs the ordinary step body with the
        gradient-ready hook installed; when the last gradient of a bucket has been launched, the hook ENDS the running
        capture and begins the next one on the same stream and memory pool - nothing executes and nothing is enqueued
        in between, so the k + 1 graphs replayed back to back are exactly the one-graph step.  Autograd's device
        thread is switched off for the capture (``set_multithreading_enabled(False)``): stream capture has to be ended by
        the thread that began it, and the hooks fire from inside ``Function.backward``."""
        from .model import layer
        sync = self.grad_sync
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        graphs, state = [], {'graph': None, 'ready': []}

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(capture_error_mode='thread_local', **({'pool': graphs[0][0].pool()} if graphs else {}))
            state['graph'], state['ready'] = g, []

        def cut():
            # (a capture that recorded no launch - the tail behind the last bucket: backward ends with the gradient that
            # completes it - is not kept: replaying an empty graph every step costs a launch for nothing; torch reports
            # such a capture with a warning, which is the only way to tell)
            import warnings
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter('always')
                state['graph'].capture_end()
            empty = any('empty' in str(w.message).lower() for w in caught)
            if not empty or not graphs:
                graphs.append((state['graph'], state['ready']))
            elif state['ready']:
                graphs.append((None, state['ready']))

        def on_bucket(bi):                       # called by GradAllReduce in place of the collective launch
            state['ready'].append(bi)
            cut()
            begin()

        import gc
        gc.collect()
        torch.cuda.synchronize()
        prev_hook = layer.GRAD_READY_HOOK
        with torch.cuda.stream(stream), torch.autograd.set_multithreading_enabled(False):
            sync.capture_cuts(on_bucket)
            layer.GRAD_READY_HOOK = sync._on_grad_ready
            try:
                begin()
                self._out = self._forward_backward(self._static_x, self._static_v, hooks=True)
                cut()
                state['graph'] = self._graph_update = torch.cuda.CUDAGraph()
                self._graph_update.capture_begin(pool=graphs[0][0].pool(), capture_error_mode='thread_local')
                self._optimizer_step(self._out)
                self._graph_update.capture_end()
                state['graph'] = None
            except BaseException:
                # a failure inside the step body leaves the stream capturing: end the running capture (its graph is
                # dropped) so the caller's error is the step's, not "operation not permitted when stream is capturing" on
                # the next unrelated call
                g = state.get('graph')
                if g is not None and torch.cuda.is_current_stream_capturing():
                    try:
                        g.capture_end()
                    except RuntimeError:
                        pass
                self._graph = self._graph_update = self._bucket_graphs = None
                raise
            finally:
                layer.GRAD_READY_HOOK = prev_hook
                sync.capture_cuts(None)
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        self._bucket_graphs = graphs
        self._graph = graphs[0][0]
