"""Data-parallel minibatch sharding: one process per GPU, persistent model replicas, bucketed gradient all-reduce
over RCCL (xGMI) overlapped with the rest of backward.

The reference's multi-GPU mode is single-process ``nn.DataParallel`` (train.py:95-97): per step it scatters the batch,
re-broadcasts all parameters, gathers the outputs to one GPU and reduce-adds the replica gradients there
(SURVEY.md §2.3).  Here every rank owns its shard of the minibatch and a full replica; the only exchange is the SUM
all-reduce of the flat gradient buffer (``optim.FlatParams``), scaled by 1/world_size inside the fused Adam kernel
(mean of shard-means == global mean for equal shards, so losses/gradients match the gathered-batch arithmetic of
train.py:222-225).  BatchNorm batch statistics stay per rank, exactly like DataParallel's per-replica BN.

Buckets are contiguous slices of the flat gradient in gradient-ready order (decoder output layer first).  When the
last gradient of a bucket has been launched on the compute stream, an event is recorded and the bucket's all-reduce is
enqueued on a dedicated communication stream; ``wait()`` joins the streams before the optimizer step.  The same
notifications cut a CAPTURED step into one hipGraph per bucket (``capture_cuts`` / ``launch_bucket``,
``train_step.VAETrainStep._capture_bucket_graphs``): replay and overlap together.  xGMI is
point-to-point (7 links/GPU), so a few large buckets (default 4 over 19.7-49.8 MB) are preferred to many small ones.
"""
import torch
import torch.distributed as dist

from .model import layer


class GradAllReduce:
    def __init__(self, flat, n_buckets=4, process_group=None):
        self.flat = flat
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.ranges = flat.bucket_ranges(n_buckets)
        self._bucket_of = {}
        self._need = []
        for bi, (lo, hi) in enumerate(self.ranges):
            ps = flat.params_in_range(lo, hi)
            self._need.append(len(ps))
            for p in ps:
                self._bucket_of[id(p)] = bi
        self._done = [0] * len(self.ranges)
        self._launched = [False] * len(self.ranges)
        self.use_cuda = flat.flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat.flat_grad.device) if self.use_cuda else None
        self._works = []
        self._on_bucket = None
        self.n_collectives = 0     # bucket launches so far (also counted on one rank, where the collective is skipped)
        # optional timing of every bucket's collective with events on the communication stream (``time_buckets(True)``;
        # bench.py reports the averages): [(bucket, start event, end event)], read by ``bucket_times_ms``
        self._time = False
        self._timed = []

    # -- hooks ---------------------------------------------------------------------------------------------
    def install(self):
        layer.GRAD_READY_HOOK = self._on_grad_ready
        return self

    def uninstall(self):
        if layer.GRAD_READY_HOOK == self._on_grad_ready:
            layer.GRAD_READY_HOOK = None

    def start_step(self):
        self._done = [0] * len(self.ranges)
        self._launched = [False] * len(self.ranges)
        self._works = []

    def _on_grad_ready(self, p):
        bi = self._bucket_of.get(id(p))
        if bi is None:
            return
        self._done[bi] += 1
        if self._done[bi] >= self._need[bi] and not self._launched[bi]:
            self._launch(bi)

    def capture_cuts(self, on_bucket):
        """While a step is being captured into hipGraphs (``VAETrainStep._capture_bucket_graphs``): a complete bucket
        calls ``on_bucket(bi)`` instead of launching its collective - the train step cuts the capture there and launches
        the collective itself between two replays (``launch_bucket``).  None switches back."""
        self._on_bucket = on_bucket

    def launch_bucket(self, bi):
        """The all-reduce of bucket ``bi`` behind everything already queued on the current stream (bucket-graph mode:
        the graph that completes the bucket has just been replayed)."""
        if not self._launched[bi]:
            self._launch(bi)

    def _launch(self, bi):
        self._launched[bi] = True
        if self._on_bucket is not None:
            self._on_bucket(bi)
            return
        self.n_collectives += 1
        if self.world_size == 1:
            return
        lo, hi = self.ranges[bi]
        buf = self.flat.flat_grad[lo:hi]
        if self.use_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if self._time:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                if self._time:
                    e1.record(self.comm_stream)
                    self._timed.append((bi, e0, e1))
        else:  # gloo / CPU tests
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def time_buckets(self, on=True):
        """Record an event pair around every bucket's all-reduce on the communication stream from now on."""
        self._time, self._timed = bool(on), []

    def bucket_times_ms(self):
        """Average device time of each bucket's collective since ``time_buckets(True)`` (synchronises; None for a bucket
        that was never launched - always on one rank, where no collective is issued)."""
        if self.use_cuda:
            torch.cuda.synchronize()
        sums, counts = [0.0] * len(self.ranges), [0] * len(self.ranges)
        for bi, e0, e1 in self._timed:
            sums[bi] += e0.elapsed_time(e1)
            counts[bi] += 1
        return [round(s_ / c, 4) if c else None for s_, c in zip(sums, counts)]

    def exchange(self):
        """All buckets at once, after everything already queued on the compute stream (the graph launch mode: the
        backward graph has been replayed, the Adam graph follows): launch the all-reduces on the communication stream
        and make the compute stream wait for them.  Asynchronous with respect to the host."""
        self.start_step()
        self.wait()

    def wait(self):
        """Flush buckets whose parameters produced no gradient this step, then join communication."""
        for bi in range(len(self.ranges)):
            if not self._launched[bi] and self._on_bucket is None:
                self._launch(bi)
        if self.world_size == 1:
            return
        if self.use_cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w in self._works:
                w.wait()
            self._works = []
