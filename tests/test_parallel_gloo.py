"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (parallel.GradAllReduce over optim.FlatParams).

The model arithmetic itself only exists on the GPU, so each rank fills its flat gradient with the ORACLE's gradients
of its own shard (per-shard BatchNorm, exactly the nn.DataParallel semantics of train.py:95) and fires the
gradient-ready hooks in backward order; after wait() every rank must hold sum over ranks, and 1/N of it must equal
the average the single-process emulation computes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import param_shapes, synth_input, synth_vec


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_grads(rank, world):
    from oracle import vae_oracle as vo
    arch, dz = 'speccnn4l1_bn', 8
    tpl = param_shapes(arch, dz, False)
    sd = vo.closed_form_state_dict(tpl, seed=5, dtype=torch.float64)
    B = 2
    x = synth_input(B * world)[rank * B:(rank + 1) * B]
    eps = synth_vec((B * world, dz), 1.1, 0.3)[rank * B:(rank + 1) * B]
    r = vo.train_step(sd, x, arch, dz, eps, None, None)
    return tpl, sd, r['grads']


def _worker(rank, world, port, ret, shared=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from preset_gen_vae_amd import optim, parallel
        from preset_gen_vae_amd.model import layer
        tpl, sd, grads = _shard_grads(rank, world)
        params = {k: torch.nn.Parameter(sd[k].float()) for k in grads}
        flat = optim.FlatParams(list(params.values()))
        sync = parallel.GradAllReduce(flat, n_buckets=3).install()
        assert sync.world_size == world and len(sync.ranges) == 3
        sync.start_step()
        late = []
        for p in flat.params:                       # gradient-ready order
            k = [kk for kk, vv in params.items() if vv is p][0]
            if shared and 'single_ch_cnn' in k:
                # a stack applied twice per forward (stacked spectrogram channels): each application announces its
                # kernel, but autograd adds the per-application gradients only AFTER the announcement - a bucket that
                # launched on the announcement count would all-reduce partial sums
                p._pgv_shared = True
                p.grad.copy_(0.25 * grads[k].float())
                layer._grad_done(p)
                layer._grad_done(p)
                late.append((p, 0.75 * grads[k].float()))
            else:
                p.grad.copy_(grads[k].float())
                layer._grad_done(p)
        if shared:
            assert not all(sync._launched)          # buckets holding shared parameters are still waiting
        for p, rest in late:
            p.grad.add_(rest)
        sync.wait()
        sync.uninstall()
        out = {k: params[k].grad.clone() for k in params}
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v.double() for k, v in grads.items()})
        if rank == 0:
            worst = 0.0
            for k in out:
                total = sum(gr[k] for gr in gathered)
                err = (out[k].double() - total).abs().max().item() / max(total.abs().max().item(), 1e-12)
                worst = max(worst, err)
            ret['worst'] = worst
            ret['launched'] = all(sync._launched)
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret['launched'] is True
    assert ret['worst'] < 1e-6


def test_shared_stack_parameters_are_reduced_after_backward_world2():
    """Parameters of a stack applied once per spectrogram channel: their buckets must not launch from the hooks."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, True), nprocs=world, join=True)
    assert ret['launched'] is True
    assert ret['worst'] < 1e-6


def test_single_process_is_a_noop():
    from preset_gen_vae_amd import optim, parallel
    p = torch.nn.Parameter(torch.ones(10))
    flat = optim.FlatParams([p])
    sync = parallel.GradAllReduce(flat, n_buckets=2)
    assert sync.world_size == 1
    sync.start_step()
    p.grad.fill_(3.0)
    sync.wait()
    assert torch.all(p.grad == 3.0)


def _worker_cuts(rank, world, port, ret):
    """The bucket-graph protocol of VAETrainStep without the graphs (they need a GPU): pass 1 = "capture" - the hooks
    report complete buckets to ``capture_cuts`` and launch NOTHING; pass 2..3 = "replays" - the gradients of a segment are
    written, then that segment's buckets are launched with ``launch_bucket``, the rest by ``wait``."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from preset_gen_vae_amd import optim, parallel
        from preset_gen_vae_amd.model import layer
        tpl, sd, grads = _shard_grads(rank, world)
        params = {k: torch.nn.Parameter(sd[k].float()) for k in grads}
        name_of = {id(v): k for k, v in params.items()}
        flat = optim.FlatParams(list(params.values()))
        sync = parallel.GradAllReduce(flat, n_buckets=3)
        segments, cur = [], []                     # [(parameters whose gradient the segment writes, buckets complete)]
        sync.capture_cuts(lambda bi: (segments.append((list(cur), [bi])), cur.clear()))
        layer.GRAD_READY_HOOK = sync._on_grad_ready
        sync.start_step()
        for p in flat.params:
            cur.append(p)
            layer._grad_done(p)
        segments.append((list(cur), []))
        layer.GRAD_READY_HOOK = None
        sync.capture_cuts(None)
        assert sync.n_collectives == 0 and [b for _, b in segments] == [[0], [1], [2], []]
        worst = 0.0
        for _ in range(2):                         # two "replayed" steps
            flat.flat_grad.zero_()
            sync.start_step()
            for ps, ready in segments:
                for p in ps:
                    p.grad.copy_(grads[name_of[id(p)]].float())
                for bi in ready:
                    sync.launch_bucket(bi)
            sync.wait()
            gathered = [None] * world
            dist.all_gather_object(gathered, {k: v.double() for k, v in grads.items()})
            for k, p in params.items():
                total = sum(gr[k] for gr in gathered)
                worst = max(worst, (p.grad.double() - total).abs().max().item() / max(total.abs().max().item(), 1e-12))
        if rank == 0:
            ret['worst'] = worst
            ret['collectives'] = sync.n_collectives
    finally:
        dist.destroy_process_group()


def test_bucket_graph_protocol_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_cuts, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret['collectives'] == 6            # 3 buckets x 2 steps, none during the capture pass
    assert ret['worst'] < 1e-6
