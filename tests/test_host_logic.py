"""CPU: host-side logic of the product package — builders / config surface, reference state-dict keys, flat parameter
storage and bucketing, mel filterbank, layer geometry.  No kernel is launched."""
import copy

import numpy as np
import pytest
import torch

from helpers import load_golden, param_shapes


def _cfg(arch, B=4, output_bn=False, dim_z=64):
    from preset_gen_vae_amd import config
    mc, tc = copy.copy(config.model), copy.copy(config.train)
    mc.encoder_architecture, mc.dim_z = arch, dim_z
    mc.input_tensor_size = (B, 1, 257, 347)
    tc.latent_flow_input_regularization = 'bn' if output_bn else 'none'
    return mc, tc


@pytest.mark.parametrize("arch,output_bn,golden", [('speccnn8l1_bn', False, 'vae8l_b2.npz'),
                                                   ('speccnn8l1_bn', True, 'vae8l_b2_outbn.npz'),
                                                   ('speccnn4l1_bn', False, 'vae4l_b2.npz')])
def test_state_dict_keys_match_reference(arch, output_bn, golden):
    """Key names and shapes equal the REFERENCE's state dict (recorded in the goldens from the real modules)."""
    from preset_gen_vae_amd.model import build
    enc, dec, ae = build.build_ae_model(*_cfg(arch, output_bn=output_bn))
    sd = ae.state_dict()
    tpl = param_shapes(arch, 64, output_bn)
    assert list(sd.keys()) == list(tpl.keys())           # same keys, same registration order
    for k, shape in tpl.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    g = load_golden(golden)
    ref_keys = {k[len('post_full/'):] for k in g.files if k.startswith('post_full/')}
    ref_keys |= {k[len('post/'):-len('/checksum')] for k in g.files if k.startswith('post/') and k.endswith('/checksum')}
    ref_keys |= {k[len('post/'):] for k in g.files if k.startswith('post/') and k.endswith('num_batches_tracked')}
    assert ref_keys == set(sd.keys())
    n_params = sum(p.numel() for p in ae.parameters())
    if arch == 'speccnn8l1_bn':
        assert n_params == 12440017 + (256 if output_bn else 0)      # SURVEY.md §2.2 [probed]


def test_fp32_products_default_is_what_bench_times_and_is_reachable_from_the_training_path():
    """the library default, ``config.train.fp32_products`` and bench.py's default are ONE setting
    (``ops.DEFAULT_FP32_PRODUCTS``); the builders and VAETrainStep can select either form."""
    import bench
    from preset_gen_vae_amd import config, ops
    from preset_gen_vae_amd.model import build
    import sys
    before = ops.fp32_products()
    try:
        ops.set_fp32_products(None)
        assert ops.fp32_products() == ops.DEFAULT_FP32_PRODUCTS == 'bf16x6'
        assert config.train.fp32_products is None                       # = "the library default"
        argv, sys.argv = sys.argv, ['bench.py']
        try:
            assert bench.parse().fp32_products is None                  # bench.py times the library default
        finally:
            sys.argv = argv
        mc, tc = _cfg('speccnn4l1_bn')
        build.build_ae_model(mc, tc)
        assert ops.fp32_products() == 'bf16x6'                          # None leaves the default alone
        tc.fp32_products = 'native'
        build.build_ae_model(mc, tc)
        assert ops.fp32_products() == 'native' and not (ops._flags() & ops.PGV_COMPUTE_F32_SPLIT)
        tc.fp32_products = 'bf16x6'
        build.build_ae_model(mc, tc)
        assert ops._flags() & ops.PGV_COMPUTE_F32_SPLIT
        with pytest.raises(ValueError):
            ops.set_fp32_products('tf32')
    finally:
        ops.set_fp32_products(before)


def test_extended_model_checkpoint_layout_matches_the_reference():
    """What the reference writes into a checkpoint (logs/logger.py:199-202: ``extended_ae_model.state_dict()`` +
    ``optimizer.state_dict()``) against tests/golden/extended_keys.npz, recorded from the reference's own ExtendedAE
    around its BasicVAE + MLPRegression: key names, registration order, shapes and dtypes of the extended model for both
    latent regularisations, and the layout of Adam's state dict over its parameters."""
    from preset_gen_vae_amd.model import build
    from preset_gen_vae_amd import optim as optim_mod
    g = load_golden('extended_keys.npz')

    class Helper:
        learnable_preset_size = 144

    for tag, output_bn in (('none', False), ('bn', True)):
        mc, tc = _cfg('speccnn8l1_bn', B=2, output_bn=output_bn)
        _, _, _, ext = build.build_extended_ae_model(mc, tc, Helper())
        sd = ext.state_dict()
        assert list(sd.keys()) == [str(k) for k in g[f'{tag}/keys']]
        assert [' '.join(str(d) for d in v.shape) for v in sd.values()] == [str(x) for x in g[f'{tag}/shapes']]
        assert [str(v.dtype) for v in sd.values()] == [str(x) for x in g[f'{tag}/dtypes']]
        assert [k for k, _ in ext.named_parameters()] == [str(k) for k in g[f'{tag}/param_names']]
    assert any(k.startswith('ae_model.encoder.') for k in sd) and any(k.startswith('reg_model.reg_model.') for k in sd)
    # optimizer.state_dict() (recorded on the 'none' variant): same top-level / group / per-parameter entries, one entry per
    # parameter in registration order
    mc, tc = _cfg('speccnn8l1_bn', B=2, output_bn=False)
    params = list(build.build_extended_ae_model(mc, tc, Helper())[3].parameters())
    assert len(params) == int(g['adam/n_params']) and list(range(len(params))) == [int(i) for i in g['adam/param_ids']]
    assert [' '.join(str(d) for d in p.shape) for p in params] == [str(x) for x in g['adam/state_shapes']]
    assert [str(k) for k in g['adam/top_keys']] == ['param_groups', 'state']
    assert {'exp_avg', 'exp_avg_sq', 'step'} <= {str(k) for k in g['adam/state_entry_keys']}
    assert {'lr', 'betas', 'eps', 'weight_decay', 'amsgrad', 'params'} <= {str(k) for k in g['adam/group_keys']}
    assert float(g['adam/step_after_one']) == 1.0
    assert hasattr(optim_mod.FusedAdam, 'state_dict') and hasattr(optim_mod.FusedAdam, 'load_state_dict')


def test_builder_surface_and_errors():
    from preset_gen_vae_amd.model import VAE, build, extendedAE, regression

    class Helper:
        learnable_preset_size = 144

    mc, tc = _cfg('speccnn8l1_bn')
    enc, dec, ae, ext = build.build_extended_ae_model(mc, tc, Helper())
    assert isinstance(ae, VAE.BasicVAE) and isinstance(ext, extendedAE.ExtendedAE)
    assert isinstance(ext.reg_model, regression.MLPRegression)
    assert ext.is_flow_based_latent_space is False and ext.is_flow_based_regression is False
    assert ext.ae_model is ae and ae.encoder is enc and ae.decoder is dec and ae.dim_z == 64
    assert ae.is_profiled is False
    assert enc.cnn_out_size == torch.Size((1, 2048, 3, 4))
    assert [k for k in ext.reg_model.state_dict()][:3] == ['reg_model.fc1.weight', 'reg_model.fc1.bias',
                                                           'reg_model.bn1.weight']
    mc.latent_flow_arch = 'realnvp_6l300'
    with pytest.raises(NotImplementedError):
        build.build_ae_model(mc, tc)
    mc, tc = _cfg('wavenet_baseline')
    with pytest.raises(NotImplementedError):
        build.build_ae_model(mc, tc)
    mc, tc = _cfg('speccnn8l1_bn')
    mc.params_regression_architecture = 'flow_realnvp_6l300'
    with pytest.raises(NotImplementedError):
        build.build_extended_ae_model(mc, tc, Helper())
    mc, tc = _cfg('speccnn8l1_bn')
    mc.stack_specs_deepest_features_mix = True
    enc, dec, ae = build.build_ae_model(mc, tc)
    assert 'encoder.single_ch_cnn.enc_nn.4x4conv.enc7conv.weight' in ae.state_dict()
    assert 'encoder.features_mixer_cnn.enc8conv.weight' in ae.state_dict()


def test_layer_geometry_traces_reference_shapes():
    """257x347 -> 129x174 -> ... -> 3x4 and back with the per-axis output_padding."""
    from preset_gen_vae_amd.model import build
    enc, dec, ae = build.build_ae_model(*_cfg('speccnn8l1_bn'))
    H, W = 257, 347
    seen = []
    for blk in enc._all_blocks():
        g = blk.geom(H, W)
        H, W = g.Hs, g.Ws
        seen.append((blk.c_out, H, W))
    assert seen == [(8, 129, 174), (16, 65, 88), (32, 33, 45), (64, 17, 23), (128, 9, 12), (256, 5, 7), (512, 3, 4),
                    (2048, 3, 4)]
    seen = []
    for blk in dec._all_blocks():
        g = blk.geom(H, W)
        H, W = g.Hb, g.Wb
        seen.append((blk.c_out, H, W))
    assert seen == [(512, 3, 4), (256, 5, 7), (128, 9, 12), (64, 17, 23), (32, 33, 45), (16, 65, 88), (8, 129, 174),
                    (1, 257, 347)]


def test_flat_params_and_buckets():
    from preset_gen_vae_amd import optim
    from preset_gen_vae_amd.model import build
    enc, dec, ae = build.build_ae_model(*_cfg('speccnn4l1_bn'))
    before = {k: v.clone() for k, v in ae.state_dict().items()}
    flat = optim.FlatParams(ae.parameters())
    for k, v in ae.state_dict().items():
        assert torch.equal(v, before[k]), k                       # values preserved, keys unchanged
    n = sum(p.numel() for p in ae.parameters())
    assert n <= flat.numel < n + 64 * len(flat.params)
    assert all(o % 64 == 0 for o in flat.offsets)
    for p, o in zip(flat.params, flat.offsets):
        assert o % 4 == 0
        assert p.data.data_ptr() == flat.flat_param.data_ptr() + 4 * o
        assert p.grad is not None and p.grad.data_ptr() == flat.flat_grad.data_ptr() + 4 * o
    # gradient-ready order: the decoder's output layer comes first, the encoder's first conv last
    names = {id(p): k for k, p in ae.named_parameters()}
    assert names[id(flat.params[0])].startswith('decoder.single_ch_cnn.dec_nn.3')
    assert names[id(flat.params[-1])] == 'encoder.single_ch_cnn.enc_nn.0.enc1conv.weight'
    ranges = flat.bucket_ranges(4)
    assert ranges[0][0] == 0 and ranges[-1][1] == flat.numel
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert sum(len(flat.params_in_range(lo, hi)) for lo, hi in ranges) == len(flat.params)
    flat.flat_param.mul_(2.0)
    assert torch.equal(ae.state_dict()['decoder.mlp.0.weight'], 2 * before['decoder.mlp.0.weight'])


def test_product_mel_basis_known_answers():
    from preset_gen_vae_amd.utils.audio import dense_to_csr, slaney_mel_basis
    fb = slaney_mel_basis(22050, 1024, 257)
    assert fb.shape == (257, 513) and int((fb != 0).sum()) == 1016
    rp, col, val = dense_to_csr(fb)
    assert rp[0] == 0 and rp[-1] == 1016 and np.all(np.diff(rp) >= 1) and np.diff(rp).max() == 14
    dense = np.zeros_like(fb)
    for r in range(257):
        dense[r, col[rp[r]:rp[r + 1]]] = val[rp[r]:rp[r + 1]]
    assert np.array_equal(dense, fb)
    np.testing.assert_allclose(fb[128, 91:94], [0.116312, 0.938339, 0.249680], atol=2e-6)


def test_config_surface():
    from preset_gen_vae_amd import config
    config.update_dynamic_config_params()
    assert config.model.input_tensor_size == (config.train.minibatch_size, 1, 257, 347)
    assert config.model.concat_midi_to_z is False and config.model.stft_args == (1024, 256)
    assert config.train.adam_betas == (0.9, 0.999) and config.train.weight_decay == 1e-4


def test_fused_adam_state_dict_is_torch_adam_layout():
    """optimizer_state_dict of the reference checkpoints (logs/logger.py:199-202, train.py:177-179): FusedAdam emits and
    accepts torch.optim.Adam's per-parameter layout in model.parameters() order (the step itself needs the GPU:
    tests/test_gpu_vae.py::test_optimizer_state_resume)."""
    from preset_gen_vae_amd import optim
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 2))]
    flat = optim.FlatParams(ps)
    opt = optim.FusedAdam(flat, lr=1e-3, weight_decay=1e-4)
    sd0 = opt.state_dict()
    assert sd0['state'] == {} and sd0['param_groups'][0]['params'] == [0, 1, 2] and opt.step_count() == 0
    # a torch Adam that has taken 3 steps on the same parameters -> into FusedAdam -> back out -> into torch Adam
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ta = torch.optim.Adam(ref, lr=5e-4, weight_decay=1e-4, betas=(0.8, 0.99))
    for _ in range(3):
        for r in ref:
            r.grad = torch.randn_like(r)
        ta.step()
    opt.load_state_dict(ta.state_dict())
    assert opt.step_count() == 3 and opt.param_groups[0]['lr'] == 5e-4 and opt.param_groups[0]['betas'] == (0.8, 0.99)
    assert abs(opt.hyper[0].item() - 5e-4) < 1e-10
    assert abs(opt.pows[0].item() - 0.8 ** 3) < 1e-15 and abs(opt.pows[1].item() - 0.99 ** 3) < 1e-15
    sd = opt.state_dict()
    tsd = ta.state_dict()
    for i, p in enumerate(ps):
        assert torch.equal(sd['state'][i]['exp_avg'], tsd['state'][i]['exp_avg'])
        assert torch.equal(sd['state'][i]['exp_avg_sq'], tsd['state'][i]['exp_avg_sq'])
        assert float(sd['state'][i]['step']) == 3.0 and sd['state'][i]['exp_avg'].shape == p.shape
        o = [off for q, off in zip(flat.params, flat.offsets) if q is p][0]   # moments sit at the flat offset
        assert torch.equal(opt.exp_avg[o:o + p.numel()].view(p.shape), tsd['state'][i]['exp_avg'])
    tb = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1.0)
    tb.load_state_dict(sd)
    for r_a, r_b in zip(ref, tb.param_groups[0]['params']):
        r_b.data.copy_(r_a.data)
        r_a.grad = torch.randn_like(r_a)
        r_b.grad = r_a.grad.clone()
    ta.step(), tb.step()
    for r_a, r_b in zip(ref, tb.param_groups[0]['params']):
        assert torch.equal(r_a, r_b)                                # the reloaded optimizer continues identically
    with pytest.raises(ValueError):
        opt.load_state_dict({'state': {}, 'param_groups': [{'lr': 1e-3, 'params': [0, 1]}]})


def test_check_configs_on_resume_from_checkpoint():
    import copy
    from preset_gen_vae_amd import config
    from preset_gen_vae_amd.model import build
    config.update_dynamic_config_params()
    mc, tc = copy.copy(config.model), copy.copy(config.train)
    saved = {'model': {k: (list(v) if isinstance(v, tuple) else v) for k, v in mc.__dict__.items()},
             'train': dict(tc.__dict__)}
    build.check_configs_on_resume_from_checkpoint(mc, tc, saved)        # json lists compare equal to tuples
    mc2 = copy.copy(mc)
    mc2.dim_z = mc.dim_z + 1
    with pytest.raises(ValueError, match="Model attribute 'dim_z'"):
        build.check_configs_on_resume_from_checkpoint(mc2, tc, saved)
    tc2 = copy.copy(tc)
    tc2.minibatch_size = 3
    with pytest.raises(ValueError, match="Train attribute 'minibatch_size'"):
        build.check_configs_on_resume_from_checkpoint(mc, tc2, saved)


def test_preset_activation_softmax_branch_matches_reference():
    """regression.py:35-51 with cat_softmax_activation=True (golden from the reference class, tests/golden/
    params_loss.npz 'act_softmax')."""
    from helpers import MiniPresetIndexesHelper, load_golden
    from preset_gen_vae_amd.model import regression
    g = load_golden('params_loss.npz')
    x = torch.tensor(g['act_softmax/in'], requires_grad=True)
    x_before = x.detach().clone()
    act = regression.PresetActivation(MiniPresetIndexesHelper(), cat_softmax_activation=True)
    y = act(x)
    assert torch.equal(x.detach(), x_before)                            # no in-place write (the reference mutates)
    assert (y.detach() - torch.tensor(g['act_softmax/out'])).abs().max().item() < 1e-12
    (y * torch.tensor(g['act_softmax/gy'])).sum().backward()
    assert (x.grad - torch.tensor(g['act_softmax/gx'])).abs().max().item() < 1e-12


def test_bench_starts_its_own_ranks_for_more_than_one_gpu():
    """``python bench.py --gpus 2`` with no launcher above it (the form the round driver uses) must start the two ranks
    itself - as child processes of torch.distributed.run, never by replacing a process - instead of exiting with an
    instruction.  Without a GPU the children get as far as the device check and fail THERE; the parent hands their
    status back."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['WORLD_SIZE'] = ''                    # an empty value counts as "no launcher" too
    env['CUDA_VISIBLE_DEVICES'] = env['HIP_VISIBLE_DEVICES'] = ''
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count('bench.py needs a ROCm GPU') >= 2, r.stderr[-2000:]
    assert 'launch with torch.distributed.run' not in r.stderr
