"""CPU tier 1: the oracle (oracle/) reproduces the golden vectors generated from the real reference
(tests/golden/*.npz, made by tests/golden/make_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from helpers import (check_big, load_golden, param_shapes, rel_l2, stack_channels, synth_input, template_from_meta,
                     unpack_mask)
from oracle import audio_oracle as ao
from oracle import vae_oracle as vo


def _template_from_golden(g):
    tpl = {}
    for k in g.files:
        if k.startswith('post_full/'):
            tpl[k[len('post_full/'):]] = tuple(g[k].shape)
        elif k.startswith('post/') and k.endswith('/checksum'):
            tpl[k[len('post/'):-len('/checksum')]] = None
        elif k.startswith('post/') and k.endswith('num_batches_tracked'):
            tpl[k[len('post/'):]] = ()
    return tpl


def reference_template(g):
    """Template in the REFERENCE's registration order (the closed-form weights depend on the key index)."""
    arch, dim_z, output_bn = str(g['meta/arch']), int(g['meta/dim_z']), bool(g['meta/output_bn'])
    tpl = param_shapes(arch, dim_z, output_bn)
    want = set(_template_from_golden(g).keys())
    assert set(tpl.keys()) == want, (sorted(set(tpl) ^ want))
    return tpl


@pytest.mark.parametrize("name", ["vae8l_b2.npz", "vae8l_b2_outbn.npz", "vae4l_b2.npz", "vae8l_b2_c2.npz",
                                  "vae8l_b2_c2_mix.npz", "vae8l_b2_big.npz",
                                  "vae8l_b16.npz", "vae4l_b16.npz", "vae8l_b16_outbn.npz", "vae4l_b16_outbn.npz"])
def test_train_step_matches_reference(name):
    g = load_golden(name)
    arch, dim_z, B = str(g['meta/arch']), int(g['meta/dim_z']), int(g['meta/B'])
    n_ch = int(g['meta/n_ch']) if 'meta/n_ch' in g.files else 1
    variant = n_ch > 1 or ('meta/force_bigger' in g.files and bool(g['meta/force_bigger']))
    # stacked spectrograms and the reference's other encoder variants (encoder.py:54-70: deepest 1x1 mixer,
    # force_bigger_network): the template travels with the golden (mixer shapes depend on the variant)
    tpl = template_from_meta(g) if variant else reference_template(g)
    sd = vo.closed_form_state_dict(tpl, seed=int(g['meta/seed']), dtype=torch.float64)
    x = stack_channels(synth_input(B), n_ch)
    eps = torch.tensor(g['in/eps'])
    enc_mask, dec_mask = unpack_mask(g, 'enc'), unpack_mask(g, 'dec')
    # eval-mode forward
    zml, _, _, _, x_out = vo.vae_forward(sd, x, arch, dim_z, training=False)
    assert rel_l2(zml, torch.tensor(g['eval/z_mu_logvar'])) < 1e-10
    check_big('eval x_out', x_out, g, 'eval/x_out', 1e-9)
    # one train step
    r = vo.train_step(sd, x, arch, dim_z, eps, enc_mask, dec_mask, beta=float(g['meta/beta']),
                      lr=float(g['meta/lr']), weight_decay=float(g['meta/weight_decay']))
    assert rel_l2(r['z_mu_logvar'], torch.tensor(g['train/z_mu_logvar'])) < 1e-10
    assert rel_l2(r['z'], torch.tensor(g['train/z'])) < 1e-10
    for key in ('recons', 'latent', 'total'):
        assert abs(r[key].item() - float(g['train/' + key])) <= 1e-10 * abs(float(g['train/' + key]))
    zml_t = r['z_mu_logvar']
    assert abs(vo.gaussian_dkl(zml_t[:, 0], zml_t[:, 1], normalize=False).item()
               - float(g['train/latent_unnormalized'])) < 1e-10
    assert abs(vo.l2_loss(r['x_out'], x).item() - float(g['train/l2loss'])) < 1e-9 * float(g['train/l2loss'])
    check_big('x_out', r['x_out'], g, 'train/x_out', 1e-9)
    for k, gr in r['grads'].items():
        check_big('grad ' + k, gr, g, 'grad/' + k, 1e-8)
    for k, v in r['new_sd'].items():
        if 'post_full/' + k in g.files:
            ref = torch.tensor(g['post_full/' + k])
            assert (v.double() - ref).abs().max().item() <= 1e-10 * max(1.0, ref.abs().max().item()), k
        elif 'post/' + k + '/checksum' in g.files:
            check_big('post ' + k, v, g, 'post/' + k, 1e-9)


def regstep_inputs(g):
    """Inputs of the regstep golden (VAE + preset-regression network, train.py:203-248) as the oracle takes them."""
    from helpers import template_from_meta
    arch, dim_z, B = str(g['meta/arch']), int(g['meta/dim_z']), int(g['meta/B'])
    sd = vo.closed_form_state_dict(param_shapes(arch, dim_z, False), seed=int(g['meta/seed']), dtype=torch.float64)
    rtpl = {str(k): (tuple(int(v) for v in str(sh).split()) if str(sh) else ())
            for k, sh in zip(g['meta/reg_keys'], g['meta/reg_shapes'])}
    rsd = vo.closed_form_state_dict(rtpl, seed=int(g['meta/reg_seed']), dtype=torch.float64)
    rmasks = [torch.tensor(np.unpackbits(g[f'in/reg_mask{i}_bits'])[:B * 1024].reshape(B, 1024), dtype=torch.float64)
              / 0.6 for i in range(2)]
    return dict(arch=arch, dim_z=dim_z, B=B, sd=sd, rsd=rsd, rtpl=rtpl, x=synth_input(B), eps=torch.tensor(g['in/eps']),
                enc_mask=unpack_mask(g, 'enc'), dec_mask=unpack_mask(g, 'dec'), v_in=torch.tensor(g['in/v_in']),
                rmasks=rmasks)


def test_train_step_with_regression_matches_reference():
    """train.py:203-248 with the preset-regression network in the step (v_out = reg_model(z_K), cont_loss, one backward
    through both networks, Adam over the extended model): golden from the reference's own modules."""
    g = load_golden('regstep_4l_b4.npz')
    i = regstep_inputs(g)
    r = vo.train_step(i['sd'], i['x'], i['arch'], i['dim_z'], i['eps'], i['enc_mask'], i['dec_mask'],
                      beta=float(g['meta/beta']), lr=float(g['meta/lr']), weight_decay=float(g['meta/weight_decay']),
                      reg=dict(sd=i['rsd'], v_in=i['v_in'], masks=i['rmasks']))
    assert rel_l2(r['z_mu_logvar'], torch.tensor(g['train/z_mu_logvar'])) < 1e-10
    assert rel_l2(r['v_out'], torch.tensor(g['train/v_out'])) < 1e-10
    for key in ('recons', 'latent', 'controls', 'total'):
        assert abs(r[key].item() - float(g['train/' + key])) <= 1e-10 * abs(float(g['train/' + key])), key
    for k, gr in r['grads'].items():
        check_big('grad ' + k, gr, g, ('grad_reg/' + k[4:]) if k.startswith('reg.') else ('grad/' + k), 1e-8)
    n = 0
    for k, v in r['new_sd'].items():
        if 'post_full/' + k in g.files:
            ref = torch.tensor(g['post_full/' + k])
            assert (v.double() - ref).abs().max().item() <= 1e-10 * max(1.0, ref.abs().max().item()), k
            n += 1
        elif 'post/' + k + '/checksum' in g.files:
            check_big('post ' + k, v, g, 'post/' + k, 1e-9)
            n += 1
    assert n == len([k for k in r['new_sd'] if not k.endswith('num_batches_tracked')])


def test_train_step_with_synth_params_loss_matches_reference():
    """The same step with the criteria train.py really uses: SynthParamsLoss incl. its categorical branch and the
    useless-parameter exclusion as controls loss (train.py:111-116, 238-239) and the two per-minibatch monitoring metrics
    (train.py:229-233); golden regstep_4l_b4_cat.npz from the reference's own classes."""
    from helpers import MiniPresetIndexesHelper
    from oracle import params_oracle as po
    g = load_golden('regstep_4l_b4_cat.npz')
    i = regstep_inputs(g)
    helper = MiniPresetIndexesHelper()
    crit = lambda v_out, v_in: po.synth_params_loss(v_out, v_in, helper, True, cat_bce=False, cat_softmax=True,  # noqa: E731
                                                    cat_softmax_t=0.2)
    r = vo.train_step(i['sd'], i['x'], i['arch'], i['dim_z'], i['eps'], i['enc_mask'], i['dec_mask'],
                      beta=float(g['meta/beta']), lr=float(g['meta/lr']), weight_decay=float(g['meta/weight_decay']),
                      reg=dict(sd=i['rsd'], v_in=i['v_in'], masks=i['rmasks'], criterion=crit))
    assert rel_l2(r['v_out'], torch.tensor(g['train/v_out'])) < 1e-10
    for key in ('recons', 'latent', 'controls', 'total'):
        assert abs(r[key].item() - float(g['train/' + key])) <= 1e-10 * abs(float(g['train/' + key])), key
    q = po.quantized_numerical_params_loss(r['v_out'], i['v_in'], helper)
    a = np.mean(list(po.categorical_params_accuracy(r['v_out'], i['v_in'], helper).values()))   # reduce=True, loss.py:313
    # (the reference collects the quantised columns in float32 tensors, loss.py:222-224: its metric is a float32 number)
    assert abs(float(q) - float(g['train/qloss'])) <= 1e-6 * float(q) and abs(float(a) - float(g['train/accuracy'])) <= 1e-9
    n = 0
    for k, gr in r['grads'].items():
        key = ('grad_reg/' + k[4:]) if k.startswith('reg.') else ('grad/' + k)
        if key + '/checksum' in g.files:
            check_big('grad ' + k, gr, g, key, 1e-8)
            n += 1
    assert n >= 10


def test_layer_blocks_match_reference():
    g = load_golden('layers_small.npz')
    names = sorted({k.split('/')[0] for k in g.files})
    assert len(names) == 16
    for name in names:
        kind = str(g[name + '/kind'])
        ci, co, k, s, p, oph, opw, bn = (int(v) for v in g[name + '/cfg'])
        sd = {kk[len(name + '/sd_in/'):]: torch.tensor(g[kk]) for kk in g.files if kk.startswith(name + '/sd_in/')}
        x = torch.tensor(g[name + '/x']).requires_grad_(True)
        params = {kk: v.clone().requires_grad_(True) for kk, v in sd.items() if vo.is_parameter_key(kk)}
        full = dict(sd)
        full.update(params)
        nb = {}
        if kind == 'conv':
            y = vo.conv_block(x, full, (name, ci, co, k, s, p, bool(bn)), '', True, nb)
        elif kind == 'tconv':
            y = vo.tconv_block(x, full, (name, ci, co, k, s, p, (oph, opw), bool(bn)), '', True, nb)
        else:
            y = torch.nn.functional.hardtanh(torch.nn.functional.conv_transpose2d(
                x, full['0.weight'], full['0.bias'], stride=s, padding=p))
        assert rel_l2(y, torch.tensor(g[name + '/y'])) < 1e-11, name
        y.backward(torch.tensor(g[name + '/gy']))
        assert rel_l2(x.grad, torch.tensor(g[name + '/gx'])) < 1e-10, name
        for kk, v in params.items():
            assert rel_l2(v.grad, torch.tensor(g[name + '/grad/' + kk])) < 1e-10, (name, kk)
        for suffix, val in nb.items():
            ref = [g[kk] for kk in g.files if kk.startswith(name + '/sd_out/') and kk.endswith(suffix)]
            assert len(ref) == 1 and rel_l2(val, torch.tensor(ref[0])) < 1e-11, (name, suffix)


def test_stft_matches_reference_spectrogram():
    g = load_golden('stft.npz')
    assert np.abs(ao.hann_symmetric(1024) - g['window']).max() < 1e-6     # reference window is float32
    assert abs(ao.norm_factor(ao.hann_symmetric(1024)) - float(g['norm_factor'])) < 1e-3
    assert abs(float(g['norm_factor']) - 511.5) < 1e-3
    for idx in range(3):
        wav = ao.synth_fm_wave(idx=idx)
        frames = g[f'wave{idx}/frames']
        mag = ao.spectrogram_mag(wav)
        assert mag.shape == (513, 347)
        ref_mag = g[f'wave{idx}/mag']            # reference: float32 torch.stft
        # fp32 STFT noise floor: absolute error relative to the frame's largest bin
        err = np.abs(mag[:, frames] - ref_mag).max(axis=0) / np.maximum(ref_mag.max(axis=0), 1e-12)
        assert err.max() < 5e-6, err
        db = ao.spectrogram_db(wav)
        ref_db = g[f'wave{idx}/db']
        strong = ref_mag > 1e-4                   # above -80 dB the fp32 reference is accurate to << 0.01 dB
        assert np.abs(db[:, frames] - ref_db)[strong].max() < 2e-2
        cs = g[f'wave{idx}/db_checksum']
        assert abs(db.max() - cs[2]) < 1e-3 and abs(db.min() - cs[3]) < 1e-3
    short = ao.spectrogram_db(ao.synth_fm_wave(n=700, idx=5))
    assert short.shape == g['short/db'].shape == (513, 3)
    m = g['short/db'] > -80
    assert np.abs(short - g['short/db'])[m].max() < 2e-2


def test_mel_filterbank_known_answers():
    """SURVEY.md §2.2 known answers of the Slaney basis (librosa itself is not installed: parity unpinned)."""
    fb = ao.mel_filterbank()
    assert fb.shape == (257, 513) and fb.dtype == np.float32
    assert int((fb != 0).sum()) == 1016
    assert (fb != 0).sum(axis=1).max() == 14 and (fb != 0).sum(axis=1).min() >= 1
    assert fb[:, 0].max() == 0 and fb[:, 512].max() == 0
    np.testing.assert_allclose(fb[0, 1], 0.330345, atol=2e-6)
    np.testing.assert_allclose(fb[1, 1], 0.669655, atol=2e-6)
    np.testing.assert_allclose(fb[2, 2], 0.660689, atol=2e-6)
    np.testing.assert_allclose(fb[128, 91:94], [0.116312, 0.938339, 0.249680], atol=2e-6)
    np.testing.assert_allclose(fb[256].sum(), 6.71158, atol=2e-5)
    np.testing.assert_allclose(fb.sum(), 508.1044, atol=2e-3)
    assert fb.sum(axis=0).max() <= 1.0 + 1e-6


def _golden_mel_basis():
    g = load_golden('mel_basis.npz')
    fb = np.zeros(tuple(g['shape']), dtype=np.float64)
    rp = g['row_ptr']
    for r in range(fb.shape[0]):
        fb[r, g['col'][rp[r]:rp[r + 1]]] = g['val'][rp[r]:rp[r + 1]]
    return g, fb


def test_mel_filterbank_matches_independent_implementation():
    """The pin of SURVEY 8 a12: the oracle's Slaney basis AND the product's host-side basis (utils/audio.slaney_mel_basis,
    the CSR the HIP kernel reads) against ``tests/golden/mel_basis.npz`` - the basis of
    transformers.audio_utils.mel_filter_bank(norm=None, mel_scale="slaney"), an independent implementation documented to
    reproduce librosa.filters.mel(htk=False, norm=None), generated in the build container by make_goldens.py (librosa 0.8
    itself, the reference's dependency, is not obtainable here)."""
    import importlib
    g, ref = _golden_mel_basis()
    assert ref.shape == (257, 513) and len(g['val']) == 1016
    fb = ao.mel_filterbank()
    assert np.array_equal(fb != 0, ref != 0)
    assert np.abs(fb.astype(np.float64) - ref).max() < 1e-7
    audio = importlib.import_module('preset_gen_vae_amd.utils.audio')
    prod = audio.slaney_mel_basis(22050, 1024, 257)
    assert prod.shape == (257, 513) and np.array_equal(prod != 0, ref != 0)
    assert np.abs(prod.astype(np.float64) - ref).max() < 1e-7
    np.testing.assert_allclose(prod.astype(np.float64).sum(axis=1), g['row_sums'], atol=2e-6)


def test_minmax_normalisation():
    s = np.array([-120.0, -60.0, 0.0])
    np.testing.assert_allclose(ao.minmax_normalize(s, -120.0, 0.0), [-1.0, 0.0, 1.0])


def regression_template():
    tpl = {}
    dims = [64, 1024, 1024, 1024, 144]
    for l in range(1, 5):
        tpl[f'reg_model.fc{l}.weight'] = (dims[l], dims[l - 1])
        tpl[f'reg_model.fc{l}.bias'] = (dims[l],)
        if l < 3:
            for n, shp in (('weight', (1024,)), ('bias', (1024,)), ('running_mean', (1024,)), ('running_var', (1024,)),
                           ('num_batches_tracked', ())):
                tpl[f'reg_model.bn{l}.{n}'] = shp
    return tpl


def test_regression_matches_reference():
    """a14: MLPRegression + numeric SynthParamsLoss of the reference (tests/golden/regression_b6.npz)."""
    g = load_golden('regression_b6.npz')
    sd = vo.closed_form_state_dict(regression_template(), seed=4321, dtype=torch.float64)
    z = torch.tensor(g['in/z']).requires_grad_(True)
    v_in = torch.tensor(g['in/v_in'])
    masks = [torch.tensor(np.unpackbits(g[f'in/mask{i}_bits'])[:6 * 1024].reshape(6, 1024), dtype=torch.float64) / 0.6
             for i in range(2)]
    nb = {}
    v_out = vo.mlp_regression_forward(sd, z, True, masks, nb)
    loss = vo.numeric_params_loss(v_out, v_in)
    loss.backward()
    assert rel_l2(v_out, torch.tensor(g['out/v_out'])) < 1e-10
    assert abs(loss.item() - float(g['out/mse'])) < 1e-12
    assert rel_l2(z.grad, torch.tensor(g['out/g_z'])) < 1e-10
    for k, v in nb.items():
        assert rel_l2(v, torch.tensor(g['post_full/' + k])) < 1e-10, k


def test_dataset_seam_matches_reference():
    """item tuples / normalisation / denormalisation of the reference's PresetDataset (fixture subclass,
    linear spectrogram) reproduced by oracle/data_oracle.py."""
    from oracle import data_oracle as do
    g = load_golden('dataset_seam.npz')
    waves, params, uids = g['in/waves'], g['in/params'], g['in/uids']
    notes = [tuple(int(v) for v in n) for n in g['in/midi_notes']]
    stats = {k: float(g['in/stats_' + k]) for k in ('min', 'max', 'mean', 'std')}
    for tag, stacked, mode in (('flat_minmax', False, 'min_max'), ('stacked_minmax', True, 'min_max'),
                               ('flat_meanstd', False, 'mean_std'), ('flat_none', False, None)):
        n = do.dataset_len(len(uids), len(notes), stacked)
        assert n == int(g[f'{tag}/len'])
        for i in range(n):
            spec, par, info, labels = do.get_item(i, waves, params, uids, notes, stats, mode, stacked,
                                                  do.linear_spectrogram_db)
            ref = g[f'{tag}/{i}/spec']
            assert spec.shape == ref.shape
            # the reference evaluates its STFT in float32 (utils/audio.py:36): above -80 dB it is accurate to << 0.01 dB
            # (same criterion as test_stft_matches_reference_spectrogram), below that its own noise floor dominates
            db_ref = do.denormalize_spectrogram(ref.astype(np.float64), stats, mode)
            strong = db_ref > -80.0
            db_err = np.abs(do.denormalize_spectrogram(spec, stats, mode) - db_ref)
            assert strong.mean() > 0.02 and db_err[strong].max() < 2e-2 and db_err.max() < 3.0
            assert np.array_equal(par, g[f'{tag}/{i}/params'])
            assert np.array_equal(info, g[f'{tag}/{i}/info']) and info.dtype == np.int32
            assert np.array_equal(labels, g[f'{tag}/{i}/labels'])
        if mode is not None:
            den = do.denormalize_spectrogram(g[f'{tag}/0/spec'], stats, mode)
            assert np.abs(den - g[f'{tag}/denorm0']).max() < 1e-4      # float32 arithmetic in the reference
            assert np.abs(do.normalize_spectrogram(den, stats, mode) - g[f'{tag}/0/spec']).max() < 1e-5


def test_gaussian_log_probabilities_match_reference():
    """utils/probability.py:13-29 + the ELBO combination of FlowVAE.latent_loss (VAE.py:183-193)."""
    g = load_golden('probability.npz')
    mu, lv, z0, zk, ladj = (torch.tensor(g[k]) for k in ('mu', 'logvar', 'z0', 'zk', 'ladj'))
    assert rel_l2(vo.gaussian_log_probability(z0, mu, lv), torch.tensor(g['log_q'])) < 1e-12
    assert rel_l2(vo.standard_gaussian_log_probability(zk), torch.tensor(g['log_p'])) < 1e-12
    zml = torch.stack([mu, lv], dim=1)
    assert abs(vo.flow_latent_loss(zml, z0, zk, ladj).item() - float(g['loss'])) < 1e-12 * abs(float(g['loss']))
    assert abs(vo.flow_latent_loss(zml, z0, zk, ladj, normalize=True).item() - float(g['loss_normalized'])) < 1e-12


PARAMS_LOSS_VARIANTS = {'cce_softmax': dict(cat_bce=False, cat_softmax=True, cat_softmax_t=0.2),
                        'cce_probs': dict(cat_bce=False, cat_softmax=False), 'bce': dict(cat_bce=True, cat_softmax=False)}


def test_params_losses_match_reference():
    """SynthParamsLoss (all categorical variants, with and without the useless-parameter exclusion),
    QuantizedNumericalParamsLoss, CategoricalParamsAccuracy and the Dexed useless-parameter rule."""
    from helpers import MiniPresetIndexesHelper
    from oracle import params_oracle as po
    g = load_golden('params_loss.npz')
    helper = MiniPresetIndexesHelper()
    u_in, raw = torch.tensor(g['in/u_in']), torch.tensor(g['in/u_out'])
    for norm in (True, False):
        for useless in (True, False):
            for name, kw in PARAMS_LOSS_VARIANTS.items():
                tag = f'{name}/norm{int(norm)}/useless{int(useless)}'
                u_out = raw.clone().requires_grad_(True)
                loss = po.synth_params_loss(u_out, u_in, helper, norm, prevent_useless_params_loss=useless, **kw)
                loss.backward()
                assert abs(loss.item() - float(g[tag + '/loss'])) <= 1e-12 * abs(float(g[tag + '/loss'])), tag
                assert rel_l2(u_out.grad, torch.tensor(g[tag + '/grad'])) < 1e-12, tag
    # (the reference gathers the columns into float32 buffers, loss.py:218-219: float32-level agreement)
    assert abs(po.quantized_numerical_params_loss(raw, u_in, helper).item() - float(g['quantized/mse'])) < 1e-6
    assert abs(po.quantized_numerical_params_loss(raw, u_in, helper, [1, 5]).item()
               - float(g['quantized/limited'])) < 1e-6
    acc = po.categorical_params_accuracy(raw, u_in, helper, percentage_output=False)
    assert list(acc.keys()) == g['accuracy/keys'].tolist()
    assert np.allclose(list(acc.values()), g['accuracy/values'])
    acc100 = po.categorical_params_accuracy(raw, u_in, helper)
    assert abs(np.mean(list(acc100.values())) - float(g['accuracy/mean'])) < 1e-12
    f2l = po.decode_full_to_learnable(g['dexed/full_to_learnable'])
    nums, cats = po.dexed_useless_learned_params_indexes(f2l, g['dexed/preset'])
    assert nums == g['dexed/useless_num'].tolist() and cats == g['dexed/useless_cat'].tolist()
    assert len(nums) > 0 and len(cats) > 0


def test_stft_complex_linear_and_dynamic_range_match_reference():
    """get_stft (complex, un-normalised), Spectrogram(log_scale=False) and the dynamic-range log scale
    (utils/audio.py:33-50, :63-69) of the oracle against the reference's own outputs."""
    g = load_golden('stft.npz')
    for idx in range(3):
        frames = g[f'wave{idx}/frames']
        z = ao.stft_complex(ao.synth_fm_wave(idx=idx))[:, frames]
        ref = g[f'wave{idx}/stft_re'] + 1j * g[f'wave{idx}/stft_im']      # reference: float32 torch.stft
        assert np.abs(z - ref).max() <= 2e-5 * np.abs(ref).max()
    mag = ao.spectrogram_mag(ao.synth_fm_wave(idx=1))
    frames = g['linear/frames']
    assert np.abs(mag[:, frames] - g['linear/mag']).max() <= 2e-5 * g['linear/mag'].max()
    dyn = ao.linear_to_log_with_dynamic_range(mag, -120.0, 60.0)[:, frames]
    assert np.abs(dyn - g['dynrange/db']).max() < 2e-2     # float32 reference near the -60 dB clamp level
    assert dyn.min() >= dyn.max() - 60.0 - 1e-9
