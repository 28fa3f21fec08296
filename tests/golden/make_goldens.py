#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REAL reference (gwendal-lv/preset-gen-vae, mounted
read-only at /root/reference) in the build container.  Runs only here (the reference does not exist on the GPU box);
the resulting small .npz fixtures are committed, the reference's sources are not.

Recipe: stub the absent third-party imports (nflows, librosa, soundfile), import
model/{layer,encoder,decoder,VAE,loss}.py and utils/audio.py, load closed-form weights (oracle.vae_oracle.
closed_form_state_dict — a formula, not an RNG stream), inject eps (patched ``Normal``) and the two Dropout masks
(swapped mask modules), run in float64, record outputs / losses / gradients / post-Adam parameters.

    python tests/golden/make_goldens.py
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

for name in ['nflows', 'nflows.flows', 'nflows.flows.realnvp', 'nflows.flows.base', 'nflows.transforms',
             'nflows.transforms.base', 'nflows.transforms.autoregressive', 'nflows.transforms.permutations',
             'nflows.transforms.coupling', 'nflows.transforms.normalization', 'nflows.nn', 'nflows.nn.nets',
             'nflows.distributions', 'nflows.distributions.normal', 'librosa', 'librosa.display', 'soundfile']:
    sys.modules[name] = mock.MagicMock()

from model import VAE as ref_VAE  # noqa: E402
from model import decoder as ref_decoder  # noqa: E402
from model import encoder as ref_encoder  # noqa: E402
from model import layer as ref_layer  # noqa: E402
from model import loss as ref_loss  # noqa: E402
from model import regression as ref_regression  # noqa: E402
from utils import audio as ref_audio  # noqa: E402

from oracle import vae_oracle as vo  # noqa: E402  (only for the closed-form weight/input formulas)

torch.set_num_threads(8)


def synth_input(B, H=257, W=347, dtype=torch.float64):
    """Spectrogram-like deterministic input in [-1, 1] (floor rows at -1, ridges near +1)."""
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    h = torch.arange(H, dtype=torch.float64).view(1, 1, H, 1)
    w = torch.arange(W, dtype=torch.float64).view(1, 1, 1, W)
    x = torch.sin(0.05 * h * (1 + 0.3 * b) + 0.5 * b) * torch.cos(0.021 * w + 0.1 * b) * torch.exp(-w / 300.0)
    x = x + 0.3 * torch.sin(0.37 * h + 0.11 * w * (1 + b))
    return torch.clamp(x * 1.2 - 0.2, -1.0, 1.0).to(dtype)


def synth_vec(shape, a, ph, dtype=torch.float64):
    n = int(np.prod(shape))
    return torch.sin(torch.arange(n, dtype=torch.float64) * a + ph).reshape(shape).to(dtype)


def keep_mask(shape, p, a, ph):
    u = 0.5 * (synth_vec(shape, a, ph) + 1.0)            # in [0,1], arcsine-distributed: fine for a fixed mask
    u = (u * 7919.0) % 1.0                               # scramble
    return (u >= p).to(torch.float64) / (1.0 - p)


class _MaskMul(nn.Module):
    """Stands in for nn.Dropout inside the reference model so the mask is known."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask.reshape(x.shape) if self.training else x


class _FixedNormal:
    eps = None

    def __init__(self, *a, **k):
        pass

    def sample(self):
        return _FixedNormal.eps


def checksum(t):
    t = t.detach().double().reshape(-1)
    n = t.numel()
    idx = torch.linspace(0, n - 1, steps=min(n, 64)).long()
    return np.array([t.sum().item(), t.abs().sum().item(), t.abs().max().item()]), t[idx].numpy(), idx.numpy()


def pack_big(prefix, t, out):
    cs, sample, idx = checksum(t)
    out[prefix + '/checksum'] = cs
    out[prefix + '/sample'] = sample
    out[prefix + '/sample_idx'] = idx


class FourLayerEncoder(nn.Module):
    """BASELINE.json's 4-layer stack assembled from the reference's OWN layer.Conv2D blocks with the hyper-parameters
    of encoder.py:241-248, followed by Dropout -> Linear (encoder.py:85), reshape (encoder.py:108)."""

    def __init__(self, dim_z, output_bn):
        super().__init__()
        act = nn.LeakyReLU
        self.single_ch_cnn = nn.Module()
        self.single_ch_cnn.enc_nn = nn.Sequential(
            ref_layer.Conv2D(1, 8, [5, 5], [2, 2], 2, [1, 1], batch_norm=None, activation=act(0.1), name_prefix='enc1'),
            ref_layer.Conv2D(8, 16, [4, 4], [2, 2], 2, [1, 1], activation=act(0.1), name_prefix='enc2'),
            ref_layer.Conv2D(16, 32, [4, 4], [2, 2], 2, [1, 1], activation=act(0.1), name_prefix='enc3'),
            ref_layer.Conv2D(32, 64, [4, 4], [2, 2], 2, [1, 1], activation=act(0.1), name_prefix='enc4'))
        self.mlp = nn.Sequential(nn.Dropout(0.3), nn.Linear(64 * 17 * 23, 2 * dim_z))
        if output_bn:
            self.mlp.add_module('lat_in_regularization', nn.BatchNorm1d(2 * dim_z))
        self.dim_z = dim_z

    def forward(self, x):
        h = self.single_ch_cnn.enc_nn(x).view(x.shape[0], -1)
        return torch.reshape(self.mlp(h), (x.shape[0], 2, self.dim_z))


class FourLayerDecoder(nn.Module):
    """Mirror: Linear -> Dropout (decoder.py:64-65) -> dec5..dec7 (decoder.py:212-217) -> ConvTranspose2d(8,1,5,2,2)
    -> Hardtanh (decoder.py:218-219), from the reference's own layer.TConv2D."""

    def __init__(self, dim_z):
        super().__init__()
        act = nn.LeakyReLU
        self.mlp = nn.Sequential(nn.Linear(dim_z, 64 * 17 * 23), nn.Dropout(0.3))
        self.single_ch_cnn = nn.Module()
        self.single_ch_cnn.dec_nn = nn.Sequential(
            ref_layer.TConv2D(64, 32, [4, 4], [2, 2], 2, output_padding=[1, 1], activation=act(0.1), name_prefix='dec5'),
            ref_layer.TConv2D(32, 16, [4, 4], [2, 2], 2, output_padding=[1, 0], activation=act(0.1), name_prefix='dec6'),
            ref_layer.TConv2D(16, 8, [4, 4], [2, 2], 2, output_padding=[1, 0], activation=act(0.1), name_prefix='dec7'),
            nn.ConvTranspose2d(8, 1, [5, 5], [2, 2], 2), nn.Hardtanh())

    def forward(self, z):
        return self.single_ch_cnn.dec_nn(self.mlp(z).view(-1, 64, 17, 23))


def build_reference_vae(arch, dim_z, B, output_bn, deepest_mix=False, n_ch=1, force_bigger=False):
    size = (B, n_ch, 257, 347)
    if arch == 'speccnn8l1_bn':
        enc = ref_encoder.SpectrogramEncoder(arch, dim_z, size, 0.3, output_bn=output_bn,
                                             deepest_features_mix=deepest_mix, force_bigger_network=force_bigger)
        dec = ref_decoder.SpectrogramDecoder(arch, dim_z, size, 0.3, force_bigger_network=force_bigger)
    else:
        enc, dec = FourLayerEncoder(dim_z, output_bn), FourLayerDecoder(dim_z)
    return ref_VAE.BasicVAE(enc, dim_z, dec, True, 'Dkl')


def stack_channels(x1, n_ch):
    """Extra spectrogram channels of the stacked-input cases: shifted, attenuated copies (a different 'MIDI note')."""
    return torch.cat([x1] + [torch.roll(x1, shifts=37 * c, dims=3) * (1.0 - 0.2 * c) for c in range(1, n_ch)], dim=1)


def run_vae_case(arch, dim_z, B, output_bn, tag, out_dir, n_ch=1, deepest_mix=False, force_bigger=False):
    vae = build_reference_vae(arch, dim_z, B, output_bn, deepest_mix=deepest_mix, n_ch=n_ch, force_bigger=force_bigger).double()
    template = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    sd = vo.closed_form_state_dict(template, seed=1234, dtype=torch.float64)
    vae.load_state_dict(sd)
    x = stack_channels(synth_input(B), n_ch)
    eps = synth_vec((B, dim_z), 1.2345, 0.4) * 1.3
    feat_enc = vae.encoder.mlp[1].in_features
    feat_dec = vae.decoder.mlp[0].out_features
    enc_mask = keep_mask((B, feat_enc), 0.3, 0.7071, 0.1)
    dec_mask = keep_mask((B, feat_dec), 0.3, 0.5772, 0.9)
    vae.encoder.mlp[0] = _MaskMul(enc_mask)
    vae.decoder.mlp[1] = _MaskMul(dec_mask)
    _FixedNormal.eps = eps
    out = {'meta/arch': np.array(arch), 'meta/dim_z': np.array(dim_z), 'meta/B': np.array(B),
           'meta/output_bn': np.array(output_bn), 'meta/seed': np.array(1234), 'meta/beta': np.array(0.2),
           'meta/lr': np.array(2e-4), 'meta/weight_decay': np.array(1e-4), 'meta/n_ch': np.array(n_ch),
           'meta/deepest_mix': np.array(deepest_mix), 'meta/force_bigger': np.array(force_bigger),
           'meta/keys': np.array(list(template.keys())),
           'meta/shapes': np.array([' '.join(str(d) for d in v) for v in template.values()]),
           'in/eps': eps.numpy(), 'in/enc_mask_bits': np.packbits((enc_mask > 0).numpy()),
           'in/dec_mask_bits': np.packbits((dec_mask > 0).numpy()),
           'in/enc_mask_shape': np.array(enc_mask.shape), 'in/dec_mask_shape': np.array(dec_mask.shape)}
    # ---- eval-mode forward (train.py:261-291 semantics: z = mu, BN running stats, no dropout)
    vae.eval()
    with torch.no_grad():
        zml, z, _, _, x_out = vae(x)
    out['eval/z_mu_logvar'] = zml.numpy()
    pack_big('eval/x_out', x_out, out)
    # ---- one train step (train.py:203-248 without the regression net)
    vae.train()
    opt = torch.optim.Adam(vae.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    with mock.patch.object(ref_VAE, 'Normal', _FixedNormal):
        opt.zero_grad()
        zml, z, zk, ladj, x_out = vae(x)
        recons = nn.MSELoss(reduction='mean')(x_out, x)
        lat = vae.latent_loss(zml)
        lat_unnorm = ref_loss.GaussianDkl(normalize=False)(zml[:, 0, :], zml[:, 1, :])
        l2 = ref_loss.L2Loss()(x_out, x)
        total = recons + 0.2 * lat
        total.backward()
    out['train/z_mu_logvar'] = zml.detach().numpy()
    out['train/z'] = z.detach().numpy()
    out['train/recons'] = np.array(recons.item())
    out['train/latent'] = np.array(lat.item())
    out['train/latent_unnormalized'] = np.array(lat_unnorm.item())
    out['train/l2loss'] = np.array(l2.item())
    out['train/total'] = np.array(total.item())
    pack_big('train/x_out', x_out, out)
    for k, p in vae.named_parameters():
        pack_big('grad/' + k, p.grad, out)
    opt.step()
    for k, v in vae.state_dict().items():
        if v.dtype == torch.long:
            out['post/' + k] = v.numpy()
        elif v.numel() <= 4096:
            out['post_full/' + k] = v.detach().numpy()
        else:
            pack_big('post/' + k, v, out)
    path = os.path.join(out_dir, tag + '.npz')
    np.savez_compressed(path, **out)
    print(tag, 'recons', recons.item(), 'lat', lat.item(), '->', path, os.path.getsize(path) // 1024, 'KiB')


def run_layer_cases(out_dir):
    """Every Conv2D/TConv2D configuration of the reference tables on small inputs, stored whole (fwd + grads)."""
    out = {}
    cases = []
    for (name, ci, co, k, s, p, bn) in vo.ENC_TABLE:
        cases.append(('conv', name, ci, co, k, s, p, (0, 0), bn))
    for (name, ci, co, k, s, p, op, bn) in vo.DEC_TABLE:
        cases.append(('tconv', name, ci, co, k, s, p, op, bn))
    cases.append(('tconv_last', 'dec8', 8, 1, 5, 2, 2, (0, 0), False))
    for kind, name, ci, co, k, s, p, op, bn in cases:
        ci_s, co_s = min(ci, 12), min(co, 10)        # channel-reduced copies keep the fixture small
        B, H, W = 3, (9 if kind == 'conv' else 5), (11 if kind == 'conv' else 6)
        if kind == 'conv':
            blk = ref_layer.Conv2D(ci_s, co_s, [k, k], [s, s], p, [1, 1], activation=nn.LeakyReLU(0.1),
                                   name_prefix=name, batch_norm=('after' if bn else None))
        elif kind == 'tconv':
            blk = ref_layer.TConv2D(ci_s, co_s, [k, k], [s, s], p, output_padding=list(op),
                                    activation=nn.LeakyReLU(0.1), name_prefix=name,
                                    batch_norm=('after' if bn else None))
        else:
            blk = nn.Sequential(nn.ConvTranspose2d(ci_s, co_s, [k, k], [s, s], p), nn.Hardtanh())
        blk = blk.double().train()
        template = {kk: tuple(v.shape) for kk, v in blk.state_dict().items()}
        sd = vo.closed_form_state_dict(template, seed=77, dtype=torch.float64)
        blk.load_state_dict(sd)
        x = (synth_vec((B, ci_s, H, W), 0.9137, 0.3) * 1.5).requires_grad_(True)
        y = blk(x)
        gy = synth_vec(tuple(y.shape), 0.7719, 1.1)
        y.backward(gy)
        pre = f'{name}/'
        out[pre + 'kind'] = np.array(kind)
        out[pre + 'cfg'] = np.array([ci_s, co_s, k, s, p, op[0], op[1], int(bn)])
        out[pre + 'x'] = x.detach().numpy()
        out[pre + 'gy'] = gy.numpy()
        out[pre + 'y'] = y.detach().numpy()
        out[pre + 'gx'] = x.grad.numpy()
        for kk, v in blk.named_parameters():
            out[pre + 'grad/' + kk] = v.grad.numpy()
        for kk, v in blk.state_dict().items():
            out[pre + 'sd_in/' + kk] = sd[kk].numpy()
            out[pre + 'sd_out/' + kk] = v.detach().numpy()
    path = os.path.join(out_dir, 'layers_small.npz')
    np.savez_compressed(path, **out)
    print('layers ->', path, os.path.getsize(path) // 1024, 'KiB')


def synth_wave(n=88576, sr=22050, idx=0):
    """Dexed-like FM voice: sin(2 pi fc t + I sin(2 pi fm t)) * ADSR, note-off at 3.0 s."""
    t = np.arange(n, dtype=np.float64) / sr
    fc = 261.63 * [0.5, 1.0, 2.0, 3.0][idx % 4]
    fm = fc * [1.0, 2.0, 3.5, 0.5][(idx // 4) % 4]
    I = 1.0 + 7.0 * ((idx * 37) % 11) / 10.0
    env = np.minimum(1.0, t / 0.01) * np.exp(-t * (0.3 + 0.2 * (idx % 3)))
    rel = np.where(t > 3.0, np.exp(-(t - 3.0) * 6.0), 1.0)
    fade = np.ones(n)
    nf = min(n, 2205)
    fade[-nf:] = np.linspace(1.0, 0.0, nf)
    return (0.9 * np.sin(2 * np.pi * fc * t + I * np.sin(2 * np.pi * fm * t)) * env * rel * fade).astype(np.float32)


def run_stft_cases(out_dir):
    """Reference utils/audio.py Spectrogram (pure torch) on seeded FM audio: STFT magnitude / norm factor / window /
    log-scale.  The mel stage (librosa, not installed) cannot be run: parity for the filterbank is unpinned."""
    spec = ref_audio.Spectrogram(1024, 256, -120.0)
    out = {'window': spec.window.numpy(), 'norm_factor': np.array(spec.spectrogram_norm_factor)}
    for idx in range(3):
        wav = synth_wave(idx=idx)
        stft = spec.get_stft(wav.astype(np.float64))
        mag = (stft.abs() / spec.spectrogram_norm_factor)
        db = spec(wav.astype(np.float64))
        assert tuple(db.shape) == (513, 347)
        frames = [0, 1, 2, 100, 173, 344, 345, 346]
        out[f'wave{idx}/frames'] = np.array(frames)
        out[f'wave{idx}/mag'] = mag[:, frames].numpy()
        out[f'wave{idx}/stft_re'] = stft.real[:, frames].numpy()      # get_stft: complex, un-normalised
        out[f'wave{idx}/stft_im'] = stft.imag[:, frames].numpy()
        out[f'wave{idx}/db'] = db[:, frames].numpy()
        out[f'wave{idx}/db_checksum'] = np.array([db.double().sum().item(), db.double().abs().sum().item(),
                                                  db.max().item(), db.min().item()])
    # log_scale=False and the dynamic-range variant of the log scale (utils/audio.py:42-50, :63-69)
    wav = synth_wave(idx=1)
    lin = ref_audio.Spectrogram(1024, 256, -120.0, log_scale=False)(wav.astype(np.float64))
    out['linear/frames'] = np.array(frames)
    out['linear/mag'] = lin[:, frames].numpy()
    dyn = ref_audio.Spectrogram(1024, 256, -120.0, dynamic_range_dB=60.0)
    out['dynrange/db'] = dyn.linear_to_log_scale_with_dynamic_range(lin)[:, frames].numpy()
    # short waveform: ragged/edge case (fewer samples than one FFT)
    short = synth_wave(n=700, idx=5)
    out['short/db'] = spec(short.astype(np.float64)).numpy()
    path = os.path.join(out_dir, 'stft.npz')
    np.savez_compressed(path, **out)
    print('stft ->', path, os.path.getsize(path) // 1024, 'KiB')


class _AllNumericalHelper:
    """Duck-typed PresetIndexesHelper for the all-numerical Dexed representation (144 learnable parameters, SURVEY.md
    §8c): the real class cannot be built here (numpy-2 assert at data/preset.py:41 and no preset database)."""
    learnable_preset_size = 144

    def get_numerical_learnable_indexes(self):
        return list(range(144))

    def get_categorical_learnable_indexes(self):
        return []


def run_regression_case(out_dir):
    """model/regression.py MLPRegression('3l1024') + the numeric branch of loss.SynthParamsLoss (model/loss.py:127-136)
    on the latent vectors of the 4-layer golden: v_out, preset-regression MSE, gradient w.r.t. z."""
    helper = _AllNumericalHelper()
    reg = ref_regression.MLPRegression('3l1024', 64, helper, dropout_p=0.4, cat_softmax_activation=False).double()
    template = {k: tuple(v.shape) for k, v in reg.state_dict().items()}
    sd = vo.closed_form_state_dict(template, seed=4321, dtype=torch.float64)
    reg.load_state_dict(sd)
    B = 6
    z = (synth_vec((B, 64), 0.913, 0.2) * 1.1).requires_grad_(True)
    v_in = 0.5 * (synth_vec((B, 144), 0.377, 0.6) + 1.0)
    masks = [keep_mask((B, 1024), 0.4, 0.61 + 0.1 * i, 0.2 * i) for i in range(2)]   # keep/(1-p), p = 0.4
    reg.reg_model.drp1 = _MaskMul(masks[0])
    reg.reg_model.drp2 = _MaskMul(masks[1])
    reg.train()
    v_out = reg(z)
    crit = ref_loss.SynthParamsLoss(helper, True, prevent_useless_params_loss=False, cat_bce=False, cat_softmax=True,
                                    cat_softmax_t=0.2)
    loss = crit(v_out.clone(), v_in.clone())
    loss.backward()
    out = {'in/z': z.detach().numpy(), 'in/v_in': v_in.numpy(),
           'in/mask0_bits': np.packbits((masks[0] > 0).numpy()), 'in/mask1_bits': np.packbits((masks[1] > 0).numpy()),
           'out/v_out': v_out.detach().numpy(), 'out/mse': np.array(loss.item()), 'out/g_z': z.grad.numpy()}
    for k, p in reg.named_parameters():
        pack_big('grad/' + k, p.grad, out)
    for k, v in reg.state_dict().items():
        if 'running' in k:
            out['post_full/' + k] = v.numpy()
    path = os.path.join(out_dir, 'regression_b6.npz')
    np.savez_compressed(path, **out)
    print('regression mse', loss.item(), '->', path, os.path.getsize(path) // 1024, 'KiB')


def run_regstep_case(out_dir):
    """The whole minibatch body of train.py:203-248 WITH the preset-regression network: ae_out = VAE(x), v_out =
    reg_model(z_K) (train.py:220), recons + beta * lat + cont_loss (numeric SynthParamsLoss, train.py:238-246), one
    backward through both networks (the controls gradient reaches the encoder through z_K), Adam over all parameters of
    the extended model (train.py:166-167)."""
    arch, dim_z, B = 'speccnn4l1_bn', 64, 4
    vae = build_reference_vae(arch, dim_z, B, False).double()
    template = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    sd = vo.closed_form_state_dict(template, seed=1234, dtype=torch.float64)
    vae.load_state_dict(sd)
    helper = _AllNumericalHelper()
    reg = ref_regression.MLPRegression('3l1024', dim_z, helper, dropout_p=0.4, cat_softmax_activation=False).double()
    rtemplate = {k: tuple(v.shape) for k, v in reg.state_dict().items()}
    reg.load_state_dict(vo.closed_form_state_dict(rtemplate, seed=4321, dtype=torch.float64))
    x = synth_input(B)
    eps = synth_vec((B, dim_z), 1.2345, 0.4) * 1.3
    v_in = 0.5 * (synth_vec((B, 144), 0.377, 0.6) + 1.0)
    enc_mask = keep_mask((B, vae.encoder.mlp[1].in_features), 0.3, 0.7071, 0.1)
    dec_mask = keep_mask((B, vae.decoder.mlp[0].out_features), 0.3, 0.5772, 0.9)
    vae.encoder.mlp[0] = _MaskMul(enc_mask)
    vae.decoder.mlp[1] = _MaskMul(dec_mask)
    rmasks = [keep_mask((B, 1024), 0.4, 0.61 + 0.1 * i, 0.2 * i) for i in range(2)]
    reg.reg_model.drp1 = _MaskMul(rmasks[0])
    reg.reg_model.drp2 = _MaskMul(rmasks[1])
    _FixedNormal.eps = eps
    vae.train(), reg.train()
    params = list(vae.parameters()) + list(reg.parameters())
    opt = torch.optim.Adam(params, lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    crit = ref_loss.SynthParamsLoss(helper, True, prevent_useless_params_loss=False, cat_bce=False, cat_softmax=True,
                                    cat_softmax_t=0.2)
    with mock.patch.object(ref_VAE, 'Normal', _FixedNormal):
        opt.zero_grad()
        zml, z0, zk, ladj, x_out = vae(x)
        v_out = reg(zk)
        recons = nn.MSELoss(reduction='mean')(x_out, x)
        lat = vae.latent_loss(zml)
        lat_b = lat * 0.2
        cont = crit(v_out, v_in.clone())
        (recons + lat_b + torch.tensor([0.0], dtype=torch.float64) + cont).backward()
    out = {'meta/arch': np.array(arch), 'meta/dim_z': np.array(dim_z), 'meta/B': np.array(B),
           'meta/seed': np.array(1234), 'meta/reg_seed': np.array(4321), 'meta/beta': np.array(0.2),
           'meta/lr': np.array(2e-4), 'meta/weight_decay': np.array(1e-4),
           'meta/reg_keys': np.array(list(rtemplate.keys())),
           'meta/reg_shapes': np.array([' '.join(str(d) for d in v) for v in rtemplate.values()]),
           'in/eps': eps.numpy(), 'in/v_in': v_in.numpy(),
           'in/enc_mask_bits': np.packbits((enc_mask > 0).numpy()), 'in/dec_mask_bits': np.packbits((dec_mask > 0).numpy()),
           'in/enc_mask_shape': np.array(enc_mask.shape), 'in/dec_mask_shape': np.array(dec_mask.shape),
           'in/reg_mask0_bits': np.packbits((rmasks[0] > 0).numpy()),
           'in/reg_mask1_bits': np.packbits((rmasks[1] > 0).numpy()),
           'train/z_mu_logvar': zml.detach().numpy(), 'train/v_out': v_out.detach().numpy(),
           'train/recons': np.array(recons.item()), 'train/latent': np.array(lat.item()),
           'train/controls': np.array(cont.item()), 'train/total': np.array((recons + lat_b + cont).item())}
    pack_big('train/x_out', x_out, out)
    for k, p in vae.named_parameters():
        pack_big('grad/' + k, p.grad, out)
    for k, p in reg.named_parameters():
        pack_big('grad_reg/' + k, p.grad, out)
    opt.step()
    for k, v in list(vae.state_dict().items()) + [('reg.' + k, v) for k, v in reg.state_dict().items()]:
        if v.dtype == torch.long:
            out['post/' + k] = v.numpy()
        elif v.numel() <= 4096:
            out['post_full/' + k] = v.detach().numpy()
        else:
            pack_big('post/' + k, v, out)
    path = os.path.join(out_dir, 'regstep_4l_b4.npz')
    np.savez_compressed(path, **out)
    print('regstep recons', recons.item(), 'lat', lat.item(), 'cont', cont.item(), '->', path,
          os.path.getsize(path) // 1024, 'KiB')


def run_extended_keys_case(out_dir):
    """What a checkpoint of the reference holds (logs/logger.py:199-202: ``extended_ae_model.state_dict()`` and
    ``optimizer.state_dict()``): key names, registration order and shapes of the reference's OWN ``ExtendedAE`` around its
    BasicVAE + MLPRegression (prefixes ``ae_model.`` / ``reg_model.``), for both latent regularisations, and the layout of
    an Adam ``state_dict()`` over ``extended_ae_model.parameters()`` after one step (train.py:166-167) - data, no source."""
    from model import extendedAE as ref_extendedAE
    out = {}
    for tag, output_bn in (('none', False), ('bn', True)):
        vae = build_reference_vae('speccnn8l1_bn', 64, 2, output_bn)
        helper = _AllNumericalHelper()
        reg = ref_regression.MLPRegression('3l1024', 64, helper, dropout_p=0.4, cat_softmax_activation=False)
        ext = ref_extendedAE.ExtendedAE(vae, reg, helper, 0.3)
        sd = ext.state_dict()
        out[f'{tag}/keys'] = np.array(list(sd.keys()))
        out[f'{tag}/shapes'] = np.array([' '.join(str(d) for d in v.shape) for v in sd.values()])
        out[f'{tag}/dtypes'] = np.array([str(v.dtype) for v in sd.values()])
        out[f'{tag}/param_names'] = np.array([k for k, _ in ext.named_parameters()])
        if tag == 'none':
            opt = torch.optim.Adam(ext.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-4)
            for p_ in ext.parameters():
                p_.grad = torch.zeros_like(p_)
            opt.step()
            osd = opt.state_dict()
            out['adam/top_keys'] = np.array(sorted(osd.keys()))
            out['adam/group_keys'] = np.array(sorted(k for k in osd['param_groups'][0].keys()))
            out['adam/n_params'] = np.array(len(osd['param_groups'][0]['params']))
            out['adam/param_ids'] = np.array(osd['param_groups'][0]['params'])
            out['adam/state_entry_keys'] = np.array(sorted(osd['state'][0].keys()))
            out['adam/state_shapes'] = np.array([' '.join(str(d) for d in osd['state'][i]['exp_avg'].shape)
                                                 for i in range(len(osd['state']))])
            out['adam/step_after_one'] = np.array(float(osd['state'][0]['step']))
    path = os.path.join(out_dir, 'extended_keys.npz')
    np.savez_compressed(path, **out)
    print('extended_keys', len(out['none/keys']), 'keys ->', path, os.path.getsize(path) // 1024, 'KiB')


def run_regstep_cat_case(out_dir):
    """train.py:203-248 as ``run_regstep_case``, with the criteria the reference actually trains with: controls loss =
    ``SynthParamsLoss`` INCLUDING its categorical branch and the useless-parameter exclusion (train.py:111-116, the
    default cat_softmax configuration), and the two monitoring metrics evaluated on every minibatch under ``no_grad``
    before it (train.py:229-233: QuantizedNumericalParamsLoss, CategoricalParamsAccuracy), on the 14-column
    representation of tests/helpers.MiniPresetIndexesHelper."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import MiniPresetIndexesHelper
    arch, dim_z, B = 'speccnn4l1_bn', 64, 4
    vae = build_reference_vae(arch, dim_z, B, False).double()
    template = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    vae.load_state_dict(vo.closed_form_state_dict(template, seed=1234, dtype=torch.float64))
    helper = MiniPresetIndexesHelper()
    reg = ref_regression.MLPRegression('3l1024', dim_z, helper, dropout_p=0.4, cat_softmax_activation=False).double()
    rtemplate = {k: tuple(v.shape) for k, v in reg.state_dict().items()}
    reg.load_state_dict(vo.closed_form_state_dict(rtemplate, seed=4321, dtype=torch.float64))
    x = synth_input(B)
    eps = synth_vec((B, dim_z), 1.2345, 0.4) * 1.3
    L = helper.learnable_preset_size
    v_in = 0.5 * (synth_vec((B, L), 0.377, 0.6) + 1.0)
    v_in[:, 1] = torch.round(v_in[:, 1] * 4) / 4
    v_in[:, 2] = torch.round(v_in[:, 2] * 2) / 2
    v_in[:, 13] = torch.round(v_in[:, 13] * 3) / 3
    v_in[2, 0] = 0.0                                   # 'operator volumes' at zero: rules of the helper fire
    v_in[3, 1] = 0.0
    for g in helper.get_categorical_learnable_indexes():
        cls = (torch.arange(B) * 3 + len(g)) % len(g)
        v_in[:, g] = torch.nn.functional.one_hot(cls, len(g)).double()
    enc_mask = keep_mask((B, vae.encoder.mlp[1].in_features), 0.3, 0.7071, 0.1)
    dec_mask = keep_mask((B, vae.decoder.mlp[0].out_features), 0.3, 0.5772, 0.9)
    vae.encoder.mlp[0] = _MaskMul(enc_mask)
    vae.decoder.mlp[1] = _MaskMul(dec_mask)
    rmasks = [keep_mask((B, 1024), 0.4, 0.61 + 0.1 * i, 0.2 * i) for i in range(2)]
    reg.reg_model.drp1 = _MaskMul(rmasks[0])
    reg.reg_model.drp2 = _MaskMul(rmasks[1])
    _FixedNormal.eps = eps
    vae.train(), reg.train()
    params = list(vae.parameters()) + list(reg.parameters())
    opt = torch.optim.Adam(params, lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-4)
    crit = ref_loss.SynthParamsLoss(helper, True, cat_bce=False, cat_softmax=True, cat_softmax_t=0.2)   # train.py:111-116
    qloss_crit = ref_loss.QuantizedNumericalParamsLoss(helper, numerical_loss=nn.MSELoss(reduction='mean'))  # :118-119
    acc_crit = ref_loss.CategoricalParamsAccuracy(helper, reduce=True, percentage_output=True)               # :120-121
    with mock.patch.object(ref_VAE, 'Normal', _FixedNormal):
        opt.zero_grad()
        zml, z0, zk, ladj, x_out = vae(x)
        v_out = reg(zk)
        recons = nn.MSELoss(reduction='mean')(x_out, x)
        lat = vae.latent_loss(zml)
        lat_b = lat * 0.2
        with torch.no_grad():                                                     # train.py:229-233
            qloss = qloss_crit(v_out, v_in)
            acc = acc_crit(v_out, v_in)
        v_out_seen = v_out.detach().clone()
        cont = crit(v_out, v_in.clone())          # (writes zeros into the useless columns of both arguments, loss.py:134-135)
        (recons + lat_b + torch.tensor([0.0], dtype=torch.float64) + cont).backward()
    out = {'meta/arch': np.array(arch), 'meta/dim_z': np.array(dim_z), 'meta/B': np.array(B),
           'meta/seed': np.array(1234), 'meta/reg_seed': np.array(4321), 'meta/beta': np.array(0.2),
           'meta/lr': np.array(2e-4), 'meta/weight_decay': np.array(1e-4),
           'meta/reg_keys': np.array(list(rtemplate.keys())),
           'meta/reg_shapes': np.array([' '.join(str(d) for d in v) for v in rtemplate.values()]),
           'in/eps': eps.numpy(), 'in/v_in': v_in.numpy(),
           'in/enc_mask_bits': np.packbits((enc_mask > 0).numpy()), 'in/dec_mask_bits': np.packbits((dec_mask > 0).numpy()),
           'in/enc_mask_shape': np.array(enc_mask.shape), 'in/dec_mask_shape': np.array(dec_mask.shape),
           'in/reg_mask0_bits': np.packbits((rmasks[0] > 0).numpy()),
           'in/reg_mask1_bits': np.packbits((rmasks[1] > 0).numpy()),
           'train/z_mu_logvar': zml.detach().numpy(), 'train/v_out': v_out_seen.numpy(),
           'train/recons': np.array(recons.item()), 'train/latent': np.array(lat.item()),
           'train/controls': np.array(cont.item()), 'train/total': np.array((recons + lat_b + cont).item()),
           'train/qloss': np.array(float(qloss)), 'train/accuracy': np.array(float(acc))}
    for k, p in reg.named_parameters():
        pack_big('grad_reg/' + k, p.grad, out)
    for k in ('encoder.mlp.1.weight', 'encoder.mlp.1.bias'):
        pack_big('grad/' + k, dict(vae.named_parameters())[k].grad, out)
    path = os.path.join(out_dir, 'regstep_4l_b4_cat.npz')
    np.savez_compressed(path, **out)
    print('regstep_cat recons', recons.item(), 'lat', lat.item(), 'cont', cont.item(), 'qloss', float(qloss), 'acc',
          float(acc), '->', path, os.path.getsize(path) // 1024, 'KiB')


def run_probability_case(out_dir):
    """utils/probability.py:13-29 and the ELBO combination of FlowVAE.latent_loss (VAE.py:183-193; the flow output z_K
    and log|det J| are given tensors here - the nflows transform is out of scope)."""
    from utils import probability as ref_prob
    B, D = 6, 16
    mu, lv = synth_vec((B, D), 0.913, 0.3) * 0.7, synth_vec((B, D), 0.477, 1.1) * 0.9 - 0.2
    z0 = mu + torch.exp(lv / 2) * synth_vec((B, D), 1.37, 0.5) * 1.2
    zk = synth_vec((B, D), 0.61, 0.8) * 1.5 + 0.1 * z0
    ladj = synth_vec((B, 1), 2.3, 0.2) * 0.4
    log_q = ref_prob.gaussian_log_probability(z0, mu, lv)
    log_p = ref_prob.standard_gaussian_log_probability(zk)
    loss = -(log_p - log_q + ladj).mean()
    out = {'mu': mu.numpy(), 'logvar': lv.numpy(), 'z0': z0.numpy(), 'zk': zk.numpy(), 'ladj': ladj.numpy(),
           'log_q': log_q.numpy(), 'log_p': log_p.numpy(), 'loss': np.array(loss.item()),
           'loss_normalized': np.array(loss.item() / D)}
    path = os.path.join(out_dir, 'probability.npz')
    np.savez_compressed(path, **out)
    print('probability ->', path)


def run_params_loss_case(out_dir):
    """the reference's SynthParamsLoss (categorical branches, useless-parameter exclusion),
    QuantizedNumericalParamsLoss and CategoricalParamsAccuracy (model/loss.py:72-315) driven with a duck-typed
    14-column PresetIndexesHelper (tests/helpers.py)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import MiniPresetIndexesHelper
    helper = MiniPresetIndexesHelper()
    B, L = 8, 14
    u_in = 0.5 * (synth_vec((B, L), 0.377, 0.6) + 1.0)
    u_in[:, 1] = torch.round(u_in[:, 1] * 4) / 4
    u_in[:, 2] = torch.round(u_in[:, 2] * 2) / 2
    u_in[:, 13] = torch.round(u_in[:, 13] * 3) / 3
    u_in[2, 0] = 0.0          # rows whose 'operator volume' makes parameters useless
    u_in[5, 0] = 0.0
    u_in[3, 1] = 0.0
    u_in[5, 1] = 0.0
    for g in helper.get_categorical_learnable_indexes():
        cls = (torch.arange(B) * 3 + len(g)) % len(g)
        u_in[:, g] = torch.nn.functional.one_hot(cls, len(g)).double()
    raw = 0.5 * (synth_vec((B, L), 0.913, 1.7) + 1.0) * 0.98 + 0.01          # in (0, 1)
    out = {'in/u_in': u_in.numpy(), 'in/u_out': raw.numpy()}
    variants = {'cce_softmax': dict(cat_bce=False, cat_softmax=True, cat_softmax_t=0.2),
                'cce_probs': dict(cat_bce=False, cat_softmax=False), 'bce': dict(cat_bce=True, cat_softmax=False)}
    for norm in (True, False):
        for useless in (True, False):
            for name, kw in variants.items():
                u_out = raw.clone().requires_grad_(True)
                crit = ref_loss.SynthParamsLoss(helper, norm, categorical_loss_factor=0.2,
                                                prevent_useless_params_loss=useless, **kw)
                loss = crit(u_out.clone(), u_in.clone())      # the reference mutates its arguments in place
                loss.backward()
                tag = f'{name}/norm{int(norm)}/useless{int(useless)}'
                out[tag + '/loss'] = np.array(loss.item())
                out[tag + '/grad'] = u_out.grad.numpy()
    out['quantized/mse'] = np.array(ref_loss.QuantizedNumericalParamsLoss(helper)(raw, u_in).item())
    out['quantized/limited'] = np.array(ref_loss.QuantizedNumericalParamsLoss(
        helper, limited_vst_params_indexes=[1, 5])(raw, u_in).item())
    out['accuracy/mean'] = np.array(ref_loss.CategoricalParamsAccuracy(helper)(raw, u_in))
    acc = ref_loss.CategoricalParamsAccuracy(helper, reduce=False, percentage_output=False)(raw, u_in)
    out['accuracy/keys'] = np.array(list(acc.keys()))
    out['accuracy/values'] = np.array(list(acc.values()))
    # PresetActivation with the softmax on categorical sub-vectors (model/regression.py:35-51)
    act = ref_regression.PresetActivation(helper, cat_softmax_activation=True)
    xin = (synth_vec((B, L), 0.531, 0.9) * 2.0)
    leaf = xin.clone().requires_grad_(True)
    y = act(leaf * 1.0)                               # (the reference writes into its argument in place)
    gy = synth_vec((B, L), 0.77, 0.4)
    (y * gy).sum().backward()
    out['act_softmax/in'], out['act_softmax/out'] = xin.numpy(), y.detach().numpy()
    out['act_softmax/gy'], out['act_softmax/gx'] = gy.numpy(), leaf.grad.numpy()
    # the Dexed useless-parameter rule of data/preset.py:259-281 through the reference's own method
    from data import preset as ref_preset
    f2l = [None] * 155
    learn = 0
    for vst_idx in range(155):
        if vst_idx % 7 == 3:
            continue                                   # not learnable
        if vst_idx % 5 == 2 and vst_idx not in [31 + 22 * i for i in range(6)]:
            f2l[vst_idx] = [learn, learn + 1, learn + 2]
            learn += 3
        else:
            f2l[vst_idx] = learn
            learn += 1
    fake = types.SimpleNamespace(_synth=ref_preset._Synth.DEXED, full_to_learnable=f2l)
    preset = 0.5 * (synth_vec((learn,), 0.61, 0.3) + 1.0)
    for op in (1, 4):
        preset[f2l[31 + 22 * op]] = 0.0
    nums, cats = ref_preset.PresetIndexesHelper.get_useless_learned_params_indexes(fake, preset)
    out['dexed/full_to_learnable'] = np.array([-1 if v is None else (v if isinstance(v, int) else -(v[0] + 2))
                                               for v in f2l])
    out['dexed/preset'] = preset.numpy()
    out['dexed/useless_num'] = np.array(nums)
    out['dexed/useless_cat'] = np.array(cats)
    path = os.path.join(out_dir, 'params_loss.npz')
    np.savez_compressed(path, **out)
    print('params loss ->', path, os.path.getsize(path) // 1024, 'KiB')


def run_dataset_seam_case(out_dir):
    """the reference's own PresetDataset.__getitem__ / denormalize_spectrogram (data/abstractbasedataset.py
    :101-145, :340-346) driven through a fixture subclass that serves synthetic waves and parameter vectors from memory
    (linear spectrogram: librosa, needed for the mel variant, is not installed)."""
    from data import abstractbasedataset as ref_ds

    n_presets, notes, n_samples, L = 3, ((60, 85), (72, 100)), 4096, 5
    waves = np.stack([np.stack([synth_wave(n_samples, idx=10 * p + j) for j in range(len(notes))])
                      for p in range(n_presets)])
    params = synth_vec((n_presets, L), 0.733, 0.21).numpy() * 0.5 + 0.5
    uids = np.array([1007, 23, 501])

    class _Params:
        def __init__(self, row):
            self.row = row

        def get_learnable(self):
            return torch.tensor(self.row).unsqueeze(0)

    class _Fixture(ref_ds.PresetDataset):
        synth_name = 'fixture'
        total_nb_presets = n_presets
        total_nb_params = L

        def get_full_preset_params(self, preset_UID):
            return _Params(params[int(np.where(uids == preset_UID)[0][0])])

        def _render_audio(self, preset_params, midi_note, midi_velocity):
            raise NotImplementedError

        def get_wav_file(self, preset_UID, midi_note, midi_velocity):
            pi = int(np.where(uids == preset_UID)[0][0])
            ni = [n[0] for n in notes].index(midi_note)
            return waves[pi, ni], 22050

    out = {'in/waves': waves, 'in/params': params, 'in/uids': uids, 'in/midi_notes': np.array(notes)}
    stats = {'min': -120.0, 'max': -2.5, 'mean': -71.25, 'std': 23.5}
    for tag, stacked, norm in (('flat_minmax', False, 'min_max'), ('stacked_minmax', True, 'min_max'),
                               ('flat_meanstd', False, 'mean_std'), ('flat_none', False, None)):
        ds = _Fixture((3.0, 1.0), 1024, 256, midi_notes=notes, multichannel_stacked_spectrograms=stacked, n_mel_bins=-1,
                      spectrogram_min_dB=-120.0, spectrogram_normalization=norm)
        ds.valid_preset_UIDs = uids
        ds.spec_stats = stats
        out[f'{tag}/len'] = np.array(len(ds))
        for i in range(len(ds)):
            spec, par, info, labels = ds[i]
            out[f'{tag}/{i}/spec'] = spec.numpy()
            out[f'{tag}/{i}/params'] = par.numpy()
            out[f'{tag}/{i}/info'] = info.numpy()
            out[f'{tag}/{i}/labels'] = labels.numpy()
        if norm is not None:
            out[f'{tag}/denorm0'] = ds.denormalize_spectrogram(torch.tensor(out[f'{tag}/0/spec'])).numpy()
    for k, v in stats.items():
        out['in/stats_' + k] = np.array(v)
    path = os.path.join(out_dir, 'dataset_seam.npz')
    np.savez_compressed(path, **out)
    print('dataset seam ->', path, os.path.getsize(path) // 1024, 'KiB')


def run_mel_basis_case(out_dir):
    """Pin for the mel filterbank.  The reference calls librosa.feature.melspectrogram(S=.., n_mels=257,
    norm=None) (utils/audio.py:85-86; librosa ~=0.8.0 is neither vendored nor installed in this image).  What the image
    does have is ``transformers.audio_utils.mel_filter_bank``, an INDEPENDENT implementation that documents its
    ``mel_scale="slaney", norm=None`` mode as reproducing ``librosa.filters.mel(htk=False, norm=None)``.  The golden is
    that implementation's basis for the reference's parameters, stored as CSR (1016 non-zeros of 257 x 513) - not
    librosa 0.8 itself, which this container cannot run."""
    # (transformers probes optional packages with importlib.util.find_spec, which rejects the MagicMock stubs above)
    stubs = {k: sys.modules.pop(k) for k in ('librosa', 'librosa.display', 'soundfile') if k in sys.modules}
    try:
        from transformers.audio_utils import mel_filter_bank
    finally:
        sys.modules.update(stubs)
    fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=257, min_frequency=0.0, max_frequency=11025.0,
                         sampling_rate=22050, norm=None, mel_scale="slaney")           # [513, 257]
    fb = np.asarray(fb, dtype=np.float64).T                                            # [n_mels, n_bins]
    rows, cols = np.nonzero(fb)
    row_ptr = np.zeros(258, dtype=np.int32)
    np.add.at(row_ptr, rows + 1, 1)
    np.savez_compressed(os.path.join(out_dir, 'mel_basis.npz'), shape=np.array(fb.shape, dtype=np.int32),
                        row_ptr=np.cumsum(row_ptr).astype(np.int32), col=cols.astype(np.int32), val=fb[rows, cols],
                        row_sums=fb.sum(axis=1), source=np.array('transformers.audio_utils.mel_filter_bank '
                                                                 '(norm=None, mel_scale="slaney")'))
    print('mel_basis', fb.shape, 'nnz', len(rows), 'max taps per row', int(np.diff(np.cumsum(row_ptr)).max()))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'mel_basis':
        run_mel_basis_case(HERE)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'dataset_seam':
        run_dataset_seam_case(HERE)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'params_loss':
        run_params_loss_case(HERE)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'probability':
        run_probability_case(HERE)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'regstep':
        run_regstep_case(HERE)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'regstep_cat':
        run_regstep_cat_case(HERE)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'b16':
        run_vae_case('speccnn8l1_bn', 64, 16, False, 'vae8l_b16', HERE)
        run_vae_case('speccnn4l1_bn', 64, 16, False, 'vae4l_b16', HERE)
        run_vae_case('speccnn8l1_bn', 64, 16, True, 'vae8l_b16_outbn', HERE)
        run_vae_case('speccnn4l1_bn', 64, 16, True, 'vae4l_b16_outbn', HERE)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'stacked':
        run_vae_case('speccnn8l1_bn', 64, 2, False, 'vae8l_b2_c2', HERE, n_ch=2, deepest_mix=False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'variants':
        # the reference's other encoder variants (encoder.py:54-70): stacked channels mixed by the deepest 1x1 layer
        # (512 * n_ch -> 1024), and the 1800-channel 4x4 layer of force_bigger_network (decoder.py:36,70 mirror it)
        run_vae_case('speccnn8l1_bn', 64, 2, False, 'vae8l_b2_c2_mix', HERE, n_ch=2, deepest_mix=True)
        run_vae_case('speccnn8l1_bn', 64, 2, False, 'vae8l_b2_big', HERE, force_bigger=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'extended_keys':
        run_extended_keys_case(HERE)
        sys.exit(0)
    run_regression_case(HERE)
    run_layer_cases(HERE)
    run_stft_cases(HERE)
    run_vae_case('speccnn8l1_bn', 64, 2, False, 'vae8l_b2', HERE)
    run_vae_case('speccnn8l1_bn', 64, 2, True, 'vae8l_b2_outbn', HERE)
    run_vae_case('speccnn4l1_bn', 64, 2, False, 'vae4l_b2', HERE)
    run_vae_case('speccnn8l1_bn', 64, 2, False, 'vae8l_b2_c2', HERE, n_ch=2, deepest_mix=False)
    run_vae_case('speccnn8l1_bn', 64, 2, False, 'vae8l_b2_c2_mix', HERE, n_ch=2, deepest_mix=True)
    run_vae_case('speccnn8l1_bn', 64, 2, False, 'vae8l_b2_big', HERE, force_bigger=True)
    run_extended_keys_case(HERE)
    run_vae_case('speccnn8l1_bn', 64, 16, False, 'vae8l_b16', HERE)       # B = 16 capture
    run_vae_case('speccnn4l1_bn', 64, 16, False, 'vae4l_b16', HERE)
    # the reference's DEFAULT regularisation ('bn': BatchNorm1d on the encoder output, config.py:92) at the capture size
    run_vae_case('speccnn8l1_bn', 64, 16, True, 'vae8l_b16_outbn', HERE)
    run_vae_case('speccnn4l1_bn', 64, 16, True, 'vae4l_b16_outbn', HERE)
    run_regstep_case(HERE)
    run_regstep_cat_case(HERE)
    run_dataset_seam_case(HERE)
    run_probability_case(HERE)
    run_params_loss_case(HERE)
    run_mel_basis_case(HERE)
