"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's conv-VAE train-step arithmetic.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package (``preset-gen-vae_amd/``) never does.  It restates, function by function, what the reference's
Python computes with stock torch ops (the reference has no native code), citing the file:line each function follows
under /root/reference.  It runs on the CPU in float32 or float64 and works on plain ``{state-dict key: tensor}``
dictionaries with the reference's key names, so it can be fed by the reference's own modules, by the fixtures in
``tests/golden/`` and by the product modules alike.

Pinning: ``tests/golden/make_goldens.py`` imports the real reference (model/layer.py, encoder.py, decoder.py, VAE.py,
loss.py) in the build container, runs it on seeded inputs and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

import contextlib

LRELU_SLOPE = 0.1          # encoder.py:240, decoder.py:204
BN_EPS, BN_MOMENTUM = 1e-5, 0.1   # torch defaults through layer.py:21,41

# (name, in_ch, out_ch, kernel, stride, pad, has_bn) — encoder.py:241-259 and mixer encoder.py:56-69
ENC_TABLE = [('enc1', 1, 8, 5, 2, 2, False), ('enc2', 8, 16, 4, 2, 2, True), ('enc3', 16, 32, 4, 2, 2, True),
             ('enc4', 32, 64, 4, 2, 2, True), ('enc5', 64, 128, 4, 2, 2, True), ('enc6', 128, 256, 4, 2, 2, True),
             ('enc7', 256, 512, 4, 2, 2, True), ('enc8', 512, 2048, 1, 1, 0, False)]
# (name, in_ch, out_ch, kernel, stride, pad, output_padding, has_bn) — decoder.py:72-75 and :205-218
DEC_TABLE = [('dec1', 2048, 512, 1, 1, 0, (0, 0), True), ('dec2', 512, 256, 4, 2, 2, (1, 1), True),
             ('dec3', 256, 128, 4, 2, 2, (1, 0), True), ('dec4', 128, 64, 4, 2, 2, (1, 1), True),
             ('dec5', 64, 32, 4, 2, 2, (1, 1), True), ('dec6', 32, 16, 4, 2, 2, (1, 0), True),
             ('dec7', 16, 8, 4, 2, 2, (1, 0), True)]


def arch_tables(arch):
    """Layer tables per architecture string.  'speccnn4l1_bn' = first four encoder rows / last three decoder rows
    (BASELINE.json's 4-layer conv-VAE, assembled from the reference's own blocks)."""
    if arch == 'speccnn8l1_bn':
        return ENC_TABLE, DEC_TABLE, (2048, 3, 4)
    if arch == 'speccnn4l1_bn':
        return ENC_TABLE[:4], DEC_TABLE[4:], (64, 17, 23)
    raise NotImplementedError(arch)


def _find(sd, suffix, scope):
    hits = [k for k in sd if k.endswith(suffix) and scope in k]
    if len(hits) != 1:
        raise KeyError(f"{suffix!r} in scope {scope!r}: {hits}")
    return sd[hits[0]]


def _bn_train_or_eval(a, sd, prefix, scope, training, new_buffers):
    """nn.BatchNorm2d / 1d (layer.py:21-26): train = biased batch variance for normalisation, unbiased for the
    running estimate, momentum 0.1; eval = running statistics."""
    gamma, beta = _find(sd, prefix + 'bn.weight', scope), _find(sd, prefix + 'bn.bias', scope)
    rm, rv = _find(sd, prefix + 'bn.running_mean', scope), _find(sd, prefix + 'bn.running_var', scope)
    if new_buffers is not None and prefix + 'bn.running_mean' in new_buffers:
        # a stack applied once per spectrogram channel (encoder.py:101-102) updates its running statistics every time
        rm, rv = new_buffers[prefix + 'bn.running_mean'], new_buffers[prefix + 'bn.running_var']
    dims = [0] + list(range(2, a.dim()))
    shape = [1, -1] + [1] * (a.dim() - 2)
    if training:
        mean = a.mean(dim=dims)
        var = a.var(dim=dims, unbiased=False)
        n = a.numel() / a.shape[1]
        if new_buffers is not None:
            new_buffers[prefix + 'bn.running_mean'] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean).detach()
            new_buffers[prefix + 'bn.running_var'] = ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / (n - 1)).detach()
    else:
        mean, var = rm, rv
    return (a - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * gamma.view(shape) + beta.view(shape)


def _leaky(y, name, act_masks):
    """LeakyReLU(0.1).  ``act_masks`` (test aid) pins the linear piece of every element: {block name: bool tensor,
    True = positive side}.  A pre-activation within ~1e-7 of the kink lands on either side depending on float32
    summation order; evaluating the reference arithmetic in the SAME linear region as the implementation under test
    makes the gradient comparison well-posed (values change by < 1e-7, derivatives by 0.9 for a flipped element)."""
    if act_masks is not None and name in act_masks:
        return torch.where(act_masks[name], y, LRELU_SLOPE * y)
    return F.leaky_relu(y, LRELU_SLOPE)


# ---- operand precision of the products ---------------------------------------------------------------------------
# The reference trains in float32 (train.py builds the model with default dtypes, no autocast).  BASELINE.json's
# config 2 asks for the 8-layer stack with bf16 matrix cores; the product's definition of that mode (DESIGN.md
# section 7) is restated here so that it has an oracle too: every convolution / transposed convolution / linear
# PRODUCT - forward, input gradient and weight gradient - sees both of its operands rounded to bfloat16 (round to
# nearest even) and accumulates exactly-representable products in float32; biases, activations, BatchNorm, losses,
# gradients in memory, Adam and the parameters themselves stay float32.  This mode has no golden from the reference
# (it does not exist there): its anchor is that with the flag off the same code path is the pinned float32 oracle.
_OPERAND_BF16 = False


@contextlib.contextmanager
def operand_precision(dtype):
    """``with operand_precision('bf16'):`` runs the oracle with bf16-rounded product operands."""
    global _OPERAND_BF16
    old = _OPERAND_BF16
    _OPERAND_BF16 = dtype in ('bf16', torch.bfloat16)
    try:
        yield
    finally:
        _OPERAND_BF16 = old


def round_bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundOperand(torch.autograd.Function):
    """Forward: round to bfloat16; backward: identity (the stored tensors are float32, only the product sees bf16)."""

    @staticmethod
    def forward(ctx, t):
        return round_bf16(t)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradient(torch.autograd.Function):
    """Forward: identity; backward: the gradient entering the two backward products is rounded to bfloat16."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return round_bf16(g)


def _product(op, x, w, b, **kw):
    """op(x, w) + b with the mode's operand precision; the bias joins after the product (its gradient is the float32
    sum of the unrounded output gradient)."""
    if not _OPERAND_BF16:
        return op(x, w, b, **kw)
    y = _RoundGradient.apply(op(_RoundOperand.apply(x), _RoundOperand.apply(w), None, **kw))
    if b is None:
        return y
    return y + (b.view(1, -1, 1, 1) if y.dim() == 4 else b)


def conv_block(x, sd, row, scope, training, new_buffers=None, taps=None, act_masks=None, tag=''):
    """layer.Conv2D (model/layer.py:10-26): Conv2d -> LeakyReLU(0.1) -> BatchNorm2d (BN after the activation).
    ``tag`` distinguishes the taps / activation masks of repeated applications (stacked spectrogram channels)."""
    name, _, _, k, s, p, has_bn = row
    w, b = _find(sd, name + 'conv.weight', scope), _find(sd, name + 'conv.bias', scope)
    a = _leaky(_product(F.conv2d, x, w, b, stride=s, padding=p), name + tag, act_masks)
    if taps is not None:
        taps[name + tag + '_act'] = a
    if has_bn:
        a = _bn_train_or_eval(a, sd, name, scope, training, new_buffers)
    if taps is not None:
        taps[name + tag] = a
    return a


def tconv_block(x, sd, row, scope, training, new_buffers=None, taps=None, act_masks=None, tag=''):
    """layer.TConv2D (model/layer.py:29-46): ConvTranspose2d(output_padding) -> LeakyReLU(0.1) -> BatchNorm2d."""
    name, _, _, k, s, p, op, has_bn = row
    w, b = _find(sd, name + 'tconv.weight', scope), _find(sd, name + 'tconv.bias', scope)
    a = _leaky(_product(F.conv_transpose2d, x, w, b, stride=s, padding=p, output_padding=op), name + tag, act_masks)
    if taps is not None:
        taps[name + tag + '_act'] = a
    if has_bn:
        a = _bn_train_or_eval(a, sd, name, scope, training, new_buffers)
    if taps is not None:
        taps[name + tag] = a
    return a


def encoder_forward(sd, x, arch, dim_z, training, dropout_mask=None, new_buffers=None, taps=None, act_masks=None):
    """SpectrogramEncoder.forward (model/encoder.py:95-108), single-channel spectrograms.
    ``dropout_mask`` = nn.Dropout keep-mask already scaled by 1/(1-p) (encoder.py:85), ``None`` = no dropout."""
    enc_rows, _, _ = arch_tables(arch)
    n_ch = x.shape[1]
    if n_ch == 1:
        h = x
        for row in enc_rows:
            h = conv_block(h, sd, row, 'encoder.', training, new_buffers, taps, act_masks)
    else:
        # stacked spectrograms (encoder.py:50-70, 99-104): the shared per-channel stack once per input channel, channel-
        # concatenated, then the features mixer - the 1x1 conv alone (deepest_features_mix, 512*C -> 1024) or the 4x4
        # conv + the 1x1 conv (256*C -> 768 -> 1024, the reference's config default)
        assert arch == 'speccnn8l1_bn'
        deepest = _find(sd, 'enc7conv.weight', 'encoder.').shape[1] == 256
        n_single = len(enc_rows) - (1 if deepest else 2)
        outs = []
        for ch in range(n_ch):
            h = x[:, ch:ch + 1]
            for row in enc_rows[:n_single]:
                h = conv_block(h, sd, row, 'encoder.', training, new_buffers, taps, act_masks, tag='' if ch == 0 else f'#{ch}')
            outs.append(h)
        h = torch.cat(outs, dim=1)
        for row in enc_rows[n_single:]:
            h = conv_block(h, sd, row, 'encoder.', training, new_buffers, taps, act_masks)
    h = h.reshape(x.shape[0], -1)                                      # encoder.py:104
    if training and dropout_mask is not None:
        h = h * dropout_mask.reshape(h.shape)
    z = _product(F.linear, h, sd['encoder.mlp.1.weight'], sd['encoder.mlp.1.bias'])   # encoder.py:85
    if 'encoder.mlp.lat_in_regularization.weight' in sd:                # output_bn, encoder.py:86-87
        lsd = {'latbn.weight': sd['encoder.mlp.lat_in_regularization.weight'],
               'latbn.bias': sd['encoder.mlp.lat_in_regularization.bias'],
               'latbn.running_mean': sd['encoder.mlp.lat_in_regularization.running_mean'],
               'latbn.running_var': sd['encoder.mlp.lat_in_regularization.running_var']}
        nb = {} if new_buffers is not None else None
        z = _bn_train_or_eval(z, lsd, 'lat', 'lat', training, nb)
        if nb:
            new_buffers['encoder.mlp.lat_in_regularization.running_mean'] = nb['latbn.running_mean']
            new_buffers['encoder.mlp.lat_in_regularization.running_var'] = nb['latbn.running_var']
    return z.reshape(x.shape[0], 2, dim_z)                              # encoder.py:108


def decoder_forward(sd, z, arch, training, dropout_mask=None, new_buffers=None, taps=None, act_masks=None):
    """SpectrogramDecoder.forward (model/decoder.py:83-92) + SpectrogramCNN (decoder.py:199-220)."""
    _, dec_rows, cnn_in = arch_tables(arch)
    h = _product(F.linear, z, sd['decoder.mlp.0.weight'], sd['decoder.mlp.0.bias'])   # decoder.py:64
    if training and dropout_mask is not None:                            # decoder.py:65
        h = h * dropout_mask.reshape(h.shape)
    h = h.view(-1, *cnn_in)                                              # decoder.py:85-86
    n_last = len(dec_rows) - (1 if arch == 'speccnn8l1_bn' else 0)      # index of ConvTranspose2d in dec_nn
    w = sd[f'decoder.single_ch_cnn.dec_nn.{n_last}.weight']
    b = sd[f'decoder.single_ch_cnn.dec_nn.{n_last}.bias']

    def tail(h, tag):
        y = _product(F.conv_transpose2d, h, w, b, stride=2, padding=2)   # decoder.py:218
        if taps is not None:
            taps['dec8' + tag + '_pre'] = y
        if act_masks is not None and 'dec8' + tag in act_masks:          # pinned Hardtanh gate (True = pass-through)
            return torch.where(act_masks['dec8' + tag], y, y.detach().clamp(-1.0, 1.0))
        return F.hardtanh(y)                                             # decoder.py:98,219

    n_ch, width = 1, 512
    if arch == 'speccnn8l1_bn':
        # un-mixer: 2048 -> C * last_4x4conv_ch (decoder.py:72-75; 512, or 1800 under force_bigger_network: decoder.py:70),
        # split into chunks of that width (decoder.py:85-86) = the input channels of the first 4x4 block
        width = _find(sd, 'dec2tconv.weight', 'decoder.').shape[0]
        n_ch = _find(sd, 'dec1tconv.weight', 'decoder.').shape[1] // width
    if n_ch == 1:
        for row in dec_rows:
            h = tconv_block(h, sd, row, 'decoder.', training, new_buffers, taps, act_masks)
        return tail(h, '')
    # stacked spectrograms (decoder.py:85-92): un-mix, split along channels, the shared stack once per chunk
    h = tconv_block(h, sd, dec_rows[0], 'decoder.', training, new_buffers, taps, act_masks)
    outs = []
    for ch, chunk in enumerate(torch.split(h, width, dim=1)):
        tag = '' if ch == 0 else f'#{ch}'
        for row in dec_rows[1:]:
            chunk = tconv_block(chunk, sd, row, 'decoder.', training, new_buffers, taps, act_masks, tag=tag)
        outs.append(tail(chunk, tag))
    return torch.cat(outs, dim=1)


def reparametrize(z_mu_logvar, eps, training):
    """BasicVAE.forward sampling (model/VAE.py:49-58)."""
    mu = z_mu_logvar[:, 0, :]
    sigma = torch.exp(z_mu_logvar[:, 1, :] / 2.0)
    return mu + sigma * eps if training else mu


def gaussian_dkl(mu, logvar, normalize=True):
    """loss.GaussianDkl.__call__ (model/loss.py:57-66)."""
    dkl = 0.5 * torch.sum(torch.exp(logvar) + torch.square(mu) - logvar - 1.0)
    dkl = dkl / mu.size(0)
    return dkl / mu.size(1) if normalize else dkl


def standard_gaussian_log_probability(samples):
    """utils/probability.py:13-18."""
    return -0.5 * (samples.shape[1] * np.log(2 * np.pi) + torch.sum(samples ** 2, dim=1))


def gaussian_log_probability(samples, mu, log_var):
    """utils/probability.py:21-29."""
    return -0.5 * (samples.shape[1] * np.log(2 * np.pi)
                   + torch.sum(log_var + ((samples - mu) ** 2 / torch.exp(log_var)), dim=1))


def flow_latent_loss(z_0_mu_logvar, z_0, z_K, log_abs_det_jac, normalize=False):
    """FlowVAE.latent_loss (model/VAE.py:183-193) for given flow outputs."""
    log_q = gaussian_log_probability(z_0, z_0_mu_logvar[:, 0, :], z_0_mu_logvar[:, 1, :])
    loss = -(standard_gaussian_log_probability(z_K) - log_q + log_abs_det_jac).mean()
    return loss / z_0.shape[1] if normalize else loss


def l2_loss(inferred, target, contents_average=False, batch_average=True):
    """loss.L2Loss.__call__ (model/loss.py:37-43)."""
    loss = torch.sum(torch.square(inferred - target))
    if batch_average:
        loss = loss / inferred.shape[0]
    if contents_average:
        loss = loss / inferred[0, :].numel()
    return loss


def vae_forward(sd, x, arch, dim_z, training, eps=None, enc_mask=None, dec_mask=None, new_buffers=None, taps=None,
                act_masks=None):
    """BasicVAE.forward (model/VAE.py:37-61) -> (z_mu_logvar, z, z, zeros[B,1], x_out)."""
    zml = encoder_forward(sd, x, arch, dim_z, training, enc_mask, new_buffers, taps, act_masks)
    z = reparametrize(zml, eps, training)
    x_out = decoder_forward(sd, z, arch, training, dec_mask, new_buffers, taps, act_masks)
    return zml, z, z, torch.zeros((x.shape[0], 1), dtype=x.dtype), x_out


def adam_update(p, g, m, v, t, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam, torch-1.7 semantics, coupled L2 (train.py:166-167)."""
    b1, b2 = betas
    g = g + weight_decay * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


def is_parameter_key(k):
    return not (k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))


def train_step(sd, x, arch, dim_z, eps, enc_mask=None, dec_mask=None, beta=0.2, normalize_losses=True, lr=2e-4,
               betas=(0.9, 0.999), weight_decay=1e-4, adam_state=None, step=1, taps=None, act_masks=None, reg=None):
    """One minibatch of train.py:203-248: forward, MSE + beta*Dkl (+ controls loss), backward, Adam.

    ``reg`` (optional) adds the preset-regression network of train.py:220,238-246: dict(sd={'reg_model...': tensor},
    v_in=[B, L] targets, masks=[two keep/(1-p) dropout masks] or None, criterion=optional controls loss); its
    parameters then appear in ``grads`` /
    ``new_sd`` / ``adam_state`` under 'reg.<key>' and the result carries 'controls' and 'v_out'.

    Returns dict(losses, outputs, grads{key}, new_sd{key}, adam_state)."""
    sd = dict(sd)
    if reg is not None:
        sd.update({'reg.' + k: v for k, v in reg['sd'].items()})
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_parameter_key(k)}
    full = dict(sd)
    full.update(params)
    new_buffers = {}
    zml, z, _, _, x_out = vae_forward(full, x, arch, dim_z, True, eps, enc_mask, dec_mask, new_buffers, taps, act_masks)
    if normalize_losses:
        recons = F.mse_loss(x_out, x, reduction='mean')                 # train.py:103-104,222
    else:
        recons = l2_loss(x_out, x)                                      # train.py:105-106
    lat = gaussian_dkl(zml[:, 0, :], zml[:, 1, :], normalize=normalize_losses)   # train.py:225
    total = recons + lat * beta                                         # train.py:227,246
    cont = v_out = None
    if reg is not None:
        rsd = {k[len('reg.'):]: v for k, v in full.items() if k.startswith('reg.')}
        rbuf = {}
        v_out = mlp_regression_forward(rsd, z, True, reg.get('masks'), rbuf)      # train.py:220
        # train.py:238-239; default = the numeric branch, ``reg['criterion']`` = any callable(v_out, v_in), e.g. the
        # full SynthParamsLoss restatement of oracle/params_oracle.py (train.py:111-116)
        cont = reg.get('criterion', numeric_params_loss)(v_out, reg['v_in'])
        total = total + cont                                                       # train.py:246
        new_buffers.update({'reg.' + k: v for k, v in rbuf.items()})
    keys = list(params.keys())
    grads = torch.autograd.grad(total, [params[k] for k in keys])
    grads = dict(zip(keys, grads))
    if adam_state is None:
        adam_state = {k: (torch.zeros_like(params[k]), torch.zeros_like(params[k])) for k in keys}
    new_sd = {k: v for k, v in sd.items()}
    new_state = {}
    for k in keys:
        m, v = adam_state[k]
        p_new, m_new, v_new = adam_update(params[k].detach(), grads[k], m, v, step, lr, betas, 1e-8, weight_decay)
        new_sd[k] = p_new
        new_state[k] = (m_new, v_new)
    for k_suffix, val in new_buffers.items():
        hits = [k for k in sd if k.endswith(k_suffix)]
        assert len(hits) == 1, (k_suffix, hits)
        new_sd[hits[0]] = val
    res = {'recons': recons.detach(), 'latent': lat.detach(), 'total': total.detach(), 'z_mu_logvar': zml.detach(),
           'z': z.detach(), 'x_out': x_out.detach(), 'grads': grads, 'new_sd': new_sd, 'adam_state': new_state}
    if reg is not None:
        res['controls'], res['v_out'] = cont.detach(), v_out.detach()
    return res


def closed_form_state_dict(template, seed=1234, dtype=torch.float32):
    """Deterministic pseudo-random weights without RNG streams or weight files: element i of tensor #idx is the
    classic hash  u_i = frac(sin(i*12.9898 + (seed+idx)*78.233) * 43758.5453123)  mapped to [-1, 1), evaluated in
    float64, scaled like torch's default init (uniform +-sqrt(3/fan_in)).  BN gamma 1 +- 0.1, beta +-0.1,
    running_mean +-0.1, running_var 1 +- 0.2, biases +-0.05.  (Smooth closed forms such as sin(i*a) give a degenerate,
    ill-conditioned network whose float32 evaluation is 5-10 % away from float64; this hash behaves like the
    reference's random init: ~2e-6 on activations, ~1e-3 on gradients.)
    ``template``: {key: shape} in the reference's registration order.  Used by tests/golden/make_goldens.py."""
    sd = {}
    for idx, (k, shape) in enumerate(template.items()):
        n = 1
        for s in shape:
            n *= s
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros(shape, dtype=torch.long)
            continue
        i = torch.arange(n, dtype=torch.float64)
        u = torch.frac(torch.sin(i * 12.9898 + (seed + idx) * 78.233) * 43758.5453123).abs()
        base = 2.0 * u - 1.0
        if k.endswith('running_var'):
            val = 1.0 + 0.2 * base
        elif k.endswith('running_mean'):
            val = 0.1 * base
        elif 'bn.weight' in k or 'lat_in_regularization.weight' in k:
            val = 1.0 + 0.1 * base
        elif 'bn.bias' in k or 'lat_in_regularization.bias' in k:
            val = 0.1 * base
        elif k.endswith('.bias'):
            val = 0.05 * base
        else:
            fan_in = n // shape[0] if len(shape) > 1 else n
            if 'tconv.weight' in k or (k.startswith('decoder.single_ch_cnn.dec_nn') and len(shape) == 4):
                # ConvTranspose2d weight is [Cin, Cout, kh, kw]: each output pixel sees Cin * (k/stride)^2 taps
                fan_in = shape[0] * shape[2] * shape[3] // 4 if shape[2] > 1 else shape[0]
            val = base * math.sqrt(3.0 / max(1, fan_in))
        sd[k] = val.reshape(shape).to(dtype)
    return sd


def mlp_regression_forward(sd, z, training, masks=None, new_buffers=None):
    """regression.MLPRegression('3l1024').forward (model/regression.py:76-102): fc -> [BN1d -> Dropout] -> ReLU, the
    last two fc layers without BN/Dropout, then PresetActivation = Hardtanh(0,1) on every output
    (regression.py:20-53 with cat_softmax_activation=False).  ``masks``: injected keep/(1-p) dropout masks."""
    n_fc = len([k for k in sd if k.startswith('reg_model.fc') and k.endswith('.weight')])
    h = z
    for l in range(1, n_fc):
        h = _product(F.linear, h, sd[f'reg_model.fc{l}.weight'], sd[f'reg_model.fc{l}.bias'])
        if f'reg_model.bn{l}.weight' in sd:
            bsd = {'rbn.weight': sd[f'reg_model.bn{l}.weight'], 'rbn.bias': sd[f'reg_model.bn{l}.bias'],
                   'rbn.running_mean': sd[f'reg_model.bn{l}.running_mean'],
                   'rbn.running_var': sd[f'reg_model.bn{l}.running_var']}
            nb = {} if new_buffers is not None else None
            h = _bn_train_or_eval(h, bsd, 'r', 'r', training, nb)
            if nb:
                new_buffers[f'reg_model.bn{l}.running_mean'] = nb['rbn.running_mean']
                new_buffers[f'reg_model.bn{l}.running_var'] = nb['rbn.running_var']
            if training and masks is not None:
                h = h * masks[l - 1]
        h = F.relu(h)
    h = _product(F.linear, h, sd[f'reg_model.fc{n_fc}.weight'], sd[f'reg_model.fc{n_fc}.bias'])
    return F.hardtanh(h, 0.0, 1.0)


def numeric_params_loss(v_out, v_in):
    """Numeric branch of loss.SynthParamsLoss with normalize_losses=True (model/loss.py:107-108,136):
    nn.MSELoss('mean') over the numerical columns (all 144 in the all-numerical representation)."""
    return F.mse_loss(v_out, v_in, reduction='mean')
