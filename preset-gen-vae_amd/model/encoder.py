"""Spectrogram encoder, MI355X-native (surface of the reference's ``model/encoder.py``).

``SpectrogramEncoder(architecture, dim_z, input_tensor_size, fc_dropout, output_bn, deepest_features_mix,
force_bigger_network)`` -> ``forward(x[B,C,257,347]) -> [B,2,dim_z]`` (reference encoder.py:23-108).  Module tree and
state-dict keys follow the reference (``single_ch_cnn.enc_nn.N.encKconv``, ``features_mixer_cnn``, ``mlp.1``,
``mlp.lat_in_regularization``).  Architectures:

* ``speccnn8l1_bn`` — the reference's only fully supported stack (encoder.py:233-259 + mixer :54-70);
* ``speccnn4l1_bn`` — BASELINE.json's "4-layer conv-VAE": the first four blocks enc1..enc4 of the same table
  (encoder.py:241-248) followed by ``Dropout -> Linear(64*17*23 -> 2*dim_z)`` (SURVEY.md §8 note N1).

The whole CNN runs as one :class:`layer.ConvStackFn` (BatchNorm folded between blocks); Dropout/Linear/BatchNorm1d
run on the HIP GEMM / element-wise kernels.
"""
import torch
import torch.nn as nn

from .. import ops
from . import layer


def available_architectures():
    return ['speccnn8l1_bn', 'speccnn4l1_bn']


def _lrelu():
    return nn.LeakyReLU(0.1)


class SpectrogramCNN(nn.Module):
    """Per-channel conv stack (reference encoder.py:111-306, ``speccnn8l1_bn`` table at :233-259)."""

    def __init__(self, architecture, last_layers_to_remove=0):
        super().__init__()
        self.architecture = architecture
        if architecture not in ('speccnn8l1_bn', 'speccnn4l1_bn'):
            raise NotImplementedError("Architecture '{}' not available".format(architecture))
        chans = [1, 8, 16, 32, 64, 128, 256]
        n_plain = 6 if architecture == 'speccnn8l1_bn' else 4
        if architecture == 'speccnn4l1_bn':
            assert last_layers_to_remove == 0
        mods = []
        for i in range(n_plain):
            k = [5, 5] if i == 0 else [4, 4]
            mods.append(layer.Conv2D(chans[i], chans[i + 1], k, [2, 2], 2, [1, 1], activation=_lrelu(),
                                     name_prefix='enc{}'.format(i + 1), batch_norm=(None if i == 0 else 'after')))
        self.enc_nn = nn.Sequential(*mods)
        if architecture == 'speccnn8l1_bn':
            if last_layers_to_remove <= 1:
                self.enc_nn.add_module('4x4conv', layer.Conv2D(256, 512, [4, 4], [2, 2], 2, [1, 1],
                                                               activation=_lrelu(), name_prefix='enc7'))
            if last_layers_to_remove == 0:
                self.enc_nn.add_module('1x1conv', layer.Conv2D(512, 1024, [1, 1], [1, 1], 0, [1, 1], batch_norm=None,
                                                               activation=_lrelu(), name_prefix='enc8'))

    def pgv_blocks(self):
        blocks = []
        for m in self.enc_nn:
            blocks += m.pgv_blocks()
        return blocks

    def forward(self, x_spectrogram):
        return layer.run_stack(x_spectrogram, self.pgv_blocks(), self.training)


class SpectrogramEncoder(nn.Module):
    """CNN + (Dropout -> Linear [-> BatchNorm1d]) producing mu and log(var) (reference encoder.py:23-108)."""

    def __init__(self, architecture, dim_z, input_tensor_size, fc_dropout, output_bn=False,
                 deepest_features_mix=True, force_bigger_network=False):
        super().__init__()
        self.dim_z = dim_z
        self.spectrogram_channels = input_tensor_size[1]
        self.architecture = architecture
        self.deepest_features_mix = deepest_features_mix
        self.fc_dropout = fc_dropout
        if architecture not in available_architectures():
            # the reference asserts the same restriction (encoder.py:53)
            raise NotImplementedError("Architecture '{}' not available".format(architecture))
        n_ch = self.spectrogram_channels
        if n_ch != 1 and architecture != 'speccnn8l1_bn':
            raise NotImplementedError("stacked multi-channel spectrograms need the speccnn8l1_bn mixer (encoder.py:53)")
        # 2048 if single-ch, 1024 if multi-channel mixer (encoder.py:46-47)
        self.mixer_1x1conv_ch = 1024 if n_ch > 1 else 2048
        if architecture == 'speccnn8l1_bn':
            self.single_ch_cnn = SpectrogramCNN(architecture, last_layers_to_remove=(1 if deepest_features_mix else 2))
            self.features_mixer_cnn = nn.Sequential()
            if deepest_features_mix:
                self.features_mixer_cnn = layer.Conv2D(512 * n_ch, self.mixer_1x1conv_ch, [1, 1], [1, 1], 0, [1, 1],
                                                       activation=_lrelu(), name_prefix='enc8', batch_norm=None)
            else:
                if not force_bigger_network:
                    n_4x4_ch = 512 if n_ch == 1 else 768      # encoder.py:62-63
                else:
                    n_4x4_ch = 1800
                self.features_mixer_cnn = nn.Sequential(
                    layer.Conv2D(256 * n_ch, n_4x4_ch, [4, 4], [2, 2], 2, [1, 1], activation=_lrelu(),
                                 name_prefix='enc7'),
                    layer.Conv2D(n_4x4_ch, self.mixer_1x1conv_ch, [1, 1], [1, 1], 0, [1, 1], activation=_lrelu(),
                                 name_prefix='enc8', batch_norm=None))
        else:
            self.single_ch_cnn = SpectrogramCNN(architecture)
            self.features_mixer_cnn = nn.Sequential()
        if n_ch > 1:
            # the per-channel stack is applied n_ch times per forward (encoder.py:101-102): its parameter gradients are
            # the SUM over the applications, so they go through autograd's accumulation instead of being written
            # straight into the flat gradient buffer
            for p_ in self.single_ch_cnn.parameters():
                p_._pgv_shared = True
        # CNN output size by shape arithmetic (the reference runs a dummy forward, encoder.py:73-78)
        C, H, W = 1, input_tensor_size[2], input_tensor_size[3]
        for blk in self.single_ch_cnn.pgv_blocks():
            g = blk.geom(H, W)
            C, H, W = blk.c_out, g.Hs, g.Ws
        for blk in self._mixer_blocks():
            g = blk.geom(H, W)
            C, H, W = blk.c_out, g.Hs, g.Ws
        self.cnn_out_size = torch.Size((1, C, H, W))
        cnn_out_items = C * H * W
        self.mlp = nn.Sequential(nn.Dropout(self.fc_dropout), nn.Linear(cnn_out_items, 2 * self.dim_z))
        if output_bn:
            self.mlp.add_module('lat_in_regularization', nn.BatchNorm1d(2 * self.dim_z))
        self.output_bn = output_bn
        # dropout mask source: None -> on-device Philox stream owned by the enclosing VAE (or a local one)
        self._rng = None

    def _mixer_blocks(self):
        mixer = self.features_mixer_cnn
        if isinstance(mixer, layer._ConvBlockBase):
            return mixer.pgv_blocks()
        blocks = []
        for m in mixer:
            blocks += m.pgv_blocks()
        return blocks

    def _all_blocks(self):
        """Single-channel input: per-channel stack and mixer run as ONE fused stack (BatchNorm folded across the seam)."""
        return self.single_ch_cnn.pgv_blocks() + self._mixer_blocks()

    def _forward_cnns(self, x_spectrograms, out_dropout=None):
        if self.spectrogram_channels == 1:
            return layer.run_stack(x_spectrograms, self._all_blocks(), self.training, out_dropout=out_dropout)
        # stacked spectrograms (encoder.py:99-104): the shared per-channel stack once per input channel - each call has
        # its own BatchNorm batch statistics and running-stat update, as in the reference - then the features mixer
        single = self.single_ch_cnn.pgv_blocks()
        outs = [layer.run_stack(x_spectrograms[:, ch:ch + 1, :, :].contiguous(), single, self.training)
                for ch in range(self.spectrogram_channels)]
        return layer.run_stack(torch.cat(outs, dim=1), self._mixer_blocks(), self.training, out_dropout=out_dropout)

    def forward(self, x_spectrograms, dropout_mask=None, reparam=None):
        """``dropout_mask`` (optional, [B, features], already scaled by 1/(1-p)) injects the Dropout mask for parity
        runs; by default it is drawn on device.  ``reparam`` = (rng, kl_scale, kl_buf) (training, from BasicVAE.forward):
        when this encoder ends in its BatchNorm1d, the reparameterisation and the Dkl term are evaluated by the same
        launch and (z_mu_logvar, z_sampled, Dkl) is returned instead of z_mu_logvar alone."""
        n_minibatch = x_spectrograms.size()[0]
        if self.training and self.fc_dropout > 0.0 and dropout_mask is None:
            # the Dropout mask is drawn and applied by the pass that applies the last conv block's BatchNorm
            from ..rng import STREAM_ENC_DROPOUT, device_rng
            drop = (device_rng(self, x_spectrograms.device), self.fc_dropout, STREAM_ENC_DROPOUT)
            cnn_out = self._forward_cnns(x_spectrograms, out_dropout=drop).view(n_minibatch, -1)
        else:
            cnn_out = self._forward_cnns(x_spectrograms).view(n_minibatch, -1)
            if self.training and self.fc_dropout > 0.0:
                cnn_out = layer.MaskMulFn.apply(cnn_out, dropout_mask.reshape(-1))
        lin = self.mlp[1]
        if reparam is not None and self.output_bn and self.training:
            bn = self.mlp.lat_in_regularization
            rng, kl_scale, kl_buf = reparam
            lin_out = layer.LinearFn.apply(cnn_out, lin.weight, lin.bias, None, True)
            y, z, kl = layer.EncoderHeadFn.apply(lin_out, bn, bn.weight, bn.bias, rng, kl_scale, kl_buf, lin.bias)
            return torch.reshape(y, (n_minibatch, 2, self.dim_z)), z, kl
        z_mu_logvar = layer.LinearFn.apply(cnn_out, lin.weight, lin.bias)
        if self.output_bn:
            bn = self.mlp.lat_in_regularization
            z_mu_logvar = layer.BatchNorm1dFn.apply(z_mu_logvar, bn, self.training, bn.weight, bn.bias)
        return torch.reshape(z_mu_logvar, (n_minibatch, 2, self.dim_z))
