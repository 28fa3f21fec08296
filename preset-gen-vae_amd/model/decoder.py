"""Spectrogram decoder, MI355X-native (surface of the reference's ``model/decoder.py``).

``SpectrogramDecoder(architecture, dim_z, output_tensor_size, fc_dropout, force_bigger_network)`` ->
``forward(z[B,dim_z]) -> [B,C,257,347]`` (reference decoder.py:9-92).  State-dict keys follow the reference
(``mlp.0``, ``features_unmixer_cnn.dec1tconv``, ``single_ch_cnn.dec_nn.N.decKtconv``, ``single_ch_cnn.dec_nn.6``).

* ``speccnn8l1_bn``: Linear(dz -> 2048*3*4) -> Dropout -> 1x1 TConv un-mixer (dec1) -> dec2..dec7 with the per-axis
  ``output_padding`` of decoder.py:205-217 -> ConvTranspose2d(8,1,5,2,2) -> Hardtanh (decoder.py:218-219);
* ``speccnn4l1_bn``: Linear(dz -> 64*17*23) -> Dropout -> dec5, dec6, dec7 (decoder.py:212-217) -> same output layer
  (SURVEY.md §8 note N1).
"""
import numpy as np
import torch
import torch.nn as nn

from . import layer


def _lrelu():
    return nn.LeakyReLU(0.1)


class SpectrogramCNN(nn.Module):
    """Single-channel transposed-conv stack (reference decoder.py:95-274, table at :199-220)."""

    def __init__(self, architecture, spectrogram_input_size, output_activation=None, append_1x1_conv=False,
                 force_bigger_network=False):
        super().__init__()
        self.architecture = architecture
        self.spectrogram_input_size = spectrogram_input_size
        if architecture not in ('speccnn8l1_bn', 'speccnn4l1_bn'):
            raise NotImplementedError("Architecture '{}' not available".format(architecture))
        assert not append_1x1_conv  # the reference asserts False on this path too (decoder.py:221-222)
        output_activation = nn.Hardtanh() if output_activation is None else output_activation
        # (in_ch, out_ch, output_padding) rows of decoder.py:205-217; the 4-layer variant keeps the last three.
        table = [((512 if not force_bigger_network else 1800), 256, [1, 1], 'dec2'), (256, 128, [1, 0], 'dec3'),
                 (128, 64, [1, 1], 'dec4'), (64, 32, [1, 1], 'dec5'), (32, 16, [1, 0], 'dec6'),
                 (16, 8, [1, 0], 'dec7')]
        if architecture == 'speccnn4l1_bn':
            table = table[3:]
        mods = [layer.TConv2D(ci, co, [4, 4], [2, 2], 2, output_padding=op, activation=_lrelu(), name_prefix=name)
                for ci, co, op, name in table]
        mods.append(nn.ConvTranspose2d(8, 1, [5, 5], [2, 2], 2))
        mods.append(output_activation)
        self.dec_nn = nn.Sequential(*mods)
        self._last_block = layer._Block(mods[-2], output_activation, None)

    def pgv_blocks(self):
        blocks = []
        for m in self.dec_nn:
            if isinstance(m, layer._ConvBlockBase):
                blocks += m.pgv_blocks()
        return blocks + [self._last_block]

    def forward(self, x_spectrogram):
        return layer.run_stack(x_spectrogram, self.pgv_blocks(), self.training)


class SpectrogramDecoder(nn.Module):
    """(Linear -> Dropout) + transposed-conv stack (reference decoder.py:9-92)."""

    def __init__(self, architecture, dim_z, output_tensor_size, fc_dropout, force_bigger_network=False):
        super().__init__()
        self.output_tensor_size = output_tensor_size
        self.spectrogram_input_size = (output_tensor_size[2], output_tensor_size[3])
        self.spectrogram_channels = output_tensor_size[1]
        self.dim_z = dim_z
        self.architecture = architecture
        self.mixer_1x1conv_ch = 2048
        self.last_4x4conv_ch = (512 if not force_bigger_network else 1800)
        self.fc_dropout = fc_dropout
        self._rng = None   # dropout mask source: the enclosing VAE's generator, else a local one (rng.device_rng)
        if architecture not in ('speccnn8l1_bn', 'speccnn4l1_bn'):
            raise NotImplementedError("Only speccnn8l1_bn / speccnn4l1_bn are available")
        if self.spectrogram_channels != 1 and architecture != 'speccnn8l1_bn':
            raise NotImplementedError("stacked multi-channel spectrograms need the speccnn8l1_bn un-mixer "
                                      "(decoder.py:35-37)")
        if self.spectrogram_input_size != (257, 347):
            raise NotImplementedError("decoder bottleneck is defined for 257x347 spectrograms only (decoder.py:58-67)")
        if architecture == 'speccnn8l1_bn':
            self.cnn_input_shape = (self.mixer_1x1conv_ch, 3, 4)
        else:
            self.cnn_input_shape = (64, 17, 23)
        self.mlp = nn.Sequential(nn.Linear(self.dim_z, int(np.prod(self.cnn_input_shape))),
                                 nn.Dropout(self.fc_dropout))
        if architecture == 'speccnn8l1_bn':
            self.features_unmixer_cnn = layer.TConv2D(self.mixer_1x1conv_ch,
                                                      self.spectrogram_channels * self.last_4x4conv_ch,
                                                      [1, 1], [1, 1], 0, activation=_lrelu(), name_prefix='dec1')
        else:
            self.features_unmixer_cnn = nn.Sequential()
        single_spec_size = list(self.spectrogram_input_size)
        single_spec_size[1] = 1
        self.single_ch_cnn = SpectrogramCNN(self.architecture, single_spec_size, append_1x1_conv=False,
                                            force_bigger_network=force_bigger_network)
        if self.spectrogram_channels > 1:
            # applied once per spectrogram channel (decoder.py:89-92): gradients are summed by autograd
            for p_ in self.single_ch_cnn.parameters():
                p_._pgv_shared = True

    def _all_blocks(self):
        blocks = []
        if isinstance(self.features_unmixer_cnn, layer._ConvBlockBase):
            blocks += self.features_unmixer_cnn.pgv_blocks()
        return blocks + self.single_ch_cnn.pgv_blocks()

    def forward(self, z_sampled, dropout_mask=None, sq_target=None, sq_scale=None):
        """``sq_target`` / ``sq_scale`` (single-channel spectrograms only): also return the squared-error reconstruction
        term, evaluated inside the output stack (layer.ConvStackFn)."""
        lin = self.mlp[0]
        if self.training and self.fc_dropout > 0.0 and dropout_mask is None:
            # Linear + Dropout as one function: the Dropout backward pass also sums the bias gradient
            from ..rng import STREAM_DEC_DROPOUT, device_rng
            mixed = layer.LinearFn.apply(z_sampled, lin.weight, lin.bias,
                                         (device_rng(self, z_sampled.device), self.fc_dropout, STREAM_DEC_DROPOUT))
        else:
            mixed = layer.LinearFn.apply(z_sampled, lin.weight, lin.bias)
            if self.training and self.fc_dropout > 0.0:
                mixed = layer.MaskMulFn.apply(mixed, dropout_mask.reshape(-1))
        mixed = mixed.view(-1, self.cnn_input_shape[0], self.cnn_input_shape[1], self.cnn_input_shape[2])
        if self.spectrogram_channels == 1:
            return layer.run_stack(mixed, self._all_blocks(), self.training, sq_target, sq_scale)
        if sq_target is not None:
            raise NotImplementedError("fused reconstruction criterion: single-channel spectrograms only")
        # stacked spectrograms (decoder.py:85-92): un-mix, split along channels, the shared stack once per chunk
        unmixed = layer.run_stack(mixed, self.features_unmixer_cnn.pgv_blocks(), self.training)
        single = self.single_ch_cnn.pgv_blocks()
        outs = [layer.run_stack(chunk.contiguous(), single, self.training)
                for chunk in torch.split(unmixed, self.last_4x4conv_ch, dim=1)]
        return torch.cat(outs, dim=1)
