"""Configuration bags for the hot path (surface of the reference's ``config.py`` / ``utils/config.py``).

Only the fields the builders and the train step read are present (SURVEY.md §8b lists them with the reference line
numbers); values are the reference defaults except where BASELINE.json's benchmark configuration says otherwise
(``latent_flow_arch=None`` -> BasicVAE, ``params_regression_architecture='mlp_3l1024'``).  Like the reference this is a
module of plain attribute bags that callers mutate, followed by ``update_dynamic_config_params()``.
"""


class _Config(object):
    pass


model = _Config()
model.name = "BasicVAE"
model.run_name = '00_debug'                    # config.py:21
model.logs_root_dir = "saved"                  # config.py:75
model.increased_dataset_size = None            # config.py:41, see update_dynamic_config_params()
model.encoder_architecture = 'speccnn8l1_bn'   # config.py:24 ; 'speccnn4l1_bn' = BASELINE "4-layer conv-VAE"
model.params_regression_architecture = 'mlp_3l1024'   # config.py:26 (flow_* variants are out of scope)
model.params_reg_softmax = False               # config.py:27
model.note_duration = (3.0, 1.0)               # config.py:29
model.sampling_rate = 22050                    # config.py:30
model.stft_args = (1024, 256)                  # config.py:31  (n_fft, hop)
model.mel_bins = 257                           # config.py:32
model.mel_f_limits = (0, 11050)                # config.py:33
model.midi_notes = ((60, 85), )                # config.py:35
model.stack_spectrograms = False               # config.py:37
model.stack_specs_deepest_features_mix = False  # config.py:38
model.spectrogram_min_dB = -120.0              # config.py:42
model.spectrogram_size = (257, 347)            # config.py:46
model.input_tensor_size = None                 # see update_dynamic_config_params()
model.concat_midi_to_z = None                  # see update_dynamic_config_params()
model.dim_z = 64                               # BASELINE.json configs (reference default 256, config.py:51)
model.latent_flow_arch = None                  # None -> BasicVAE (build.py:45-47)
model.forward_controls_loss = True             # config.py:58
model.learnable_params_tensor_length = 144     # all-numerical Dexed representation

train = _Config()
train.minibatch_size = 256                     # BASELINE.json metric (reference default 160, config.py:80)
train.latent_loss = 'Dkl'                      # config.py:90
train.latent_flow_input_regularization = 'bn'  # config.py:92 -> output_bn=True (build.py:25)
train.normalize_losses = True                  # config.py:98
train.optimizer = 'Adam'
train.test_holdout_proportion = 0.2            # config.py:82
train.lr_warmup_epochs = 6                     # config.py:107
train.lr_warmup_start_factor = 0.1             # config.py:108
train.scheduler_name = 'ReduceLROnPlateau'     # config.py:120
train.scheduler_lr_factor = 0.2                # config.py:123
train.scheduler_patience = 6                   # config.py:125
train.scheduler_cooldown = 6                   # config.py:126
train.scheduler_threshold = 1e-4               # config.py:127
train.initial_learning_rate = 2e-4             # config.py:105
train.adam_betas = (0.9, 0.999)                # config.py:109
train.weight_decay = 1e-4                      # config.py:110
train.fc_dropout = 0.3                         # config.py:111
train.reg_fc_dropout = 0.4                     # config.py:112
train.beta = 0.2                               # config.py:114
train.beta_start_value = 0.1                   # config.py:115
train.beta_warmup_epochs = 25                  # config.py:117
# not a reference field: how fp32 products are evaluated on the matrix cores (ops.set_fp32_products).  None = the library
# default ('bf16x6': six bf16 instructions on exact three-way operand splits, fp32 accumulation; what bench.py times);
# 'native' = v_mfma_f32_16x16x4_f32 everywhere.  Read by model.build.build_ae_model.
train.fp32_products = None


def update_dynamic_config_params():
    """Derived fields (reference config.py:148-202, the part the builders read)."""
    model.stack_spectrograms = model.stack_spectrograms and (len(model.midi_notes) > 1)
    model.concat_midi_to_z = (len(model.midi_notes) > 1) and not model.stack_spectrograms
    model.increased_dataset_size = (len(model.midi_notes) > 1) and not model.stack_spectrograms   # config.py:157
    model.input_tensor_size = (train.minibatch_size, 1 if not model.stack_spectrograms else len(model.midi_notes),
                               model.spectrogram_size[0], model.spectrogram_size[1])


update_dynamic_config_params()
