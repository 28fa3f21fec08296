"""Model builders — the operator surface ``train.py`` / ``eval.py`` call (reference ``model/build.py``).

Same function names, arguments, return tuples and config fields as the reference
(``build_encoder_and_decoder_models`` build.py:11-31, ``build_ae_model`` :34-52, ``build_extended_ae_model`` :55-80);
the returned modules run on the HIP kernels of ``include/pgv_hip.h``.
"""
from . import VAE, decoder, encoder, extendedAE, regression


def build_encoder_and_decoder_models(model_config, train_config):
    # Backward compatibility - recently added config args (build.py:13-14)
    if not hasattr(model_config, 'stack_specs_deepest_features_mix'):
        model_config.stack_specs_deepest_features_mix = True
    force_bigger_network = ((len(model_config.midi_notes) > 1) and not model_config.stack_spectrograms)
    enc_z_length = (model_config.dim_z - 2 if model_config.concat_midi_to_z else model_config.dim_z)
    encoder_model = encoder.SpectrogramEncoder(
        model_config.encoder_architecture, enc_z_length, model_config.input_tensor_size, train_config.fc_dropout,
        output_bn=(train_config.latent_flow_input_regularization.lower() == 'bn'),
        deepest_features_mix=model_config.stack_specs_deepest_features_mix,
        force_bigger_network=force_bigger_network)
    decoder_model = decoder.SpectrogramDecoder(
        model_config.encoder_architecture, model_config.dim_z, model_config.input_tensor_size,
        train_config.fc_dropout, force_bigger_network=force_bigger_network)
    return encoder_model, decoder_model


def build_ae_model(model_config, train_config):
    """:return: Tuple: encoder, decoder, full AE model"""
    encoder_model, decoder_model = build_encoder_and_decoder_models(model_config, train_config)
    if model_config.latent_flow_arch is None:
        ae_model = VAE.BasicVAE(encoder_model, model_config.dim_z, decoder_model, train_config.normalize_losses,
                                train_config.latent_loss)
    else:
        raise NotImplementedError("FlowVAE (nflows latent flows) is out of scope of the MI355X hot path; "
                                  "set model.latent_flow_arch = None")
    return encoder_model, decoder_model, ae_model


def build_extended_ae_model(model_config, train_config, idx_helper):
    """Spectral VAE + synth-parameters regression model, integrated into an ExtendedAE."""
    encoder_model, decoder_model, ae_model = build_ae_model(model_config, train_config)
    if not hasattr(model_config, 'params_reg_softmax'):
        model_config.params_reg_softmax = True  # legacy default (build.py:61-62)
    if model_config.params_regression_architecture.startswith("mlp_"):
        assert model_config.forward_controls_loss is True
        reg_arch = model_config.params_regression_architecture.replace("mlp_", "")
        reg_model = regression.MLPRegression(reg_arch, model_config.dim_z, idx_helper, train_config.reg_fc_dropout,
                                             cat_softmax_activation=model_config.params_reg_softmax)
    elif model_config.params_regression_architecture.startswith("flow_"):
        raise NotImplementedError("FlowRegression (nflows) is out of scope of the MI355X hot path")
    else:
        raise NotImplementedError("Synth param regression arch '{}' not implemented"
                                  .format(model_config.params_regression_architecture))
    extended_ae_model = extendedAE.ExtendedAE(ae_model, reg_model, idx_helper, train_config.fc_dropout)
    return encoder_model, decoder_model, ae_model, extended_ae_model


# Which fields of a checkpoint's config.json must agree with the running config.py before training may resume
# (behaviour of reference build.py:90-122: a mismatch in any of them raises ValueError).
_RESUME_LOCKED = (
    ('model', 'Model', ('name', 'run_name', 'encoder_architecture', 'dim_z', 'concat_midi_to_z', 'latent_flow_arch',
                        'logs_root_dir', 'note_duration', 'stack_spectrograms', 'increased_dataset_size', 'stft_args',
                        'spectrogram_size', 'mel_bins')),
    ('train', 'Train', ('minibatch_size', 'test_holdout_proportion', 'normalize_losses', 'optimizer',
                        'scheduler_name')),
)


def _canonical(value):
    """JSON has no tuples: a tuple saved to config.json comes back as a list, at any nesting depth."""
    if isinstance(value, (list, tuple)):
        return tuple(_canonical(v) for v in value)
    return value


def _config_attr(cfg, attr):
    """Instance attribute first (the reference's config namespaces are filled at import), class attribute otherwise."""
    return vars(cfg)[attr] if attr in vars(cfg) else getattr(cfg, attr)


def check_configs_on_resume_from_checkpoint(new_model_config, new_train_config, config_json_checkpoint):
    """Refuse to resume a run whose saved configuration disagrees with the current one on a field that shapes the
    model, the data or the optimizer (the fields of ``_RESUME_LOCKED``: the ones reference build.py:90-122 checks, same
    exception type).  Two deliberate differences from the reference, recorded in INTEGRATION.md section 3: lists and
    tuples compare equal at ANY nesting depth (the reference converts the top level only, so a nested tuple such as
    ``stft_args`` inside a list fails its check after a JSON round trip), and the message names both values.

    :raises: ValueError naming the first field that differs"""
    current = {'model': new_model_config, 'train': new_train_config}
    for section, title, fields in _RESUME_LOCKED:
        saved = config_json_checkpoint[section]
        for field in fields:
            now = _config_attr(current[section], field)
            if _canonical(saved[field]) != _canonical(now):
                raise ValueError(f"{title} attribute '{field}' changed since the checkpoint was written: "
                                 f"config.py has {now!r}, the checkpoint's config.json has {saved[field]!r}")
