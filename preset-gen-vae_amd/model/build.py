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


def _is_attr_equal(attr1, attr2):
    """Compares two config attributes - lists auto converted to tuples (reference build.py:83-87)."""
    _attr1 = tuple(attr1) if isinstance(attr1, list) else attr1
    _attr2 = tuple(attr2) if isinstance(attr2, list) else attr2
    return _attr1 == _attr2


def check_configs_on_resume_from_checkpoint(new_model_config, new_train_config, config_json_checkpoint):
    """Consistency check between the config saved with the last checkpoint (config.json) and the new config.py
    (reference build.py:90-122, same attribute lists, same messages).

    :raises: ValueError if any incompatibility is found"""
    prev_config = config_json_checkpoint['model']
    attributes_to_check = ['name', 'run_name', 'encoder_architecture', 'dim_z', 'concat_midi_to_z', 'latent_flow_arch',
                           'logs_root_dir', 'note_duration', 'stack_spectrograms', 'increased_dataset_size',
                           'stft_args', 'spectrogram_size', 'mel_bins']
    for attr in attributes_to_check:
        if not _is_attr_equal(prev_config[attr], _config_attr(new_model_config, attr)):
            raise ValueError("Model attribute '{}' is different in the new config.py ({}) and the old config.json ({})"
                             .format(attr, _config_attr(new_model_config, attr), prev_config[attr]))
    prev_config = config_json_checkpoint['train']
    attributes_to_check = ['minibatch_size', 'test_holdout_proportion', 'normalize_losses', 'optimizer',
                           'scheduler_name']
    for attr in attributes_to_check:
        if not _is_attr_equal(prev_config[attr], _config_attr(new_train_config, attr)):
            raise ValueError("Train attribute '{}' is different in the new config.py ({}) and the old config.json ({})"
                             .format(attr, _config_attr(new_train_config, attr), prev_config[attr]))


def _config_attr(cfg, attr):
    """The reference reads ``cfg.__dict__[attr]`` (its config classes are plain namespaces filled at import); class
    attributes of a config class that was never instantiated live on the class, so look there as well."""
    d = cfg.__dict__
    if attr in d:
        return d[attr]
    return getattr(cfg, attr)
