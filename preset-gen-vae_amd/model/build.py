"""Model builders — the operator surface ``train.py`` / ``eval.py`` call (reference ``model/build.py``).

Same function names, arguments, return tuples and config fields as the reference
(``build_encoder_and_decoder_models`` build.py:11-31, ``build_ae_model`` :34-52, ``build_extended_ae_model`` :55-80);
the returned modules run on the HIP kernels of ``include/pgv_hip.h``.
"""
from . import VAE, decoder, encoder, extendedAE, regression


# Fields an older config.json / config.py may lack, with the value the reference assumes for them (build.py:13-14,61-62).
_LEGACY_MODEL_DEFAULTS = {'stack_specs_deepest_features_mix': True, 'params_reg_softmax': True}


def _with_legacy_defaults(model_config, *names):
    for name in names:
        if not hasattr(model_config, name):
            setattr(model_config, name, _LEGACY_MODEL_DEFAULTS[name])


def _stack_kwargs(model_config, train_config):
    """Constructor arguments shared by both conv stacks + the encoder's own, derived from the config bags: several MIDI
    notes fed one at a time (not stacked as channels) ask for the wider network; two latent slots are reserved for the
    MIDI note when it is concatenated to z."""
    several_notes = len(model_config.midi_notes) > 1
    common = dict(force_bigger_network=several_notes and not model_config.stack_spectrograms)
    enc_only = dict(output_bn=train_config.latent_flow_input_regularization.lower() == 'bn',
                    deepest_features_mix=model_config.stack_specs_deepest_features_mix)
    dim_z_enc = model_config.dim_z - (2 if model_config.concat_midi_to_z else 0)
    return dim_z_enc, common, enc_only


def build_encoder_and_decoder_models(model_config, train_config):
    """:return: Tuple: encoder, decoder (same architecture string for both; reference build.py:11-31)"""
    _with_legacy_defaults(model_config, 'stack_specs_deepest_features_mix')
    dim_z_enc, common, enc_only = _stack_kwargs(model_config, train_config)
    arch, shape, p_drop = model_config.encoder_architecture, model_config.input_tensor_size, train_config.fc_dropout
    return (encoder.SpectrogramEncoder(arch, dim_z_enc, shape, p_drop, **enc_only, **common),
            decoder.SpectrogramDecoder(arch, model_config.dim_z, shape, p_drop, **common))


def build_ae_model(model_config, train_config):
    """:return: Tuple: encoder, decoder, full AE model (reference build.py:34-52; only the flow-less VAE is in scope)"""
    products = getattr(train_config, 'fp32_products', None)
    if products is not None:   # (not a reference field; None leaves the process-wide setting - default 'bf16x6' - alone)
        from .. import ops
        ops.set_fp32_products(products)
    if model_config.latent_flow_arch is not None:
        raise NotImplementedError("FlowVAE (nflows latent flows) is out of scope of the MI355X hot path; "
                                  "set model.latent_flow_arch = None")
    stacks = build_encoder_and_decoder_models(model_config, train_config)
    vae = VAE.BasicVAE(stacks[0], model_config.dim_z, stacks[1], train_config.normalize_losses, train_config.latent_loss)
    return stacks + (vae,)


# params_regression_architecture = '<family>_<arch>': how each family is built (reference build.py:63-77)
def _mlp_regression(arch, model_config, train_config, idx_helper):
    assert model_config.forward_controls_loss is True   # an MLP cannot be run backwards on target values
    return regression.MLPRegression(arch, model_config.dim_z, idx_helper, train_config.reg_fc_dropout,
                                    cat_softmax_activation=model_config.params_reg_softmax)


def _flow_regression(arch, model_config, train_config, idx_helper):
    raise NotImplementedError("FlowRegression (nflows) is out of scope of the MI355X hot path")


_REGRESSION_FAMILIES = {'mlp': _mlp_regression, 'flow': _flow_regression}


def build_extended_ae_model(model_config, train_config, idx_helper):
    """:return: Tuple: encoder, decoder, AE model, ExtendedAE (the VAE + a preset regression on its latent vector;
    reference build.py:55-80)"""
    parts = build_ae_model(model_config, train_config)
    _with_legacy_defaults(model_config, 'params_reg_softmax')
    family, _, arch = model_config.params_regression_architecture.partition('_')
    if family not in _REGRESSION_FAMILIES or not arch:
        raise NotImplementedError("Synth param regression arch '{}' not implemented"
                                  .format(model_config.params_regression_architecture))
    reg_model = _REGRESSION_FAMILIES[family](arch, model_config, train_config, idx_helper)
    return parts + (extendedAE.ExtendedAE(parts[2], reg_model, idx_helper, train_config.fc_dropout),)


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
