"""'Extended auto-encoder': spectrogram VAE + a network inferring synth parameters from the latent vector
(surface of the reference's ``model/extendedAE.py:13-51``).  Pure delegation; it only fixes the call signatures
``forward(x, sample_info=None)`` and ``latent_loss(4 args)`` that ``train.py:209,225`` use.  Must survive
``nn.DataParallel`` wrapping: no per-call tensors are stored on ``self``."""
import torch.nn as nn

from . import VAE, regression

# accepted sub-model classes -> is it flow-based (only the flow-less ones exist on this path)
_AE_KINDS = ((VAE.BasicVAE, False),)
_REG_KINDS = ((regression.MLPRegression, False),)


def _kind_of(module, kinds, what):
    for cls, flow_based in kinds:
        if isinstance(module, cls):
            return flow_based
    raise TypeError(f"Unrecognized {what} model")


class ExtendedAE(nn.Module):
    def __init__(self, ae_model, reg_model, idx_helper, dropout_p=0.0):
        super().__init__()
        flow_latent = _kind_of(ae_model, _AE_KINDS, "auto-encoder")
        flow_reg = _kind_of(reg_model, _REG_KINDS, "synth params regression")
        self.idx_helper = idx_helper
        self.ae_model, self.reg_model = ae_model, reg_model    # state-dict prefixes 'ae_model.' / 'reg_model.'
        self._flow_based = (flow_latent, flow_reg)

    is_flow_based_latent_space = property(lambda self: self._flow_based[0])
    is_flow_based_regression = property(lambda self: self._flow_based[1])

    def forward(self, x, sample_info=None, **inject):
        """Auto-encodes the input (does NOT perform synth parameters regression)."""
        return self.ae_model(x, sample_info, **inject)

    def latent_loss(self, z_0_mu_logvar, z_0_sampled, z_K_sampled, log_abs_det_jac):
        return self.ae_model.latent_loss(z_0_mu_logvar, z_0_sampled, z_K_sampled, log_abs_det_jac)
