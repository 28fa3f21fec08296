"""'Extended auto-encoder': spectrogram VAE + a network inferring synth parameters from the latent vector
(surface of the reference's ``model/extendedAE.py:13-51``).  Pure delegation; it only fixes the call signatures
``forward(x, sample_info=None)`` and ``latent_loss(4 args)`` that ``train.py:209,225`` use.  Must survive
``nn.DataParallel`` wrapping: no per-call tensors are stored on ``self``."""
import torch.nn as nn

from . import VAE, regression


class ExtendedAE(nn.Module):
    def __init__(self, ae_model, reg_model, idx_helper, dropout_p=0.0):
        super().__init__()
        self.idx_helper = idx_helper
        self.ae_model = ae_model
        if isinstance(self.ae_model, VAE.BasicVAE):
            self._is_flow_based_latent_space = False
        else:
            raise TypeError("Unrecognized auto-encoder model")
        self.reg_model = reg_model
        if isinstance(self.reg_model, regression.MLPRegression):
            self._is_flow_based_regression = False
        else:
            raise TypeError("Unrecognized synth params regression model")

    @property
    def is_flow_based_latent_space(self):
        return self._is_flow_based_latent_space

    @property
    def is_flow_based_regression(self):
        return self._is_flow_based_regression

    def forward(self, x, sample_info=None, **inject):
        """Auto-encodes the input (does NOT perform synth parameters regression)."""
        return self.ae_model(x, sample_info, **inject)

    def latent_loss(self, z_0_mu_logvar, z_0_sampled, z_K_sampled, log_abs_det_jac):
        return self.ae_model.latent_loss(z_0_mu_logvar, z_0_sampled, z_K_sampled, log_abs_det_jac)
