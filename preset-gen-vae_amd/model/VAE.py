"""Basic (Gaussian, diagonal) VAE on HIP kernels (surface of the reference's ``model/VAE.py``).

:class:`BasicVAE` keeps the reference's arithmetic (VAE.py:37-66: encode, sigma = exp(logvar/2), eps ~ N(0,1),
z = mu + sigma*eps in train mode / z = mu in eval mode, decode, 5-tuple output) but accepts the **FlowVAE-shaped call
signatures** that ``ExtendedAE`` / ``train.py`` actually use (``forward(x, sample_info=None)``,
``latent_loss(z_0_mu_logvar, z_0_sampled, z_K_sampled, log_abs_det_jac)``; reference VAE.py:137,183,
extendedAE.py:42-51) — the reference's own ``BasicVAE`` cannot be driven through ``ExtendedAE``.
``FlowVAE`` (nflows RealNVP/MAF) is out of scope.
"""
import torch
import torch.nn as nn

from .. import ops
from . import loss as loss_mod


class _ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ml, eps):
        ml = ml.contiguous()
        ctx.save_for_backward(ml, eps)
        z, _ = ops.reparam_kl_fwd(ml, eps, 0.0, want_z=True)
        return z

    @staticmethod
    def backward(ctx, g_z):
        ml, eps = ctx.saved_tensors
        return ops.reparam_kl_bwd(ml, eps, g_z.contiguous(), None, 0.0), None


class BasicVAE(nn.Module):
    """A standard VAE over given encoder/decoder networks; dim_z independent Gaussians (reference VAE.py:19-66)."""

    def __init__(self, encoder, dim_z, decoder, normalize_latent_loss, latent_loss_type):
        super().__init__()
        self.encoder = encoder
        self.dim_z = dim_z
        self.decoder = decoder
        self.is_profiled = False
        if latent_loss_type.lower() == 'dkl':
            self.latent_criterion = loss_mod.GaussianDkl(normalize=normalize_latent_loss)
        else:
            raise NotImplementedError("Latent loss '{}' unavailable".format(latent_loss_type))

    def forward(self, x, sample_info=None, eps=None, enc_dropout_mask=None, dec_dropout_mask=None):
        """:returns: z_mu_logvar, z_sampled, zK_sampled=z_sampled, logabsdetjacT=0.0, x_out.

        ``eps`` / ``*_dropout_mask`` inject the random draws (parity harness); by default they come from the on-device
        Philox stream (``rng.py``)."""
        z_mu_logvar = self.encoder(x, dropout_mask=enc_dropout_mask)
        n_minibatch = z_mu_logvar.size()[0]
        if self.training:
            if eps is None:
                from ..rng import device_rng
                eps = device_rng(self, z_mu_logvar.device).normal((n_minibatch, self.dim_z))
            z_sampled = _ReparamFn.apply(z_mu_logvar, eps.contiguous())
        else:  # eval mode: no random sampling (VAE.py:57-58)
            z_sampled = _ReparamFn.apply(z_mu_logvar, None)
        x_out = self.decoder(z_sampled, dropout_mask=dec_dropout_mask)
        return z_mu_logvar, z_sampled, z_sampled, torch.zeros((n_minibatch, 1), device=x.device), x_out

    def latent_loss(self, z_0_mu_logvar, z_0_sampled=None, z_K_sampled=None, log_abs_det_jac=None, **kwargs):
        """Dkl vs. zero-mean unit-variance Gaussian (reference VAE.py:63-66); extra args exist for flow compatibility."""
        return self.latent_criterion.from_mu_logvar(z_0_mu_logvar)
