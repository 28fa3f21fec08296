"""Gaussian log-probabilities of latent samples (reference ``utils/probability.py:13-29``) and the ELBO terms the
flow-based VAE builds from them (``model/VAE.py:183-193``).  The flow transform itself (nflows) is out of scope; these
are the row reductions around it, evaluated on the device the samples live on (``[B, D]`` with B = 256, D = 64..512:
a few KiB - stock torch reductions, nothing here is bandwidth- or compute-relevant)."""
import math

import torch

_LOG_2_PI = math.log(2.0 * math.pi)


def standard_gaussian_log_probability(samples):
    """log N(samples; 0, I) per row (probability.py:13-18)."""
    return -0.5 * (samples.shape[1] * _LOG_2_PI + torch.sum(samples ** 2, dim=1))


def gaussian_log_probability(samples, mu, log_var):
    """log N(samples; mu, diag(exp(log_var))) per row (probability.py:21-29)."""
    return -0.5 * (samples.shape[1] * _LOG_2_PI
                   + torch.sum(log_var + ((samples - mu) ** 2 / torch.exp(log_var)), dim=1))


def flow_latent_loss(z_0_mu_logvar, z_0_sampled, z_K_sampled, log_abs_det_jac, normalize=False):
    """Negative ELBO latent terms of ``FlowVAE.latent_loss`` (VAE.py:183-193): ``-(log p(z_K) - log q(z_0) +
    log|det J|)`` averaged over the batch (and divided by D when ``normalize``)."""
    log_q = gaussian_log_probability(z_0_sampled, z_0_mu_logvar[:, 0, :], z_0_mu_logvar[:, 1, :])
    log_p = standard_gaussian_log_probability(z_K_sampled)
    loss = -(log_p - log_q + log_abs_det_jac).mean()
    return loss / z_0_sampled.shape[1] if normalize else loss
