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


class _ReparamKlFn(torch.autograd.Function):
    """z = mu + sigma * eps AND the Dkl term from one launch (the kernel reads mu / logvar once for both); backward takes
    whichever of the two output gradients exist, again in one launch."""

    @staticmethod
    def forward(ctx, ml, eps, kl_scale, rng=None, kl_buf=None):
        """``eps`` None: drawn from ``rng`` inside the kernel (one launch for draw, z and Dkl); ``kl_buf``: a zeroed
        element of the step scratch for the Dkl term (no clearing launch)."""
        ml = ml.contiguous()
        ctx.kl_scale = kl_scale
        ctx.set_materialize_grads(False)
        if eps is None:
            z, kl, eps = rng.reparam_kl(ml, kl_scale, kl_buf)
        else:
            z, kl = ops.reparam_kl_fwd(ml, eps, kl_scale, want_z=True)
        ctx.save_for_backward(ml, eps)
        return z, kl

    @staticmethod
    def backward(ctx, g_z, g_kl):
        ml, eps = ctx.saved_tensors
        if g_z is None and g_kl is None:
            return None, None, None, None, None
        return ops.reparam_kl_bwd(ml, eps, None if g_z is None else g_z.contiguous(),
                                  None if g_kl is None else g_kl.contiguous(), ctx.kl_scale), None, None, None, None


_NO_FLOW_LADJ = {}


def _zero_log_abs_det_jac(n, device):
    """The flow placeholder of the 5-tuple (VAE.py:60: no flow, log|det J| = 0): one constant tensor per (device, n)
    instead of a fill launch per forward.  Read-only by contract (the reference only feeds it to latent_loss)."""
    key = (str(device), n)
    t = _NO_FLOW_LADJ.get(key)
    if t is None:
        t = _NO_FLOW_LADJ[key] = torch.zeros((n, 1), device=device)
    return t


class BasicVAE(nn.Module):
    """A standard VAE over given encoder/decoder networks; dim_z independent Gaussians (reference VAE.py:19-66)."""

    def __init__(self, encoder, dim_z, decoder, normalize_latent_loss, latent_loss_type):
        super().__init__()
        self.encoder = encoder
        self.dim_z = dim_z
        self.decoder = decoder
        self.is_profiled = False
        # 'mse_mean' / 'l2_batch' / 'l2_batch_contents' / None: evaluate that reconstruction criterion against the
        # input inside the decoder's output stack (train mode) and hand it to loss.MSELoss / loss.L2Loss through an
        # attribute of x_out, so that the criterion's backward fuses with the output block's (set by VAETrainStep).
        # '+deferred' appended: the value is produced BY the backward kernel and reads zero before backward has run
        self.fuse_recons_criterion = None
        if latent_loss_type.lower() == 'dkl':
            self.latent_criterion = loss_mod.GaussianDkl(normalize=normalize_latent_loss)
        else:
            raise NotImplementedError("Latent loss '{}' unavailable".format(latent_loss_type))

    def forward(self, x, sample_info=None, eps=None, enc_dropout_mask=None, dec_dropout_mask=None):
        """:returns: z_mu_logvar, z_sampled, zK_sampled=z_sampled, logabsdetjacT=0.0, x_out.

        ``eps`` / ``*_dropout_mask`` inject the random draws (parity harness); by default they come from the on-device
        Philox stream (``rng.py``)."""
        rng = None
        if self.training:
            # ONE generator for the three random draws of a step (two fc Dropout masks, eps), each on its own Philox
            # stream id (independent masks, as the reference's torch generator gives), advanced once per forward
            from ..rng import device_rng
            rng = device_rng(self, x.device)
            object.__setattr__(self.encoder, '_rng', rng)
            object.__setattr__(self.decoder, '_rng', rng)
            rng.begin()
        try:
            fused_head = None
            if self.training and eps is None:
                # eps is drawn inside the launch that uses it, Dkl accumulated into a zeroed word of the step scratch; an
                # encoder that ends in BatchNorm1d evaluates all of it in that layer's launch (layer.EncoderHeadFn)
                from . import layer
                kl_buf = layer._small_zeros(next(self.parameters()), (1,), 'kl')   # (None: cleared by the call)
                kl_scale = self.latent_criterion.kl_scale(torch.empty((x.shape[0], 2, self.dim_z), device='meta'))
                enc_out = self.encoder(x, dropout_mask=enc_dropout_mask, reparam=(rng, kl_scale, kl_buf))
                if isinstance(enc_out, tuple):
                    z_mu_logvar, fused_head = enc_out[0], enc_out[1:]
                else:
                    z_mu_logvar = enc_out
            else:
                z_mu_logvar = self.encoder(x, dropout_mask=enc_dropout_mask)
            n_minibatch = z_mu_logvar.size()[0]
            if self.training:
                # the Dkl term rides along (same kernel, same read of mu / logvar) and is handed to latent_loss() through an
                # attribute of the returned tensor OBJECT; a caller that passes another tensor simply recomputes it
                kl_scale = self.latent_criterion.kl_scale(z_mu_logvar)
                if fused_head is not None:
                    z_sampled, kl = fused_head
                elif eps is None:
                    z_sampled, kl = _ReparamKlFn.apply(z_mu_logvar, None, kl_scale, rng, kl_buf)
                else:
                    z_sampled, kl = _ReparamKlFn.apply(z_mu_logvar, eps.contiguous(), kl_scale)
                z_mu_logvar._pgv_kl = (kl_scale, kl)
            else:  # eval mode: no random sampling (VAE.py:57-58)
                z_sampled = _ReparamFn.apply(z_mu_logvar, None)
            kind = self.fuse_recons_criterion
            if self.training and kind is not None and x.shape[1] == 1:
                opts = kind.split('+')[1:]
                deferred = 'deferred' in opts            # value delivered by the backward kernel (layer.ConvStackFn)
                # 'unit' (with 'deferred'): the caller's promise that the value enters its total with gradient 1 and that
                # x_out gets no other gradient - the criterion then rides in the output layer's forward kernel
                from . import layer as _layer
                unit = deferred and 'unit' in opts and _layer.UNIT_RECONS_GRADIENT
                scale = {'mse_mean': 1.0 / x.numel(), 'l2_batch': 1.0 / x.shape[0],
                         'l2_batch_contents': 1.0 / x.numel()}[kind.split('+')[0]]
                x_out, recons = self.decoder(z_sampled, dropout_mask=dec_dropout_mask, sq_target=x,
                                             sq_scale=((-scale, 'unit') if unit else (-scale if deferred else scale)))
                x_out._pgv_recons = (x.data_ptr(), scale, recons)
            else:
                x_out = self.decoder(z_sampled, dropout_mask=dec_dropout_mask)
        finally:
            # also when the encoder / decoder raises: a generator left in deferred mode would hand every later
            # stand-alone training-mode call the same Dropout masks (its offset would never advance)
            if rng is not None:
                rng.flush()
        return z_mu_logvar, z_sampled, z_sampled, _zero_log_abs_det_jac(n_minibatch, x.device), x_out

    def latent_loss(self, z_0_mu_logvar, z_0_sampled=None, z_K_sampled=None, log_abs_det_jac=None, **kwargs):
        """Dkl vs. zero-mean unit-variance Gaussian (reference VAE.py:63-66); extra args exist for flow compatibility."""
        cached = getattr(z_0_mu_logvar, '_pgv_kl', None)
        if cached is not None and cached[0] == self.latent_criterion.kl_scale(z_0_mu_logvar):
            return cached[1]
        return self.latent_criterion.from_mu_logvar(z_0_mu_logvar)
