"""preset-gen-vae hot path, MI355X-native: the conv-VAE train step and the STFT->mel front-end behind the
reference's ``config`` / ``model.build`` operator surface, on hand-written gfx950 HIP kernels."""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "config", "model", "utils", "optim", "rng", "train_step", "parallel"]
