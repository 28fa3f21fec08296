"""Model surface of the reference's ``model/`` package for the conv-VAE hot path."""
