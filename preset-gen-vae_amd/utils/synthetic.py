"""Synthetic 'Dexed-like' audio for benchmarks and smoke runs (no dataset ships with the reference: the preset DB is
a git-LFS pointer and the rendered wavs are git-ignored): an FM voice
sin(2 pi fc t + I sin(2 pi fm t)) * envelope, 88 576 samples @ 22 050 Hz (173 render buffers of 512, reference
synth/dexed.py:222-223), note-off at 3.0 s (config.py:29), linear fade-out over the last 2 205 samples
(synth/dexed.py:252-255)."""
import numpy as np


def fm_voice(n=88576, sr=22050, idx=0):
    t = np.arange(n, dtype=np.float64) / sr
    fc = 261.63 * [0.5, 1.0, 2.0, 3.0][idx % 4]
    fm = fc * [1.0, 2.0, 3.5, 0.5][(idx // 4) % 4]
    mod_index = 1.0 + 7.0 * ((idx * 37) % 11) / 10.0
    env = np.minimum(1.0, t / 0.01) * np.exp(-t * (0.3 + 0.2 * (idx % 3)))
    rel = np.where(t > 3.0, np.exp(-(t - 3.0) * 6.0), 1.0)
    fade = np.ones(n)
    nf = min(n, 2205)
    fade[-nf:] = np.linspace(1.0, 0.0, nf)
    return (0.9 * np.sin(2 * np.pi * fc * t + mod_index * np.sin(2 * np.pi * fm * t)) * env * rel * fade).astype(
        np.float32)
