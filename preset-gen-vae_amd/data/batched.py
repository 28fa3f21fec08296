"""The step immediately before the encoder: raw audio -> scaled dB (mel-)spectrogram minibatches with
the reference's item contract, computed on the GPU by the HIP front-end instead of per item in DataLoader workers.

Mirrors ``data/abstractbasedataset.py``:

* ``__getitem__`` (:101-145): ``(spectrograms[C,F,T], params[L], int32[UID, pitch, vel], labels)`` per item — here
  ``get_batch(indexes)`` returns the same four entries with a leading batch dimension, spectrograms already on the
  device in the ``[B, C, 257, 347]`` layout the encoder consumes;
* normalisation (:129-133) ``'min_max'`` -> ``-1 + (s - min) / ((max - min) / 2)`` (fused into the front-end kernel's
  epilogue), ``'mean_std'`` -> ``(s - mean) / std``, ``None``; ``denormalize_spectrogram`` (:340-346);
* the statistics pass (:348-391): per-item min / max / mean / var of the un-normalised spectrogram, dataset-wide
  ``min = min(mins)``, ``max = max(maxs)``, ``mean = mean(means)``, ``std = sqrt(mean(vars))`` (torch.var: unbiased).

The preset database, the synth renderer and the wav cache are out of scope: waveforms, parameter vectors and UIDs are
given as in-memory arrays (one row per preset and MIDI note)."""
import json

import numpy as np
import torch

from ..utils import audio


class BatchedPresetSpectrograms:
    def __init__(self, waves, params, uids, midi_notes=((60, 85),), labels=None, n_fft=1024, fft_hop=256,
                 n_mel_bins=257, spectrogram_min_dB=-120.0, spectrogram_normalization='min_max', device='cuda',
                 multichannel_stacked_spectrograms=False):
        """``waves``: float array ``[n_presets, n_notes, n_samples]`` (or ``[n_presets, n_samples]`` for one note);
        ``params``: ``[n_presets, L]`` learnable parameter values in [0, 1]; ``uids``: ``[n_presets]`` ints;
        ``midi_notes``: the (pitch, velocity) pairs of the note axis (reference config.py:35)."""
        waves = np.asarray(waves, dtype=np.float32)
        if waves.ndim == 2:
            waves = waves[:, None, :]
        if waves.shape[1] != len(midi_notes):
            raise ValueError("waves must hold one row per MIDI note")
        if len(midi_notes) == 1 and multichannel_stacked_spectrograms:
            raise AssertionError("a 1-note dataset cannot stack spectrograms")      # abstractbasedataset.py:58-59
        self.waves = torch.from_numpy(waves)
        self.params = torch.as_tensor(np.asarray(params), dtype=torch.float32)
        self.valid_preset_UIDs = np.asarray(uids)
        self.labels = labels
        self.midi_notes = tuple(tuple(n) for n in midi_notes)
        self._multichannel_stacked_spectrograms = multichannel_stacked_spectrograms
        self.n_fft, self.fft_hop, self.n_mel_bins = n_fft, fft_hop, n_mel_bins
        self.device = torch.device(device)
        if n_mel_bins <= 0:                                                       # abstractbasedataset.py:70-74
            self.spectrogram = audio.Spectrogram(n_fft, fft_hop, spectrogram_min_dB, device=device)
        else:
            self.spectrogram = audio.MelSpectrogram(n_fft, fft_hop, spectrogram_min_dB, n_mel_bins, 22050,
                                                    device=device)
        if spectrogram_normalization not in (None, 'min_max', 'mean_std'):
            raise ValueError(f"unknown spectrogram_normalization {spectrogram_normalization!r}")
        self.spectrogram_normalization = spectrogram_normalization
        self.spec_stats = None

    # ---- sizes (abstractbasedataset.py:95-99, 147-170) ----------------------------------------------------------
    @property
    def valid_presets_count(self):
        return len(self.valid_preset_UIDs)

    @property
    def midi_notes_per_preset(self):
        return len(self.midi_notes)

    def __len__(self):
        if self._multichannel_stacked_spectrograms:
            return self.valid_presets_count
        return self.valid_presets_count * self.midi_notes_per_preset

    # ---- normalisation -----------------------------------------------------------------------------------------
    def _raw_spectrograms(self, wav_rows):
        """[n, n_samples] host rows -> un-normalised dB spectrograms [n, F, T] on the device."""
        spec = self.spectrogram
        saved = getattr(spec, 'minmax', None)
        spec.minmax = None
        try:
            return spec(wav_rows.to(self.device, non_blocking=True))
        finally:
            spec.minmax = saved

    def normalize_spectrogram(self, spectrogram):
        if self.spectrogram_normalization == 'min_max':
            return -1.0 + (spectrogram - self.spec_stats['min']) / ((self.spec_stats['max'] - self.spec_stats['min']) / 2.0)
        if self.spectrogram_normalization == 'mean_std':
            return (spectrogram - self.spec_stats['mean']) / self.spec_stats['std']
        return spectrogram

    def denormalize_spectrogram(self, spectrogram):
        if self.spectrogram_normalization == 'min_max':
            return (spectrogram + 1.0) * ((self.spec_stats['max'] - self.spec_stats['min']) / 2.0) + self.spec_stats['min']
        if self.spectrogram_normalization == 'mean_std':
            return spectrogram * self.spec_stats['std'] + self.spec_stats['mean']
        return spectrogram

    # ---- statistics pass ---------------------------------------------------------------------------------------
    def compute_and_store_spectrograms_stats(self, json_path=None, batch_size=64):
        """Per-item min / max / mean / var on the device, aggregated as the reference does; optionally written as the
        reference's ``.json`` (same keys).  Returns ``(dataset_stats, full_stats)``."""
        rows = self.waves.reshape(-1, self.waves.shape[-1])
        mins, maxs, means, vrs = [], [], [], []
        for i in range(0, rows.shape[0], batch_size):
            s = self._raw_spectrograms(rows[i:i + batch_size]).double()
            flat = s.reshape(s.shape[0], -1)
            mins.append(flat.amin(dim=1))
            maxs.append(flat.amax(dim=1))
            means.append(flat.mean(dim=1))
            vrs.append(flat.var(dim=1))              # torch.var default: unbiased, as abstractbasedataset.py:388
        full = {'UID': np.repeat(self.valid_preset_UIDs, self.midi_notes_per_preset),
                'min': torch.cat(mins).cpu().numpy(), 'max': torch.cat(maxs).cpu().numpy(),
                'mean': torch.cat(means).cpu().numpy(), 'var': torch.cat(vrs).cpu().numpy()}
        stats = {'min': float(full['min'].min()), 'max': float(full['max'].max()),
                 'mean': float(full['mean'].mean()), 'std': float(np.sqrt(full['var'].mean()))}
        full['std'] = np.sqrt(full['var'])
        del full['var']
        if json_path is not None:
            with open(json_path, 'w') as f:
                json.dump(stats, f)
        self.set_spec_stats(stats)
        return stats, full

    def set_spec_stats(self, stats):
        self.spec_stats = dict(stats)
        if self.spectrogram_normalization == 'min_max':     # fused into the front-end kernel's epilogue
            self.spectrogram.set_minmax_normalization(stats['min'], stats['max'])

    # ---- items -------------------------------------------------------------------------------------------------
    def get_batch(self, indexes):
        """The reference's item tuple for a minibatch of dataset indexes (abstractbasedataset.py:101-145):
        ``(spectrograms [B, C, F, T] on the device, params [B, L], int32 [B, 3] = (UID, pitch, velocity), labels)``."""
        if self.spectrogram_normalization is not None and self.spec_stats is None:
            raise RuntimeError("spectrogram statistics are not set: run compute_and_store_spectrograms_stats() "
                               "or set_spec_stats()")
        indexes = np.asarray(indexes, dtype=np.int64)
        n_notes = self.midi_notes_per_preset
        if n_notes > 1 and not self._multichannel_stacked_spectrograms:
            preset_idx, note_idx = indexes // n_notes, indexes % n_notes
            wav = self.waves[torch.from_numpy(preset_idx), torch.from_numpy(note_idx)]            # [B, n_samples]
            C = 1
            ref_notes = [self.midi_notes[int(j)] for j in note_idx]
        else:
            preset_idx = indexes
            wav = self.waves[torch.from_numpy(preset_idx)].reshape(-1, self.waves.shape[-1])      # [B*C, n_samples]
            C = n_notes
            ref_notes = [self.midi_notes[0]] * len(indexes)
        if self.spectrogram_normalization == 'min_max':
            spec = self.spectrogram(wav.to(self.device, non_blocking=True))       # normalisation fused in the kernel
        else:
            spec = self.normalize_spectrogram(self._raw_spectrograms(wav))
        spec = spec.reshape(len(indexes), C, spec.shape[-2], spec.shape[-1])
        info = torch.tensor([[int(self.valid_preset_UIDs[p]), int(n[0]), int(n[1])] for p, n in zip(preset_idx, ref_notes)],
                            dtype=torch.int32)
        if self.labels is None:   # 'NoLabel' is the only default label (abstractbasedataset.py:270-273)
            labels = torch.ones((len(indexes), 1), dtype=torch.int8)
        else:
            labels = torch.stack([torch.as_tensor(self.labels[int(p)], dtype=torch.int8) for p in preset_idx])
        return spec, self.params[torch.from_numpy(preset_idx)], info, labels
