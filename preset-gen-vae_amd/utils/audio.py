"""STFT -> mel -> dB front-end on the HIP kernel ``pgv_stft_mel`` (surface of the reference's ``utils/audio.py``:
``Spectrogram`` :20-69, ``MelSpectrogram`` :73-92; dataset min-max normalisation ``data/abstractbasedataset.py:129-131``).

The reference computes one spectrogram per item on the CPU inside DataLoader workers (torch.stft + dense librosa mel
matmul).  Here a whole minibatch of waveforms ``[B, n_samples]`` already in HBM becomes ``[B, 1, n_mels, n_frames]`` in
one launch: frames are cut from an LDS-resident copy of the samples, transformed by a radix-4 LDS FFT, projected on
the mel filterbank in CSR form (1016 non-zeros of 257x513), clamped, converted to dB and (optionally) min-max
normalised in the same kernel.

The mel filterbank is ``librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm=None)`` (librosa
~=0.8.0, reference requirements.txt:5 — NOT vendored in the reference and not installed here: **parity unpinned** for
this third-party piece; restated from the published Slaney construction, see ``slaney_mel_basis``)."""
import numpy as np
import torch

from .. import ops


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide='ignore', invalid='ignore'):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """Slaney-style triangular mel filterbank without area normalisation, float32 [n_mels, n_fft//2+1].

    Published construction of ``librosa.filters.mel(..., htk=False, norm=None)``: mel scale linear below 1 kHz
    (200/3 Hz per mel) and logarithmic above (step ln(6.4)/27); n_mels+2 equally spaced mel points; weights
    ``max(0, min(lower_slope, upper_slope))`` from ``np.subtract.outer(mel_f, fftfreqs)``."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = n_fft // 2 + 1
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins, endpoint=True)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    return weights.astype(np.float32)


def dense_to_csr(w):
    rows, cols = np.nonzero(w)
    row_ptr = np.zeros(w.shape[0] + 1, dtype=np.int32)
    np.add.at(row_ptr, rows + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.int32)
    return row_ptr, cols.astype(np.int32), w[rows, cols].astype(np.float32)


class Spectrogram:
    """dB spectrogram of raw audio (reference utils/audio.py:20-69).  ``__call__`` accepts what the reference accepts
    (a 1-D float array -> Tensor[n_fft/2+1, T]) and additionally a batch ``[B, n_samples]`` -> ``[B, rows, T]``.
    Computation happens on ``device`` (a ROCm device is required)."""

    def __init__(self, n_fft, fft_hop, min_dB, dynamic_range_dB=None, log_scale=True, device='cuda'):
        if n_fft != 1024:
            raise NotImplementedError("the HIP front-end implements n_fft=1024 (reference config.py:31)")
        self.n_fft = n_fft
        self.fft_hop = fft_hop
        self.log_scale = log_scale
        self.min_dB = min_dB
        self.dynamic_range_dB = dynamic_range_dB
        self.device = torch.device(device)
        # symmetric Hann (torch.hann_window(periodic=False), audio.py:30) and its DC gain (= max|rfft(w)|, :31)
        n = np.arange(n_fft, dtype=np.float64)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / (n_fft - 1))
        self.window = torch.tensor(win, dtype=torch.float32)
        self.spectrogram_norm_factor = float(self.window.double().sum().item())
        self._window_dev = None
        self._mel_csr = None
        self.n_mel_bins = 0
        # optional fused min-max normalisation: out = -1 + (s - min)/((max-min)/2)  (abstractbasedataset.py:129-131)
        self.minmax = None

    # -- helpers -----------------------------------------------------------------------------------------------
    def set_minmax_normalization(self, spec_min, spec_max):
        self.minmax = (float(spec_min), float(spec_max))

    def n_frames(self, n_samples):
        return 1 + n_samples // self.fft_hop  # torch.stft(center=True)

    def _prep(self, x_wav):
        single = False
        if not torch.is_tensor(x_wav):
            x_wav = torch.tensor(np.asarray(x_wav), dtype=torch.float32)
        x_wav = x_wav.to(device=self.device, dtype=torch.float32)
        if x_wav.dim() == 1:
            x_wav, single = x_wav.unsqueeze(0), True
        if self._window_dev is None or self._window_dev.device != x_wav.device:
            self._window_dev = self.window.to(x_wav.device)
        return x_wav.contiguous(), single

    def _run(self, x_wav, log_scale=True, out=None, mode=None):
        x, single = self._prep(x_wav)
        floor = 10 ** (self.min_dB / 20.0)
        a, b = 1.0, 0.0
        if mode is None:
            mode = ops.STFT_DB if log_scale else ops.STFT_LINEAR
        if mode == ops.STFT_COMPLEX:
            res = ops.stft_mel(x, self.fft_hop, self.n_frames(x.shape[1]), self._window_dev,
                               self.spectrogram_norm_factor, None, 0, floor, 1.0, 0.0, mode=mode)
            return res[0] if single else res
        if self.minmax is not None and mode == ops.STFT_DB:
            mn, mx = self.minmax
            a = 2.0 / (mx - mn)
            b = -1.0 - mn * a
        csr = None
        if self.n_mel_bins > 0:
            if self._mel_csr is None or self._mel_csr[2].device != x.device:
                rp, col, val = dense_to_csr(self.mel_basis)
                self._mel_csr = (torch.tensor(rp, device=x.device), torch.tensor(col, device=x.device),
                                 torch.tensor(val, device=x.device))
            csr = self._mel_csr
        out = ops.stft_mel(x, self.fft_hop, self.n_frames(x.shape[1]), self._window_dev,
                           self.spectrogram_norm_factor, csr, self.n_mel_bins, floor, a, b, out=out, mode=mode)
        return out[0] if single else out

    def get_stft(self, x_wav):
        """The complex, non-normalised STFT (reference audio.py:33-40: torch.stft(n_fft, hop, window, center=True,
        pad_mode='constant', onesided=True)): complex64 ``[n_fft/2+1, T]`` (``[B, n_fft/2+1, T]`` for a batch)."""
        return self._run(x_wav, mode=ops.STFT_COMPLEX)

    def __call__(self, x_wav):
        """Log-scale spectrogram with the 'floor' dB value, or (``log_scale=False``) the normalised amplitudes
        (audio.py:42-50)."""
        return self._run(x_wav, self.log_scale)

    def linear_to_log_scale(self, spectrogram):
        spectrogram = torch.clamp(spectrogram, min=10 ** (self.min_dB / 20.0))
        return 20.0 * torch.log10(spectrogram)

    def linear_to_log_scale_with_dynamic_range(self, spectrogram):
        """audio.py:63-69: log scale, then everything below (max - dynamic_range_dB) is raised to that level."""
        assert self.dynamic_range_dB is not None  # (the reference asserts the same)
        spectrogram = self.linear_to_log_scale(spectrogram)
        return torch.maximum(spectrogram, torch.max(spectrogram) - self.dynamic_range_dB)

    def log_to_linear_scale(self, spectrogram):
        return torch.pow(10.0, spectrogram / 20.0) * self.spectrogram_norm_factor


class MelSpectrogram(Spectrogram):
    """Log-scale mel spectrogram (reference utils/audio.py:73-87)."""

    def __init__(self, n_fft, fft_hop, min_dB, n_mel_bins, Fs, device='cuda'):
        super().__init__(n_fft, fft_hop, min_dB, log_scale=True, device=device)
        self.Fs = Fs
        self.n_mel_bins = n_mel_bins
        # librosa.feature.melspectrogram(S=..., n_mels=..., norm=None) ignores self.Fs and uses its default sr=22050
        # (audio.py:85-86); the reference's own Fs is 22050 (config.py:30), so both coincide.
        self.mel_basis = slaney_mel_basis(22050, n_fft, n_mel_bins)

    def mel_dB_to_STFT(self, mel_spectrogram):
        """Reference audio.py:89-92 inverts the filterbank with ``librosa.feature.inverse.mel_to_stft`` - an iterative
        non-negative least-squares solve of an under-determined system (257 mel rows, 513 bins) inside librosa ~=0.8
        (not vendored, not installed): its result depends on that solver's iteration, not on a closed form, so it has no
        restatement here.  Off the train-step path (evaluation / audio export only)."""
        raise NotImplementedError("mel_dB_to_STFT needs librosa's NNLS mel inversion (third-party, out of scope); use "
                                  "log_to_linear_scale() for the mel-amplitude spectrogram")

    def batch(self, wav, out=None):
        """[B, n_samples] -> [B, 1, n_mels, T]: the tensor layout the encoder consumes.  ``out`` (contiguous
        [B, 1, n_mels, T]): write there - e.g. ``VAETrainStep.static_input``, so that the captured step consumes the
        spectrograms where the front-end leaves them."""
        res = self._run(wav, out=out)
        return out if out is not None else res.unsqueeze(1)
