"""GPU: the fused STFT -> mel -> dB kernel against the numpy oracle and the reference's own Spectrogram outputs.

fp32 STFT noise: the reference's float32 torch.stft differs from float64 by ~1e-7 of the frame's largest bin, which
is a large *relative* error for bins near the -120 dB floor, so the comparison is absolute in the linear domain
(<= 3e-6 of the frame maximum) and 0.02 dB wherever the magnitude is above -80 dB."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import audio_oracle as ao

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")


def _compare_db(got_db, ref_db, strong_db=-80.0):
    got_db, ref_db = np.asarray(got_db, dtype=np.float64), np.asarray(ref_db, dtype=np.float64)
    assert got_db.shape == ref_db.shape
    lin_g, lin_r = 10 ** (got_db / 20), 10 ** (ref_db / 20)
    frame_max = np.maximum(lin_r.max(axis=-2, keepdims=True), 1e-6)
    assert (np.abs(lin_g - lin_r) / frame_max).max() < 3e-6
    strong = ref_db > strong_db
    if strong.any():
        assert np.abs(got_db - ref_db)[strong].max() < 2e-2


def test_mel_batch_matches_oracle():
    _need_gpu()
    from preset_gen_vae_amd.utils.audio import MelSpectrogram
    waves = np.stack([ao.synth_fm_wave(idx=i) for i in range(5)])
    mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
    out = mel.batch(torch.tensor(waves).cuda())
    assert out.shape == (5, 1, 257, 347) and out.dtype == torch.float32
    out = out.cpu().numpy()
    for i in range(5):
        _compare_db(out[i, 0], ao.mel_spectrogram_db(waves[i]))
    assert out.min() >= -120.0 - 1e-4
    # reference call convention: 1-D float64 numpy array -> Tensor[n_mels, T]
    single = mel(waves[0].astype(np.float64))
    assert tuple(single.shape) == (257, 347)
    np.testing.assert_allclose(single.cpu().numpy(), out[0, 0], atol=1e-5)
    # product filterbank == oracle filterbank (independent restatements of the Slaney construction)
    np.testing.assert_allclose(mel.mel_basis, ao.mel_filterbank(), atol=1e-6)


def test_linear_spectrogram_matches_reference_golden():
    _need_gpu()
    from preset_gen_vae_amd.utils.audio import Spectrogram
    g = load_golden('stft.npz')
    spec = Spectrogram(1024, 256, -120.0)
    assert abs(spec.spectrogram_norm_factor - float(g['norm_factor'])) < 1e-3
    np.testing.assert_allclose(spec.window.numpy(), g['window'], atol=1e-6)
    for idx in range(3):
        wav = ao.synth_fm_wave(idx=idx)
        db = spec(wav).cpu().numpy()
        assert db.shape == (513, 347)
        frames = g[f'wave{idx}/frames']
        _compare_db(db[:, frames], ao.spectrogram_db(wav)[:, frames])
        # against the float32 reference itself: both carry fp32 noise -> compare where the signal is
        ref = g[f'wave{idx}/db']
        strong = ref > -70
        assert np.abs(db[:, frames] - ref)[strong].max() < 5e-2
    short = spec(ao.synth_fm_wave(n=700, idx=5)).cpu().numpy()     # ragged input shorter than one FFT
    assert short.shape == (513, 3)
    _compare_db(short, ao.spectrogram_db(ao.synth_fm_wave(n=700, idx=5)))


def _stft_np(wav, window):
    """float64 |STFT| / 1 of one waveform: centre zero padding, hop 256, n_fft 1024 -> [513, T]."""
    n = len(wav)
    pad = np.concatenate([np.zeros(512), wav.astype(np.float64), np.zeros(512)])
    T = 1 + n // 256
    frames = np.stack([pad[256 * t:256 * t + 1024] for t in range(T)])
    return np.abs(np.fft.rfft(frames * window.astype(np.float64)[None], axis=1)).T


@pytest.mark.parametrize("case", ["slaney257", "slaney64", "slaney400", "wide16", "scattered", "empty_rows"])
def test_group_kernel_filterbanks_and_gather_path(case):
    """The second-generation front-end kernel (hop 256 + mel projection: 16-frame groups, mel phase with lanes along
    frames) against a float64 STFT + dense projection, for filterbanks that take its planned path (contiguous taps, <= 16 per
    row: 257 / 64 / 400 Slaney rows) and its gather path (rows wider than 16 taps, scattered columns, empty rows), in the
    linear and the dB output mode, at ragged lengths - and against the first-generation kernel (kernel policy 1), which
    the repository's earlier rounds pinned to the reference's spectrograms."""
    _need_gpu()
    from preset_gen_vae_amd import ops, _lib
    from preset_gen_vae_amd.utils.audio import slaney_mel_basis, dense_to_csr, Spectrogram
    rng = np.random.default_rng(5)
    if case.startswith("slaney"):
        w = slaney_mel_basis(22050, 1024, int(case[6:]))
    elif case == "wide16":
        w = slaney_mel_basis(22050, 1024, 16)              # up to ~70 taps per row
        assert (w > 0).sum(axis=1).max() > 16
    elif case == "scattered":
        w = np.zeros((96, 513), np.float32)
        for r in range(96):
            cols = rng.choice(513, size=rng.integers(1, 12), replace=False)
            w[r, cols] = rng.uniform(0.1, 1.0, len(cols)).astype(np.float32)
    else:
        w = slaney_mel_basis(22050, 1024, 120)
        w[[0, 7, 8, 63, 119]] = 0.0                        # rows without taps
    rows = w.shape[0]
    rp, col, val = (torch.tensor(a).cuda() for a in dense_to_csr(w))
    window = Spectrogram(1024, 256, -120.0).window
    win_d = window.cuda()
    lib = _lib.load()
    for n in (88576, 256 * 16 + 5, 256 * 47, 700):          # 347 frames; 17 (one full group + 1); 48 = 3 groups; 3
        waves = np.stack([ao.synth_fm_wave(n=n, idx=i) for i in range(3)])
        waves[1] += 0.05 * rng.standard_normal(n).astype(np.float32)
        x = torch.tensor(waves).cuda()
        T = 1 + n // 256
        ref = np.stack([w.astype(np.float64) @ _stft_np(waves[i], window.numpy()) for i in range(3)]) / 300.0
        for mode in (ops.STFT_LINEAR, ops.STFT_DB):
            got = {}
            for policy in (0, 1):
                lib.pgv_set_kernel_policy(policy)
                try:
                    got[policy] = ops.stft_mel(x, 256, T, win_d, 300.0, (rp, col, val), rows, 1e-6, 1.0, 0.0, mode=mode).cpu().numpy()
                finally:
                    lib.pgv_set_kernel_policy(0)
            assert got[0].shape == (3, rows, T)
            if mode == ops.STFT_LINEAR:
                scale = np.maximum(ref.max(axis=1, keepdims=True), 1e-6)
                assert (np.abs(got[0] - ref) / scale).max() < 3e-6, (case, n)
                assert (np.abs(got[0] - got[1]) / scale).max() < 1e-6, (case, n)
            else:
                _compare_db(got[0], 20 * np.log10(np.maximum(ref, 1e-6)))
                lin0, lin1 = 10 ** (got[0].astype(np.float64) / 20), 10 ** (got[1].astype(np.float64) / 20)
                # (through a float32 dB value: one ulp at 0 dB is 2e-7 dB, at -100 dB 8e-6 dB = 1e-6 relative)
                assert (np.abs(lin0 - lin1) / np.maximum(lin1.max(axis=1, keepdims=True), 1e-6)).max() < 3e-6, (case, n)


def test_group_kernel_non_finite_samples_stay_in_their_frames():
    """A NaN sample reaches the output (torch.maximum of utils/audio.py:53 keeps a NaN; fmaxf would have put the floor there) and
    poisons the frames that contain it - widened to whole frame PAIRS, since two real frames are transformed as one complex
    signal (the reference: frames 39 .. 42 only) - and nothing else: the magnitude array of a 16-frame group is shared by its
    frames, and padded taps multiply whatever lies behind a row's last bin by zero."""
    _need_gpu()
    from preset_gen_vae_amd.utils.audio import MelSpectrogram
    mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
    wav = torch.tensor(np.stack([ao.synth_fm_wave(idx=i) for i in range(2)])).cuda()
    clean = mel.batch(wav).clone()
    bad = wav.clone()
    bad[1, 256 * 40 + 3] = float('nan')
    out = mel.batch(bad)
    hit = torch.isnan(out[1, 0]).any(dim=0).nonzero().flatten().tolist()
    assert hit == [38, 39, 40, 41, 42, 43], hit
    assert torch.isnan(out[1, 0][:, 38:44]).all()
    keep = [t for t in range(347) if t not in hit]
    assert torch.equal(out[1, 0][:, keep], clean[1, 0][:, keep]) and torch.equal(out[0], clean[0])


def test_minmax_fused_and_odd_lengths():
    _need_gpu()
    from preset_gen_vae_amd.utils.audio import MelSpectrogram
    mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
    mel.set_minmax_normalization(-120.0, 3.5)
    for n in (88200, 4096, 256 * 33 + 17):        # 345 frames; 17 frames (FT+1: ragged last tile); odd tail
        wav = ao.synth_fm_wave(n=n, idx=2)
        out = mel.batch(torch.tensor(wav[None]).cuda())[0, 0].cpu().numpy()
        ref = ao.minmax_normalize(ao.mel_spectrogram_db(wav), -120.0, 3.5)
        assert out.shape == ref.shape == (257, 1 + n // 256)
        undo = (out + 1.0) * ((3.5 + 120.0) / 2.0) - 120.0
        _compare_db(undo, ao.mel_spectrogram_db(wav))
        assert out.min() >= -1.0 - 1e-5


def test_frontend_feeds_encoder_shape():
    """Raw-audio minibatch -> [B,1,257,347] is exactly what 88 576-sample Dexed renders give."""
    _need_gpu()
    from preset_gen_vae_amd.utils.audio import MelSpectrogram
    mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
    out = mel.batch(torch.zeros(3, 88576, device='cuda'))
    assert out.shape == (3, 1, 257, 347)
    assert (out + 120.0).abs().max().item() < 1e-4   # silence sits on the floor


def test_dataset_seam_f1_matches_oracle(tmp_path):
    """the batched, device-side dataset seam (item tuple contract, min-max / mean-std normalisation,
    denormalisation, statistics pass) against oracle/data_oracle.py (pinned by the reference's PresetDataset golden)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import json
    from oracle import data_oracle as do
    from preset_gen_vae_amd.data import BatchedPresetSpectrograms
    n_presets, notes, n_samples, L = 5, ((60, 85), (72, 100)), 88576 // 4, 7
    waves = np.stack([np.stack([ao.synth_fm_wave(n_samples, idx=10 * p + j) for j in range(len(notes))])
                      for p in range(n_presets)])
    params = np.random.default_rng(0).random((n_presets, L)).astype(np.float32)
    uids = np.array([1007, 23, 501, 77, 4242])
    mel = lambda x: ao.mel_spectrogram_db(x)                                  # noqa: E731
    ref_specs = [mel(waves[p, j]) for p in range(n_presets) for j in range(len(notes))]
    ref_stats = do.spectrogram_stats(ref_specs)
    for stacked, mode in ((False, 'min_max'), (True, 'min_max'), (False, 'mean_std'), (False, None)):
        ds = BatchedPresetSpectrograms(waves, params, uids, midi_notes=notes, spectrogram_normalization=mode,
                                       multichannel_stacked_spectrograms=stacked)
        assert len(ds) == do.dataset_len(n_presets, len(notes), stacked)
        stats, full = ds.compute_and_store_spectrograms_stats(json_path=tmp_path / 'stats.json', batch_size=4)
        assert json.load(open(tmp_path / 'stats.json')).keys() == {'min', 'max', 'mean', 'std'}
        assert len(full['min']) == n_presets * len(notes)
        # float32 front-end vs float64 oracle: dB values agree to ~1e-3 dB above the floor
        assert abs(stats['min'] - ref_stats['min']) < 1e-3 and abs(stats['max'] - ref_stats['max']) < 5e-3
        assert abs(stats['mean'] - ref_stats['mean']) < 2e-2 and abs(stats['std'] - ref_stats['std']) < 2e-2
        ds.set_spec_stats(ref_stats)
        idx = list(range(len(ds)))[::-1]
        spec, par, info, labels = ds.get_batch(idx)
        assert spec.is_cuda and spec.shape[:2] == (len(idx), len(notes) if stacked else 1)
        assert info.dtype == torch.int32 and info.shape == (len(idx), 3)
        assert labels.dtype == torch.int8 and labels.shape == (len(idx), 1) and bool((labels == 1).all())
        for b, i in enumerate(idx):
            r_spec, r_par, r_info, _ = do.get_item(i, waves, params, uids, notes, ref_stats, mode, stacked, mel)
            assert np.array_equal(info[b].numpy(), r_info)
            assert np.array_equal(par[b].numpy(), r_par)
            got_db = do.denormalize_spectrogram(spec[b].double().cpu().numpy(), ref_stats, mode)
            ref_db = do.denormalize_spectrogram(r_spec, ref_stats, mode)
            strong = ref_db > -80.0
            assert np.abs(got_db - ref_db)[strong].max() < 2e-2
            assert np.abs(got_db - ref_db).max() < 1.0
        if mode is not None:
            back = ds.denormalize_spectrogram(spec)
            assert torch.allclose(ds.normalize_spectrogram(back), spec, atol=1e-5)
    with pytest.raises(RuntimeError):
        BatchedPresetSpectrograms(waves, params, uids, midi_notes=notes).get_batch([0])


def test_get_stft_linear_scale_and_dynamic_range_match_reference():
    """SURVEY 8b front-end surface: Spectrogram.get_stft (complex), log_scale=False, linear_to_log_scale_with_dynamic_range
    (utils/audio.py:33-50, :63-69) against the reference's own outputs (tests/golden/stft.npz) and the numpy oracle."""
    _need_gpu()
    from oracle import audio_oracle as ao
    from preset_gen_vae_amd.utils.audio import MelSpectrogram, Spectrogram
    g = load_golden('stft.npz')
    spec = Spectrogram(1024, 256, -120.0)
    for idx in range(3):
        wav = ao.synth_fm_wave(idx=idx)
        z = spec.get_stft(wav.astype(np.float64))
        assert z.dtype == torch.complex64 and tuple(z.shape) == (513, 347)
        frames = g[f'wave{idx}/frames']
        ref = g[f'wave{idx}/stft_re'] + 1j * g[f'wave{idx}/stft_im']
        got = z.cpu().numpy()[:, frames]
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
        full = ao.stft_complex(wav)                                  # float64 oracle, every frame
        assert np.abs(z.cpu().numpy() - full).max() <= 2e-5 * np.abs(full).max()
    zb = spec.get_stft(torch.tensor(np.stack([ao.synth_fm_wave(idx=i) for i in range(3)])))
    assert tuple(zb.shape) == (3, 513, 347)
    assert np.abs(zb[2].cpu().numpy() - ao.stft_complex(ao.synth_fm_wave(idx=2))).max() < 2e-5 * 512
    # log_scale=False: the normalised amplitudes
    wav = ao.synth_fm_wave(idx=1)
    lin = Spectrogram(1024, 256, -120.0, log_scale=False)(wav.astype(np.float64))
    frames = g['linear/frames']
    assert np.abs(lin.cpu().numpy()[:, frames] - g['linear/mag']).max() <= 2e-5 * g['linear/mag'].max()
    assert np.abs(lin.cpu().numpy() - ao.spectrogram_mag(wav)).max() <= 2e-5 * g['linear/mag'].max()
    dyn = Spectrogram(1024, 256, -120.0, dynamic_range_dB=60.0).linear_to_log_scale_with_dynamic_range(lin)
    assert np.abs(dyn.cpu().numpy()[:, frames] - g['dynrange/db']).max() < 2e-2
    with pytest.raises(AssertionError):
        spec.linear_to_log_scale_with_dynamic_range(lin)             # no dynamic range given (reference asserts too)
    with pytest.raises(NotImplementedError, match="librosa"):
        MelSpectrogram(1024, 256, -120.0, 257, 22050).mel_dB_to_STFT(torch.zeros(257, 4))
