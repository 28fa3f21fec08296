#!/usr/bin/env python
"""Auxiliary benchmark for BASELINE.json configs[4]: the on-GPU STFT -> mel -> dB (+min-max) front-end on raw-audio
minibatches (88 576-sample synthetic FM voices), waveforms/s and algorithmic HBM GB/s (0.711 MB per spectrogram:
88 576 x 4 B in + 257 x 347 x 4 B out), next to the numpy oracle on the host (reference-style per-item
loop, data/abstractbasedataset.py:126-134).  Not the headline metric: bench.py is."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-items", type=int, default=8)
    args = ap.parse_args()
    from preset_gen_vae_amd.utils.audio import MelSpectrogram
    from preset_gen_vae_amd.utils.synthetic import fm_voice
    waves = np.stack([fm_voice(idx=i) for i in range(16)])
    wav = torch.tensor(np.tile(waves, (args.batch // 16 + 1, 1))[:args.batch], device='cuda')
    mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
    mel.set_minmax_normalization(-120.0, 0.0)
    out = mel.batch(wav)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        out = mel.batch(wav)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    per = wav.shape[1] * 4 + 257 * out.shape[-1] * 4
    from oracle import audio_oracle as ao   # CPU baseline leg only
    t0 = time.perf_counter()
    for i in range(args.cpu_items):
        ao.minmax_normalize(ao.mel_spectrogram_db(waves[i % 16], dtype=np.float32), -120.0, 0.0)
    cpu = args.cpu_items / (time.perf_counter() - t0)
    print(json.dumps({"metric": "waveforms/sec STFT->mel->dB front-end (88576 samples -> 1x257x347)",
                      "value": round(args.batch / ms * 1e3, 1), "unit": "waveforms/s", "batch": args.batch,
                      "ms_per_batch": round(ms, 4),
                      "roofline": {"bound": "hbm", "achieved": round(per * args.batch / ms / 1e6, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(per * args.batch / ms / 1e6 / 8000.0, 5)},
                      "cpu_baseline": {"value": round(cpu, 2), "unit": "waveforms/s", "cores": 1, "kind": "port",
                                       "sample": f"{args.cpu_items} items, numpy float32 per-item loop"}}))


if __name__ == "__main__":
    main()
