"""Time of the STFT -> mel front-end kernel for 256 waveforms (graph replay, cold operands, as bench.py measures)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
if os.environ.get('PGV_ALT_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'scratch', os.environ['PGV_ALT_LIB'])
import bench
from preset_gen_vae_amd.utils.audio import MelSpectrogram
fe = MelSpectrogram(1024, 256, -120.0, 257, 22050)
fe.set_minmax_normalization(-120.0, 3.5)
wav = 0.3 * torch.randn(256, 88576, device='cuda')
out = torch.empty(256, 1, 257, 347, device='cuda')
fn = lambda: fe.batch(wav, out=out)
fn()
ts = [bench.time_kernel(fn, iters=5) * 1e3 for _ in range(3)]
print('stft_mel 256 waveforms: %.1f us (runs: %s)' % (min(ts), ', '.join('%.1f' % t for t in ts)))
