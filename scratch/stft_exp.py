import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
if os.environ.get('PGV_DBG_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'scratch', os.environ['PGV_DBG_LIB'])
from preset_gen_vae_amd.utils.audio import MelSpectrogram
mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
x = torch.randn(256, 88576, device='cuda') * 0.1
for _ in range(3): mel.batch(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): mel.batch(x)
e1.record(); torch.cuda.synchronize()
print(os.environ.get('PGV_DBG_LIB', 'normal'), f"{e0.elapsed_time(e1)/20*1e3:.1f} us")
