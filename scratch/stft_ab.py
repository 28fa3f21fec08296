"""Front-end kernel A/B: second-generation kernel (policy 0) against the first (policy 1): max difference and time."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops, _lib
from preset_gen_vae_amd.utils.audio import MelSpectrogram

mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
torch.manual_seed(0)
B = int(os.environ.get('B', 256))
x = torch.randn(B, 88576, device='cuda') * 0.1
t = torch.arange(88576, device='cuda') / 22050
x[:, :] += torch.sin(2 * torch.pi * 440 * t)[None] * torch.linspace(0, 1, B, device='cuda')[:, None]


def run(policy):
    _lib.load().pgv_set_kernel_policy(policy)
    out = mel.batch(x).clone()
    for _ in range(3): mel.batch(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): mel.batch(x)
    e1.record(); torch.cuda.synchronize()
    _lib.load().pgv_set_kernel_policy(0)
    return out, e0.elapsed_time(e1) / 20 * 1e3


o1, t1 = run(1)
o0, t0 = run(0)
d = (o0 - o1).abs()
print(f"first kernel {t1:.1f} us, second {t0:.1f} us; max |diff| {d.max().item():.3e} dB, mean {d.mean().item():.3e}, "
      f"finite {bool(torch.isfinite(o0).all())}, mismatches > 1e-3: {(d > 1e-3).sum().item()}")
for n in (256 * 5 + 17, 1024, 88576 - 3):
    y = x[:3, :n].contiguous()
    _lib.load().pgv_set_kernel_policy(1); a = mel.batch(y).clone(); _lib.load().pgv_set_kernel_policy(0); b_ = mel.batch(y)
    print(n, tuple(b_.shape), (a - b_).abs().max().item())
