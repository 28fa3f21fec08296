import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops, _lib
from preset_gen_vae_amd.utils.audio import MelSpectrogram
mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
n = 88576
t = torch.arange(n, device='cuda', dtype=torch.float64)
x = torch.sin(2 * torch.pi * 100.3 / 1024 * t).float()[None].repeat(2, 1)
x[1] = torch.randn(n, device='cuda') * 0.1
lib = _lib.load()
win = mel.window.cuda()
NR = 513
rp = torch.arange(NR + 1, dtype=torch.int32, device='cuda')
col = torch.arange(NR, dtype=torch.int32, device='cuda')
val = torch.ones(NR, device='cuda')
nf = 1 + n // 256
def run(pol):
    lib.pgv_set_kernel_policy(pol)
    o = ops.stft_mel(x, 256, nf, win, 1.0, (rp, col, val), NR, 1e-6, 1.0, 0.0, mode=ops.STFT_LINEAR).clone()
    lib.pgv_set_kernel_policy(0)
    return o
a, b = run(1), run(0)
d = (a - b).abs()
print('max', d.amax(dim=(1, 2)).tolist(), 'ref max', a.amax(dim=(1,2)).tolist())
for s in range(2):
    rel = d[s] / a[s].amax()
    bad = (rel > 1e-5)
    print('sample', s, 'bad count', int(bad.sum()), 'of', bad.numel())
    print(' bad per frame parity: even', int(bad[:, 0::2].sum()), 'odd', int(bad[:, 1::2].sum()))
    byk = bad.sum(dim=1)
    print(' bad by bin (k: count) first 40 nonzero:', [(int(k), int(c)) for k, c in enumerate(byk.tolist()) if c][:40])
    print(' bad by lane (k&63):', torch.stack([byk[l::64].sum() for l in range(64)]).tolist())
    print(' bad by q (k>>6):', [int(byk[64*q:64*q+64].sum()) for q in range(9)])
    fr = 40
    print(' old f40 k 96..106', [round(v, 3) for v in a[s, 96:106, fr].tolist()])
    print(' new f40 k 96..106', [round(v, 3) for v in b[s, 96:106, fr].tolist()])
    print(' old f41 k 96..106', [round(v, 3) for v in a[s, 96:106, fr+1].tolist()])
    print(' new f41 k 96..106', [round(v, 3) for v in b[s, 96:106, fr+1].tolist()])
