import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from preset_gen_vae_amd import ops
torch.manual_seed(0)
B, Cb, Cs, Hb, Wb = 2, 8, 16, 129, 174
g = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, 4, 4, device='cuda') * 0.1
bias = torch.randn(Cs, device='cuda')
ref = F.conv2d(big.double(), w.double(), bias.double(), stride=2, padding=2)
got = ops.conv_down(g, big, w, bias, 0, 0.0)
bad = ((got.double() - ref).abs() > 1e-4).nonzero()
print('bad count', bad.shape[0], 'of', ref.numel())
import collections
print('by (b)', collections.Counter(bad[:, 0].tolist()))
print('by channel', collections.Counter(bad[:, 1].tolist()))
print('by row', sorted(collections.Counter(bad[:, 2].tolist()).items())[:70])
print('by col', sorted(collections.Counter(bad[:, 3].tolist()).items())[:20])
print(bad[:10].tolist())
print(got[tuple(bad[0].tolist())].item(), ref[tuple(bad[0].tolist())].item())
