"""The 1 <-> 8 channel 5x5 end-layer launches of the step, a few repetitions each (driver of scratch/pmc_c1.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
B = 256
g = ops.ConvGeom(1, 8, 5, 2, 2, 257, 347)
big = torch.randn(B, 1, 257, 347, device='cuda'); small = torch.randn(B, 8, g.Hs, g.Ws, device='cuda')
w = torch.randn(8, 1, 5, 5, device='cuda') * 0.1
b8, b1 = torch.zeros(8, device='cuda'), torch.zeros(1, device='cuda')
sc, sh = torch.ones(8, device='cuda'), torch.zeros(8, device='cuda')
C8 = ops.CLS_COPIES
a = small * 1.3 + 0.1
coef = torch.cat([torch.ones(8, device='cuda'), torch.zeros(16, device='cuda')])
gw = torch.empty_like(w)
for _ in range(4):
    ops.conv_down(g, big, w, b8, 1, 0.1)
    ops.conv_up(g, small, w, b1, 2, 0.0, in_scale=sc, in_shift=sh)
    ops.conv_down(g, big, w, None, 0, 0.0, bwd_fuse=(a, coef, torch.zeros(C8 * 8, device='cuda'), 1, 0.1, torch.zeros(C8 * 32, device='cuda'), C8))
    ops.conv_wgrad(g, big, small, gw)
torch.cuda.synchronize()
