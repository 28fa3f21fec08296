"""One split-product deep kernel under rocprofv3 --pmc: WHAT = down | up | wgrad, LAYER = 0..3 (17x23, 9x12, 5x7, 1x1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
B = 256
WHAT = os.environ.get('WHAT', 'down'); L = int(os.environ.get('LAYER', 2)); MODE = os.environ.get('MODE', 'bf16x6')
Cb, Cs, H, W = [(64, 128, 17, 23), (128, 256, 9, 12), (256, 512, 5, 7), (512, 2048, 3, 4)][L]
g = ops.ConvGeom(Cb, Cs, 1, 1, 0, H, W) if H == 3 else ops.ConvGeom(Cb, Cs, 4, 2, 2, H, W)
k = 1 if H == 3 else 4
torch.manual_seed(1)
big = torch.randn(B, Cb, H, W, device='cuda'); small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
ops.set_fp32_products(MODE)
sh = ops.conv_weight_shadow(g, w)
kw = dict(w_shadow=sh) if sh is not None else {}
gw = torch.empty_like(w)
for _ in range(6):
    if WHAT == 'down': ops.conv_down(g, big, w, None, ops.PGV_ACT_NONE, 0.0, **kw)
    elif WHAT == 'up': ops.conv_up(g, small, w, None, ops.PGV_ACT_NONE, 0.0, **kw)
    else: ops.conv_wgrad(g, big, small, gw)
torch.cuda.synchronize()
