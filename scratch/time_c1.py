"""Timing of the 1 <-> 8 channel 5x5 launches (graph replay, cold operands): dec8 forward, enc1 forward, dec8 input
gradient plain / fused."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from preset_gen_vae_amd import ops
B = 256
dev = torch.device('cuda', 0)
g = ops.ConvGeom(1, 8, 5, 2, 2, 257, 347)
big = torch.randn(B, 1, 257, 347, device=dev); small = torch.randn(B, 8, 129, 174, device=dev)
w = torch.randn(8, 1, 5, 5, device=dev) * 0.1
b1, b8 = torch.zeros(1, device=dev), torch.zeros(8, device=dev)
sc, sh = torch.ones(8, device=dev), torch.zeros(8, device=dev)
ob, os_ = torch.empty_like(big), torch.empty_like(small)
a = torch.randn_like(small); coef = torch.cat([torch.ones(8, device=dev), 0.01 * torch.randn(16, device=dev)])
gb, cls = torch.zeros(8, device=dev), torch.zeros(32, device=dev)
rows = [('up  (dec8 fwd, hardtanh, folded BN)', lambda: ops.conv_up(g, small, w, b1, 2, 0.0, in_scale=sc, in_shift=sh, out=ob), 275),
        ('down (enc1 fwd, leaky)', lambda: ops.conv_down(g, big, w, b8, 1, 0.1, out=os_), 275),
        ('down plain (dgrad dec8)', lambda: ops.conv_down(g, big, w, None, 0, 0.0, out=os_), 275),
        ('down fused (dgrad dec8 + bn/act bwd + cls)', lambda: ops.conv_down(g, big, w, None, 0, 0.0, out=os_, bwd_fuse=(a, coef, gb, 1, 0.1, cls)), 459)]
for label, fn, mb in rows:
    us = bench.time_kernel(fn, iters=5) * 1e3
    print(f"{label:46s} {us:7.1f} us  {mb / us * 1e-0:6.2f} TB/s" if False else f"{label:46s} {us:7.1f} us  {mb / us:5.2f} TB/s", flush=True)
