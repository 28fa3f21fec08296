"""Cost of the end-of-kernel reduction atomics: forward convs with and without BatchNorm statistics (graph-differential)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from preset_gen_vae_amd import ops
B = 256
dev = torch.device('cuda', 0)
for (Cb, Cs, k, Hb, Wb) in ((8, 16, 4, 129, 174), (16, 32, 4, 65, 88), (32, 64, 4, 33, 45), (64, 128, 4, 17, 23), (128, 256, 4, 9, 12), (256, 512, 4, 5, 7)):
    geom = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device=dev); small = torch.randn(B, Cs, geom.Hs, geom.Ws, device=dev)
    w = torch.randn(Cs, Cb, k, k, device=dev) * 0.05
    bs, bb = torch.zeros(Cs, device=dev), torch.zeros(Cb, device=dev)
    sc_b, sh_b = torch.ones(Cb, device=dev), torch.zeros(Cb, device=dev)
    sc_s, sh_s = torch.ones(Cs, device=dev), torch.zeros(Cs, device=dev)
    st_s = torch.zeros(2 * Cs, device=dev, dtype=torch.float64); st_b = torch.zeros(2 * Cb, device=dev, dtype=torch.float64)
    out_s, out_b = torch.empty_like(small), torch.empty_like(big)
    st_s8 = torch.zeros(8 * 2 * Cs, device=dev, dtype=torch.float64); st_b8 = torch.zeros(8 * 2 * Cb, device=dev, dtype=torch.float64)
    for name, fn in (("down fwd, no stats", lambda: ops.conv_down(geom, big, w, bs, 1, 0.1, in_scale=sc_b, in_shift=sh_b, out=out_s)),
                     ("down fwd, stats   ", lambda: ops.conv_down(geom, big, w, bs, 1, 0.1, in_scale=sc_b, in_shift=sh_b, stats=st_s, out=out_s, prezeroed=True)),
                     ("down fwd, copies  ", lambda: ops.conv_down(geom, big, w, bs, 1, 0.1, in_scale=sc_b, in_shift=sh_b, stats=st_s8, out=out_s, prezeroed=True, stats_copies=True)),
                     ("up fwd, copies    ", lambda: ops.conv_up(geom, small, w, bb, 1, 0.1, in_scale=sc_s, in_shift=sh_s, stats=st_b8, out=out_b, prezeroed=True, stats_copies=True)),
                     ("up fwd, no stats  ", lambda: ops.conv_up(geom, small, w, bb, 1, 0.1, in_scale=sc_s, in_shift=sh_s, out=out_b)),
                     ("up fwd, stats     ", lambda: ops.conv_up(geom, small, w, bb, 1, 0.1, in_scale=sc_s, in_shift=sh_s, stats=st_b, out=out_b, prezeroed=True))):
        print(f"{Hb}x{Wb} {name} {bench.time_kernel(fn, iters=5) * 1e3:7.1f} us", flush=True)
