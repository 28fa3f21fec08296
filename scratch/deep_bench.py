"""Standalone timing of the deep-layer conv kernels (graph-differential, cold cache) in both operand precisions."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from preset_gen_vae_amd import ops
B = 256
which = sys.argv[1:] or ['down', 'up', 'wgrad']
for nm, (Cb, Cs, Hb, Wb) in {'G5': (64, 128, 17, 23), 'G6': (128, 256, 9, 12), 'G7': (256, 512, 5, 7), 'G8': (512, 2048, 3, 4)}.items():
    kk = 1 if nm == 'G8' else 4
    g = ops.ConvGeom(Cb, Cs, kk, 2 if kk == 4 else 1, 2 if kk == 4 else 0, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, kk, kk, device='cuda') * 0.05
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    sc, sh = 1 + 0.1 * torch.randn(Cb, device='cuda'), 0.1 * torch.randn(Cb, device='cuda')
    scs, shs = 1 + 0.1 * torch.randn(Cs, device='cuda'), 0.1 * torch.randn(Cs, device='cuda')
    st_s = torch.empty(2 * Cs, device='cuda', dtype=torch.float64); st_b = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
    bs, bb = torch.zeros(Cs, device='cuda'), torch.zeros(Cb, device='cuda')
    gw = torch.empty_like(w)
    flops = 2.0 * B * Cs * g.Hs * g.Ws * Cb * kk * kk
    for mode in ('fp32', 'bf16'):
        ops.set_compute_dtype(mode)
        line = f"{nm} {mode}:"
        if 'down' in which:
            t = bench.time_kernel(lambda: ops.conv_down(g, big, w, bs, 1, 0.1, in_scale=sc, in_shift=sh, stats=st_s), iters=5)
            line += f" down {t*1e3:7.1f} us {flops/t/1e9:6.1f} TF |"
        if 'up' in which:
            t = bench.time_kernel(lambda: ops.conv_up(g, small, w, bb, 1, 0.1, in_scale=scs, in_shift=shs, stats=st_b), iters=5)
            line += f" up {t*1e3:7.1f} us {flops/t/1e9:6.1f} TF |"
        if 'wgrad' in which:
            t = bench.time_kernel(lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh), iters=5)
            line += f" wgrad {t*1e3:7.1f} us {flops/t/1e9:6.1f} TF"
        print(line, flush=True)
    ops.set_compute_dtype('fp32')
