"""Phase-toggle timing of the deep-layer kernels: PGV_DBG_LIB selects a variant library built with -DPGV_DEEP_EXP=n."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from preset_gen_vae_amd import _lib, ops
if os.environ.get('PGV_DBG_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'scratch', os.environ['PGV_DBG_LIB'])
B = 256
which = sys.argv[1:] or ['down']
out = os.environ.get('PGV_DBG_LIB', 'normal') + ':'
for nm, (Cb, Cs, Hb, Wb) in {'G5': (64, 128, 17, 23), 'G6': (128, 256, 9, 12), 'G7': (256, 512, 5, 7)}.items():
    g = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, 4, 4, device='cuda') * 0.05
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    sc, sh = 1 + 0.1 * torch.randn(Cb, device='cuda'), 0.1 * torch.randn(Cb, device='cuda')
    scs, shs = 1 + 0.1 * torch.randn(Cs, device='cuda'), 0.1 * torch.randn(Cs, device='cuda')
    st_s = torch.empty(2 * Cs, device='cuda', dtype=torch.float64); st_b = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
    bs, bb = torch.zeros(Cs, device='cuda'), torch.zeros(Cb, device='cuda')
    gw = torch.empty_like(w)
    for mode in (os.environ.get('MODES', 'fp32').split(',')):
        ops.set_compute_dtype(mode)
        if 'down' in which:
            t = bench.time_kernel(lambda: ops.conv_down(g, big, w, bs, 1, 0.1, in_scale=sc, in_shift=sh, stats=st_s), iters=5)
            out += f" {nm} {mode} down {t*1e3:6.1f}"
        if 'up' in which:
            t = bench.time_kernel(lambda: ops.conv_up(g, small, w, bb, 1, 0.1, in_scale=scs, in_shift=shs, stats=st_b), iters=5)
            out += f" {nm} {mode} up {t*1e3:6.1f}"
        if 'wgrad' in which:
            t = bench.time_kernel(lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh), iters=5)
            out += f" {nm} {mode} wgrad {t*1e3:6.1f}"
    ops.set_compute_dtype('fp32')
print(out, flush=True)
