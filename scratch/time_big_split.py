"""Large-plane k4 s2 p2 layers of the 4-layer stack: products as six bf16 instructions (conv_big_split.hip,
ops.set_fp32_products('bf16x6')) against the native fp32 kernels - forward form and fused input gradient of both directions,
cold operands (a 512 MB read between launches), error against float64 on a few samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import preset_gen_vae_amd  # noqa
if os.environ.get('PGV_LIB'):
    from preset_gen_vae_amd import _lib
    _lib.LIB_PATH = os.environ['PGV_LIB']
from preset_gen_vae_amd import ops

B = int(os.environ.get('B', 256))
flush = torch.empty(128 * 1024 * 1024, device='cuda')


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    tot = 0.0
    for _ in range(n):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1000


C8 = ops.CLS_COPIES
for Cb, Cs, Hb, Wb in ((8, 16, 129, 174), (16, 32, 65, 88), (32, 64, 33, 45)):
    g = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
    torch.manual_seed(1)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda')
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    w = torch.randn(Cs, Cb, 4, 4, device='cuda') * 0.05
    bias_b, bias_s = torch.randn(Cb, device='cuda') * 0.1, torch.randn(Cs, device='cuda') * 0.1
    ssc, ssh = torch.rand(Cs, device='cuda') + 0.5, torch.randn(Cs, device='cuda') * 0.1
    bsc, bsh = torch.rand(Cb, device='cuda') + 0.5, torch.randn(Cb, device='cuda') * 0.1
    a_b, a_s = big * 1.3 + 0.1, small * 1.3 + 0.1
    cf = lambda C: torch.cat([1.0 + 0.3 * torch.rand(C, device='cuda'), 0.05 * torch.randn(C, device='cuda'), 0.02 * torch.randn(C, device='cuda')])
    coef_b, coef_s = cf(Cb), cf(Cs)
    nb = 4
    oph, opw = Hb - ((g.Hs - 1) * 2 - 4 + 4), Wb - ((g.Ws - 1) * 2 - 4 + 4)
    ref_u = F.conv_transpose2d(small[:nb].double(), w.double(), None, stride=2, padding=2, output_padding=(oph, opw))
    ref_d = F.conv2d(big[:nb].double(), w.double(), None, stride=2, padding=2)
    for mode in ('native', 'bf16x6'):
        ops.set_fp32_products(mode)
        sh = ops.conv_weight_shadow(g, w)
        kw = dict(w_shadow=sh) if sh is not None else {}
        ou = ops.conv_up(g, small[:nb].contiguous(), w, None, ops.PGV_ACT_NONE, 0.0, **kw)
        od = ops.conv_down(g, big[:nb].contiguous(), w, None, ops.PGV_ACT_NONE, 0.0, **kw)
        eu = ((ou.double() - ref_u).norm() / ref_u.norm()).item()
        ed = ((od.double() - ref_d).norm() / ref_d.norm()).item()
        stb = torch.zeros(C8 * 2 * Cb, device='cuda', dtype=torch.float64)
        sts = torch.zeros(C8 * 2 * Cs, device='cuda', dtype=torch.float64)
        t_uf = timeit(lambda: ops.conv_up(g, small, w, bias_b, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=ssc, in_shift=ssh, stats=stb,
                                          prezeroed=True, stats_copies=True, **kw))
        t_df = timeit(lambda: ops.conv_down(g, big, w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=bsc, in_shift=bsh, stats=sts,
                                            prezeroed=True, stats_copies=True, **kw))
        gbb, gbs = torch.zeros(C8 * Cb, device='cuda'), torch.zeros(C8 * Cs, device='cuda')
        cls = torch.zeros(C8 * 4 * Cs, device='cuda')
        t_ub = timeit(lambda: ops.conv_up(g, small, w, None, ops.PGV_ACT_NONE, 0.0,
                                          bwd_fuse=(a_b, coef_b, gbb, ops.PGV_ACT_LEAKY_RELU, 0.1, None, C8), **kw))
        t_db = timeit(lambda: ops.conv_down(g, big, w, None, ops.PGV_ACT_NONE, 0.0,
                                            bwd_fuse=(a_s, coef_s, gbs, ops.PGV_ACT_LEAKY_RELU, 0.1, cls, C8), **kw))
        print(f'{Hb}x{Wb} {Cb}<->{Cs} {mode:7s}: down fwd {t_df:6.1f} fused+cls {t_db:6.1f} | up fwd {t_uf:6.1f} fused {t_ub:6.1f} us'
              f' | rel L2 vs float64 down {ed:.2e} up {eu:.2e}', flush=True)
    ops.set_fp32_products('native')

# ---- weight gradients
for Cb, Cs, Hb, Wb in ((8, 16, 129, 174), (16, 32, 65, 88), (32, 64, 33, 45)):
    g = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
    torch.manual_seed(1)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda')
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    bsc, bsh = torch.rand(Cb, device='cuda') + 0.5, torch.randn(Cb, device='cuda') * 0.1
    nb = 4
    wv = torch.zeros(Cs, Cb, 4, 4, dtype=torch.float64, device='cuda', requires_grad=True)
    F.conv2d(big[:nb].double(), wv, None, stride=2, padding=2).backward(small[:nb].double())
    for mode in ('native', 'bf16x6'):
        ops.set_fp32_products(mode)
        gw = torch.empty(Cs, Cb, 4, 4, device='cuda')
        ops.conv_wgrad(g, big[:nb].contiguous(), small[:nb].contiguous(), gw)
        err = ((gw.double() - wv.grad).norm() / wv.grad.norm()).item()
        gwf = torch.zeros(Cs, Cb, 4, 4, device='cuda')
        t = timeit(lambda: ops.conv_wgrad(g, big, small, gwf, big_scale=bsc, big_shift=bsh))
        print(f'{Hb}x{Wb} {Cb}<->{Cs} {mode:7s}: weight gradient (+ reduce launch) {t:6.1f} us | rel L2 vs float64 {err:.2e}', flush=True)
    ops.set_fp32_products('native')
