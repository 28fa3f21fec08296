"""Robustness sweep: the three conv products of the stride-2 k=4 layers at odd batch sizes against torch (MIOpen) fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from preset_gen_vae_amd import ops
torch.manual_seed(1)
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm()).item()
worst = 0.0
for (Cb, Cs, Hb, Wb) in [(8, 16, 129, 174), (16, 32, 65, 88), (32, 64, 33, 45), (1, 8, 257, 347)]:
    k = 5 if Cb == 1 else 4
    for B in (1, 5, 19, 41, 64, 100, 257):
        g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
        big = torch.randn(B, Cb, Hb, Wb, device='cuda'); small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
        w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.1
        bs, bb = torch.randn(Cs, device='cuda'), torch.randn(Cb, device='cuda')
        sc_b, sh_b = torch.rand(Cb, device='cuda') + 0.5, torch.randn(Cb, device='cuda') * 0.1
        sc_s, sh_s = torch.rand(Cs, device='cuda') + 0.5, torch.randn(Cs, device='cuda') * 0.1
        aff = lambda t, sc, sh: t * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        st = torch.zeros(2 * Cs, device='cuda', dtype=torch.float64)
        ref = F.leaky_relu(F.conv2d(aff(big, sc_b, sh_b), w, bs, stride=2, padding=2), 0.1)
        got = ops.conv_down(g, big, w, bs, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc_b, in_shift=sh_b, stats=st)
        e1 = rel(got, ref); e1s = rel(st, torch.cat([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))]))
        ref = F.conv2d(big, w, None, stride=2, padding=2)
        e2 = rel(ops.conv_down(g, big, w, None, ops.PGV_ACT_NONE, 0.0), ref)
        oph, opw = Hb - ((g.Hs - 1) * 2 - 4 + k), Wb - ((g.Ws - 1) * 2 - 4 + k)
        ref = F.leaky_relu(F.conv_transpose2d(aff(small, sc_s, sh_s), w, bb, stride=2, padding=2, output_padding=(oph, opw)), 0.1)
        stb = torch.zeros(2 * Cb, device='cuda', dtype=torch.float64)
        got = ops.conv_up(g, small, w, bb, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc_s, in_shift=sh_s, stats=stb)
        e3 = rel(got, ref); e3s = rel(stb, torch.cat([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))]))
        ref = F.conv_transpose2d(small, w, None, stride=2, padding=2, output_padding=(oph, opw))
        e4 = rel(ops.conv_up(g, small, w, None, ops.PGV_ACT_NONE, 0.0), ref)
        wv = w.clone().requires_grad_(True)
        F.conv2d(aff(big, sc_b, sh_b), wv, None, stride=2, padding=2).backward(small)
        gw = torch.empty_like(w); ops.conv_wgrad(g, big, small, gw, big_scale=sc_b, big_shift=sh_b)
        e5 = rel(gw, wv.grad)
        wv = w.clone().requires_grad_(True)
        F.conv2d(big, wv, None, stride=2, padding=2).backward(aff(small, sc_s, sh_s))
        ops.conv_wgrad(g, big, small, gw, small_scale=sc_s, small_shift=sh_s)
        e6 = rel(gw, wv.grad)
        m = max(e1, e1s, e2, e3, e3s, e4, e5, e6)
        worst = max(worst, m)
        flag = '' if m < 3e-4 else '   <-- CHECK'
        print(f"{Cb:3d}->{Cs:3d} {Hb}x{Wb} B={B:3d}: down {e1:.1e}/{e1s:.1e} dgrad-down {e2:.1e} up {e3:.1e}/{e3s:.1e} dgrad-up {e4:.1e} wgrad {e5:.1e} {e6:.1e}{flag}", flush=True)
print('worst', worst)
