import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preset_gen_vae_amd import ops
B = 256
def r(t): return t.bfloat16().float()
def t_(fn, n=20):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for nm, (Cb, Cs, Hb, Wb) in {'L2': (8, 16, 129, 174), 'L3': (16, 32, 65, 88), 'L4': (32, 64, 33, 45)}.items():
    g = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, 4, 4, device='cuda') * 0.05
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    sc, sh = 1 + 0.1 * torch.randn(Cb, device='cuda'), 0.1 * torch.randn(Cb, device='cuda')
    scs, shs = 1 + 0.1 * torch.randn(Cs, device='cuda'), 0.1 * torch.randn(Cs, device='cuda')
    res = {}
    for mode in ('fp32', 'bf16'):
        ops.set_compute_dtype(mode)
        rr = r if mode == 'bf16' else (lambda t: t)
        out = ops.conv_down(g, big, w, None, 0, 0.0, in_scale=sc, in_shift=sh)
        ref = F.conv2d(rr(big * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).double(), rr(w).double(), None, stride=2, padding=2)
        e_d = ((out.double() - ref).norm() / ref.norm()).item()
        td = t_(lambda: ops.conv_down(g, big, w, None, 0, 0.0, in_scale=sc, in_shift=sh))
        outu = ops.conv_up(g, small, w, None, 0, 0.0, in_scale=scs, in_shift=shs)
        oph, opw = Hb - ((g.Hs - 1) * 2 - 4 + 4), Wb - ((g.Ws - 1) * 2 - 4 + 4)
        refu = F.conv_transpose2d(rr(small * scs.view(1, -1, 1, 1) + shs.view(1, -1, 1, 1)).double(), rr(w).double(), None, stride=2, padding=2, output_padding=(oph, opw))
        e_u = ((outu.double() - refu).norm() / refu.norm()).item()
        tu = t_(lambda: ops.conv_up(g, small, w, None, 0, 0.0, in_scale=scs, in_shift=shs))
        gw = torch.empty_like(w)
        ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh)
        wv = rr(w).double().clone().requires_grad_(True)
        F.conv2d(rr(big * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).double(), wv, None, stride=2, padding=2).backward(rr(small).double())
        e_w = ((gw.double() - wv.grad).norm() / wv.grad.norm()).item()
        tw = t_(lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh))
        print(f"{nm} {mode}: down err {e_d:.2e} {td:6.1f} us | up err {e_u:.2e} {tu:6.1f} us | wgrad err {e_w:.2e} {tw:6.1f} us")
    ops.set_compute_dtype('fp32')
