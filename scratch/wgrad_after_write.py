import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preset_gen_vae_amd import ops
B = 256
Cb, Cs, k, Hb, Wb = 8, 16, 4, 129, 174
g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
sc, sh = torch.ones(Cb, device='cuda'), torch.zeros(Cb, device='cuda')
scs, shs = torch.ones(Cs, device='cuda'), torch.zeros(Cs, device='cuda')
gw = torch.empty_like(w)
big = torch.randn(B, Cb, Hb, Wb, device='cuda'); small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
other = torch.randn(B, Cb, Hb, Wb, device='cuda')
flush = torch.empty(128 << 20, device='cuda')
def t(pre, fn, n=10):
    fn(); torch.cuda.synchronize(); tot = 0
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
f_enc = lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh)
f_dec = lambda: ops.conv_wgrad(g, big, small, gw, small_scale=scs, small_shift=shs)
f_none = lambda: ops.conv_wgrad(g, big, small, gw)
for nm, f in [('enc-style (big affine)', f_enc), ('dec-style (small affine)', f_dec), ('no affine', f_none)]:
    print(f"{nm:26s} nothing before {t(lambda: None, f):6.1f} | after flush {t(lambda: flush.fill_(1.0), f):6.1f} | "
          f"after writing big {t(lambda: big.mul_(1.0), f):6.1f} | after writing small {t(lambda: small.mul_(1.0), f):6.1f} | "
          f"after writing other {t(lambda: other.mul_(1.0), f):6.1f}")
