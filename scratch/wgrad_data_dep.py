import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preset_gen_vae_amd import ops
B = 256
Cb, Cs, k, Hb, Wb = 8, 16, 4, 129, 174
g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
sc, sh = torch.ones(Cb, device='cuda'), torch.zeros(Cb, device='cuda')
gw = torch.empty_like(w)
flush = torch.empty(128 << 20, device='cuda')
def t(fn, n=20, do_flush=False):
    fn(); torch.cuda.synchronize()
    tot = 0
    for _ in range(n):
        if do_flush: flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
for name, big, small in [
    ('randn / randn', torch.randn(B, Cb, Hb, Wb, device='cuda'), torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')),
    ('randn / 1e-7*randn', torch.randn(B, Cb, Hb, Wb, device='cuda'), 1e-7 * torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')),
    ('randn / 1e-30*randn', torch.randn(B, Cb, Hb, Wb, device='cuda'), 1e-30 * torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')),
    ('zeros / zeros', torch.zeros(B, Cb, Hb, Wb, device='cuda'), torch.zeros(B, Cs, g.Hs, g.Ws, device='cuda')),
    ('leaky-like / sparse grads', torch.nn.functional.leaky_relu(torch.randn(B, Cb, Hb, Wb, device='cuda'), 0.1),
     torch.randn(B, Cs, g.Hs, g.Ws, device='cuda') * (torch.rand(B, Cs, g.Hs, g.Ws, device='cuda') > 0.5)),
]:
    f = lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh)
    print(f"{name:28s} warm {t(f):7.1f} us   cold {t(f, do_flush=True):7.1f} us")
# sustained: 300 back-to-back launches
big, small = torch.randn(B, Cb, Hb, Wb, device='cuda'), torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
f = lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh)
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    f(); torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        for _ in range(50): f()
torch.cuda.synchronize()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"graph of 50 launches x6: {e0.elapsed_time(e1)/300*1e3:.1f} us per launch (incl. memset)")
