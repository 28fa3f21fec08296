import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preset_gen_vae_amd import ops
B = 256
Cb, Cs, k, Hb, Wb = 8, 16, 4, 129, 174
g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
sc, sh = torch.ones(Cb, device='cuda'), torch.zeros(Cb, device='cuda')
bias = torch.zeros(Cs, device='cuda')
gw = torch.empty_like(w)
st = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
N = 8
bigs = [torch.randn(B, Cb, Hb, Wb, device='cuda') for _ in range(N)]
smalls = [torch.randn(B, Cs, g.Hs, g.Ws, device='cuda') for _ in range(N)]
def bench(fn, label):
    for i in range(N): fn(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr):
            for r in range(5):
                for i in range(N): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:50s} {e0.elapsed_time(e1)/(4*5*N)*1e3:7.1f} us per launch")
bench(lambda i: ops.conv_wgrad(g, bigs[0], smalls[0], gw, big_scale=sc, big_shift=sh), 'wgrad same buffers')
bench(lambda i: ops.conv_wgrad(g, bigs[i], smalls[i], gw, big_scale=sc, big_shift=sh), 'wgrad rotating 8 buffer pairs (2.2 GB)')
bench(lambda i: ops.conv_down(g, bigs[0], w, bias, 1, 0.1, in_scale=sc, in_shift=sh, stats=st, out=smalls[0]), 'down same buffers')
bench(lambda i: ops.conv_down(g, bigs[i], w, bias, 1, 0.1, in_scale=sc, in_shift=sh, stats=st, out=smalls[i]), 'down rotating')
bench(lambda i: ops.conv_up(g, smalls[0], w, None, 0, 0.0, out=bigs[0]), 'up same buffers')
bench(lambda i: ops.conv_up(g, smalls[i], w, None, 0, 0.0, out=bigs[i]), 'up rotating')
