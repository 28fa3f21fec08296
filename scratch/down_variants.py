import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preset_gen_vae_amd import ops
B = 256
def t(fn, n=20):
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
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda'); out_s = torch.empty_like(small); out_b = torch.empty_like(big)
    bias = torch.zeros(Cs, device='cuda'); bias_b = torch.zeros(Cb, device='cuda')
    st = torch.zeros(2 * Cs, device='cuda', dtype=torch.float64); stb = torch.zeros(2 * Cb, device='cuda', dtype=torch.float64)
    sc, sh = torch.ones(Cb, device='cuda'), torch.zeros(Cb, device='cuda')
    scs, shs = torch.ones(Cs, device='cuda'), torch.zeros(Cs, device='cuda')
    print(nm, 'down: plain %.1f | +bias+act %.1f | +affine %.1f | +stats %.1f | all %.1f' % (
        t(lambda: ops.conv_down(g, big, w, None, 0, 0.0, out=out_s)),
        t(lambda: ops.conv_down(g, big, w, bias, 1, 0.1, out=out_s)),
        t(lambda: ops.conv_down(g, big, w, None, 0, 0.0, in_scale=sc, in_shift=sh, out=out_s)),
        t(lambda: ops.conv_down(g, big, w, None, 0, 0.0, stats=st, out=out_s, prezeroed=True)),
        t(lambda: ops.conv_down(g, big, w, bias, 1, 0.1, in_scale=sc, in_shift=sh, stats=st, out=out_s, prezeroed=True))))
    print(nm, 'up:   plain %.1f | +bias+act %.1f | +affine %.1f | +stats %.1f | all %.1f' % (
        t(lambda: ops.conv_up(g, small, w, None, 0, 0.0, out=out_b)),
        t(lambda: ops.conv_up(g, small, w, bias_b, 1, 0.1, out=out_b)),
        t(lambda: ops.conv_up(g, small, w, None, 0, 0.0, in_scale=scs, in_shift=shs, out=out_b)),
        t(lambda: ops.conv_up(g, small, w, None, 0, 0.0, stats=stb, out=out_b, prezeroed=True)),
        t(lambda: ops.conv_up(g, small, w, bias_b, 1, 0.1, in_scale=scs, in_shift=shs, stats=stb, out=out_b, prezeroed=True))))
