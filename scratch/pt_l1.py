import os, sys, runpy
os.environ['PGV_DBG_LIB'] = 'libpgv_hip_pt.so'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, 'scratch', 'phase_timing.py')).read()
head = src[:src.index("for which, (Cb, Cs, k, Hb, Wb) in {'L2'")]
exec(compile(head, 'phase_timing_head', 'exec'))
g = ops.ConvGeom(1, 8, 5, 2, 2, 257, 347)
big = torch.randn(B, 1, 257, 347, device='cuda'); small = torch.randn(B, 8, g.Hs, g.Ws, device='cuda')
gw = torch.empty(8, 1, 5, 5, device='cuda')
sc = torch.ones(8, device='cuda'); sh = torch.zeros(8, device='cuda')
run('wgrad L1 (enc1)', lambda: ops.conv_wgrad(g, big, small, gw), 1024)
run('wgrad L1 (dec8, affine on small)', lambda: ops.conv_wgrad(g, big, small, gw, small_scale=sc, small_shift=sh), 1024)
