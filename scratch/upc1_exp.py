import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
if os.environ.get('PGV_DBG_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'scratch', os.environ['PGV_DBG_LIB'])
import bench
from preset_gen_vae_amd import ops
B = 256
g = ops.ConvGeom(1, 8, 5, 2, 2, 257, 347)
small = torch.randn(B, 8, g.Hs, g.Ws, device='cuda'); w = torch.randn(8, 1, 5, 5, device='cuda') * 0.1
bb = torch.zeros(1, device='cuda'); out = torch.empty(B, 1, 257, 347, device='cuda')
sc, sh = torch.ones(8, device='cuda'), torch.zeros(8, device='cuda')
t = bench.time_kernel(lambda: ops.conv_up(g, small, w, bb, 2, 0.0, in_scale=sc, in_shift=sh, out=out), iters=5)
print(os.environ.get('PGV_DBG_LIB', 'normal'), f"up_c1: {t*1e3:.1f} us")
big = torch.randn(B, 1, 257, 347, device='cuda'); outs = torch.empty_like(small); bs = torch.zeros(8, device='cuda')
t = bench.time_kernel(lambda: ops.conv_down(g, big, w, bs, 1, 0.1, out=outs), iters=5)
print(f"   down_c1: {t*1e3:.1f} us")
t = bench.time_kernel(lambda: out.copy_(big), iters=5)
print(f"   copy 91MB->91MB: {t*1e3:.1f} us")
t = bench.time_kernel(lambda: outs.copy_(small), iters=5)
print(f"   copy 184MB->184MB: {t*1e3:.1f} us")
