"""Standalone timing of the mask-free dropout kernels (graph-differential, cold cache), with and without the affine."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from preset_gen_vae_amd import ops
from preset_gen_vae_amd.rng import DeviceRNG
rng = DeviceRNG(torch.device('cuda'), seed=1)
x4 = torch.randn(256, 64, 16, 24, device='cuda'); x2 = x4.view(256, -1)
sc = torch.rand(64, device='cuda') + 0.5; sh = torch.randn(64, device='cuda')
y, saved = ops.dropout_fwd(rng.state, 2, 0.3, x2)
g = torch.randn_like(x2)
m = torch.ones_like(x2)
stats = torch.zeros(128, device='cuda', dtype=torch.float64); ops.bn_stats(x4, stats)
vec = [torch.empty(64, device='cuda') for _ in range(4)]
src = ops.bn_src(stats, 256 * 384, sc, sh, 1e-5, 0.1, None, None, None, *vec)
x4b = torch.randn(256, 64, 16, 24, device='cuda')
def after_write():
    x4b.mul_(1.0001)            # the tensor was just written by the previous kernel, as in the step
    ops.dropout_fwd(rng.state, 2, 0.3, x4b, sc, sh)
def write_only():
    x4b.mul_(1.0001)
for name, fn in (("fwd bn-src", lambda: ops.dropout_fwd(rng.state, 2, 0.3, x4, in_bn=src)),
                 ("write + fwd affine", after_write), ("write only", write_only),
                 ("fwd plain", lambda: ops.dropout_fwd(rng.state, 2, 0.3, x2)),
                 ("fwd affine", lambda: ops.dropout_fwd(rng.state, 2, 0.3, x4, sc, sh)),
                 ("bwd", lambda: ops.dropout_bwd(saved, 2, 0.3, g)),
                 ("apply (stored mask)", lambda: ops.dropout_apply(rng.state, 2, 0.3, x2)),
                 ("mul", lambda: ops.mul(g, m)),
                 ("affine", lambda: ops.affine_nchw(x4, sc, sh))):
    print(f"{name:22s} {bench.time_kernel(fn, iters=5) * 1e3:6.1f} us")
