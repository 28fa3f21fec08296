"""fp32 transposed convolution onto 65x88: products as six bf16 instructions (ops.set_fp32_products('bf16x6')) against the
native fp32 kernels, plain forward form and fused input gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
B = 256
g = ops.ConvGeom(16, 32, 4, 2, 2, 65, 88)
torch.manual_seed(1)
big = torch.randn(B, 16, 65, 88, device='cuda'); small = torch.randn(B, 32, g.Hs, g.Ws, device='cuda')
w = torch.randn(32, 16, 4, 4, device='cuda') * 0.05; bias_b = torch.randn(16, device='cuda') * 0.1
ssc = torch.rand(32, device='cuda') + 0.5; ssh = torch.randn(32, device='cuda') * 0.1
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
a = big * 1.3 + 0.1
coef = torch.cat([1.0 + 0.3 * torch.rand(16, device='cuda'), 0.05 * torch.randn(16, device='cuda'), 0.02 * torch.randn(16, device='cuda')])
nb = 8
ref = F.conv_transpose2d(small[:nb].double(), w.double(), None, stride=2, padding=2, output_padding=(1, 0))
for mode in ('native', 'bf16x6'):
    ops.set_fp32_products(mode)
    sh = ops.conv_weight_shadow(g, w)
    kw = dict(w_shadow=sh) if sh is not None else {}
    o = ops.conv_up(g, small[:nb].contiguous(), w, None, ops.PGV_ACT_NONE, 0.0, **kw)
    err = ((o.double() - ref).norm() / ref.norm()).item()
    st = torch.zeros(ops.CLS_COPIES * 32, device='cuda', dtype=torch.float64)
    t = timeit(lambda: ops.conv_up(g, small, w, bias_b, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=ssc, in_shift=ssh, stats=st, prezeroed=True, stats_copies=True, **kw))
    gbc = torch.zeros(ops.CLS_COPIES * 16, device='cuda')
    fz = (a, coef, gbc, ops.PGV_ACT_LEAKY_RELU, 0.1, None, ops.CLS_COPIES)
    tf = timeit(lambda: ops.conv_up(g, small, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fz, **kw))
    print(f'{mode:7s}: forward (affine, bias, act, stats) {t:6.1f} us   fused input gradient {tf:6.1f} us   rel L2 error vs float64 {err:.2e}')
ops.set_fp32_products('native')
