"""Weight gradient of the three stride-2 k=4 layers in bf16 operand mode: conv_wgrad_bf16.hip (variant 0) against the band
kernels (variant 1), batch 256, with exactness check against float64 on the rounded operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib, ops
lib = _lib.load()
B = int(os.environ.get('B', 256))
ops.set_compute_dtype('bf16')
def bf(t): return t.float().bfloat16().double()
for which, (Cb, Cs, k, Hb, Wb) in {'L2': (8, 16, 4, 129, 174), 'L3': (16, 32, 4, 65, 88), 'L4': (32, 64, 4, 33, 45)}.items():
    g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
    torch.manual_seed(1)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda'); gw = torch.empty_like(w)
    sc = torch.rand(Cb, device='cuda') + 0.5; sh = torch.randn(Cb, device='cuda') * 0.1
    ssc = torch.rand(Cs, device='cuda') + 0.5; ssh = torch.randn(Cs, device='cuda') * 0.1
    nb = min(B, 32)
    for form in ('big', 'small', 'none'):
        kw = dict(big_scale=sc, big_shift=sh) if form == 'big' else dict(small_scale=ssc, small_shift=ssh) if form == 'small' else {}
        bb = torch.addcmul(sh.view(1, -1, 1, 1), big[:nb], sc.view(1, -1, 1, 1)) if form == 'big' else big[:nb]
        ss = torch.addcmul(ssh.view(1, -1, 1, 1), small[:nb], ssc.view(1, -1, 1, 1)) if form == 'small' else small[:nb]
        wv = w.double().clone().requires_grad_(True)
        F.conv2d(bf(bb), wv, None, stride=2, padding=2).backward(bf(ss))
        for v in (0, 1):
            lib.pgv_dbg_set_wgrad_bf16_variant(v)
            gs = torch.empty_like(w)
            ops.conv_wgrad(g, big[:nb].contiguous(), small[:nb].contiguous(), gs, **kw)
            err = ((gs.double() - wv.grad).norm() / wv.grad.norm()).item()
            for _ in range(5): ops.conv_wgrad(g, big, small, gw, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): ops.conv_wgrad(g, big, small, gw, **kw)
            e1.record(); torch.cuda.synchronize()
            print(f'{which} {form:5s} variant {v}: {e0.elapsed_time(e1) / 50 * 1000:7.1f} us   rel err (B={nb}) {err:.2e}', flush=True)
lib.pgv_dbg_set_wgrad_bf16_variant(0)
