"""Weight gradient of the large-plane layers: error against float64 of the native and six-instruction forms per affine variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import torch.nn.functional as F
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
import test_gpu_kernels as T

dev = T.dev
for case in T.BIG_SPLIT_CASES:
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = T._conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    if os.environ.get('RANDN', '1') == '1':
        gen = torch.Generator().manual_seed(B)
        big = (torch.randn(big.shape, generator=gen) + 0.5).to(big.dtype)
        small = torch.randn(small.shape, generator=gen).to(small.dtype)
    big_n = T._affine_fma(big, sc_b, sh_b).double()
    for name, kw_n, bigd, smalld in (('big', {'big_scale': dev(sc_b), 'big_shift': dev(sh_b)}, big_n, small.double()),
                               ('small', {'small_scale': dev(sc_s), 'small_shift': dev(sh_s)}, big.double(),
                                T._affine_fma(small, sc_s, sh_s).double()),
                               ('none', {}, big.double(), small.double())):
        wv = w.double().clone().requires_grad_(True)
        F.conv2d(bigd, wv, None, stride=s, padding=p).backward(smalld)
        out = []
        for mode in ('native', 'bf16x6'):
            ops.set_fp32_products(mode)
            gw = torch.empty((Cs, Cb, k, k), device='cuda')
            ops.conv_wgrad(geom, dev(big), dev(small), gw, **kw_n)
            out.append(T.rel_l2(gw, wv.grad))
            d = (gw.double().cpu() - wv.grad.cpu()).abs()
            out.append(float(d.max() / wv.grad.abs().max()))
        ops.set_fp32_products('native')
        print(case, name, ' native %.2e (max %.2e)  bf16x6 %.2e (max %.2e)' % tuple(out), flush=True)
