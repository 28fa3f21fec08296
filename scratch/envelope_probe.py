"""What the six-instruction fp32 products do outside N(0,1) operands: wide dynamic range, denormal residual planes, inf/nan."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import preset_gen_vae_amd
from preset_gen_vae_amd import ops
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()
def run(case, big, small, w, tag):
    Cb, Cs, k, s, p, Hb, Wb, B = case
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    ref = F.conv2d(big.double(), w.double(), None, stride=s, padding=p)
    wv = w.double().clone().requires_grad_(True)
    F.conv2d(big.double(), wv, None, stride=s, padding=p).backward(small.double())
    res = {}
    for mode in ('native', 'bf16x6'):
        ops.set_fp32_products(mode)
        sh = ops.conv_weight_shadow(geom, w.cuda())
        d = ops.conv_down(geom, big.cuda(), w.cuda(), None, 0, 0.0, w_shadow=sh)
        gw = torch.empty_like(w, device='cuda')
        ops.conv_wgrad(geom, big.cuda(), small.cuda(), gw)
        res[mode] = (rel(d, ref), rel(gw, wv.grad), torch.isfinite(d).all().item())
    print(tag, case[:2], case[5:], {m: tuple('%.2e' % v if isinstance(v, float) else v for v in r) for m, r in res.items()}, flush=True)
gen = torch.Generator().manual_seed(5)
def wide(shape, lo, hi):
    e = torch.rand(shape, generator=gen) * (hi - lo) + lo
    return (torch.sign(torch.randn(shape, generator=gen)) * torch.exp2(e)).float()
for case in [(16, 32, 4, 2, 2, 65, 88, 3), (64, 128, 4, 2, 2, 17, 23, 3), (512, 2048, 1, 1, 0, 3, 4, 4)]:
    Cb, Cs, k, s, p, Hb, Wb, B = case
    Hs, Ws = (Hb + 2 * p - k) // s + 1, (Wb + 2 * p - k) // s + 1
    bs, ss, ws = (B, Cb, Hb, Wb), (B, Cs, Hs, Ws), (Cs, Cb, k, k)
    run(case, wide(bs, -40, 40), wide(ss, -20, 20), wide(ws, -40, 40), 'wide 2^+-40  ')
    run(case, wide(bs, -60, 60), wide(ss, -2, 2), wide(ws, -60, 60), 'wide 2^+-60  ')
    big = torch.randn(bs, generator=gen); m = torch.rand(bs, generator=gen) < 0.1
    big[m] = big[m] * 2.0 ** -120
    run(case, big, torch.randn(ss, generator=gen), torch.randn(ws, generator=gen) * 0.1, 'mixed tiny   ')
    run(case, torch.randn(bs, generator=gen) * 2.0 ** -118, torch.randn(ss, generator=gen), torch.randn(ws, generator=gen), 'all 2^-118   ')
    run(case, torch.randn(bs, generator=gen) * 2.0 ** -100, torch.randn(ss, generator=gen), torch.randn(ws, generator=gen), 'all 2^-100   ')
    for bad in (float('inf'), float('nan')):
        big = torch.randn(bs, generator=gen); big[1, 3, 5, 6 % Wb] = bad
        geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
        for mode in ('native', 'bf16x6'):
            ops.set_fp32_products(mode)
            w = torch.randn(ws, generator=gen).cuda()
            d = ops.conv_down(geom, big.cuda(), w, None, 0, 0.0, w_shadow=ops.conv_weight_shadow(geom, w))
            nf = (~torch.isfinite(d)).sum().item()
            ref = F.conv2d(big.double(), w.double().cpu(), None, stride=s, padding=p)
            print('   ', bad, mode, 'non-finite outputs', nf, 'reference', (~torch.isfinite(ref)).sum().item(), 'nan', torch.isnan(d).sum().item(), flush=True)
