"""Deep k4 s2 p2 layers in fp32: products as six bf16 instructions (ops.set_fp32_products('bf16x6'), conv_deep_split.hip)
against the native fp32 kernels of conv_deep.hip.  WHAT = down | up | all"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
B = int(os.environ.get('B', 256))
WHAT = os.environ.get('WHAT', 'all')
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
torch.manual_seed(1)
for (Cb, Cs, H, W) in [(64, 128, 17, 23), (128, 256, 9, 12), (256, 512, 5, 7), (512, 2048, 3, 4)]:
    K1 = H == 3
    if K1 and WHAT == 'wgrad': continue
    g = ops.ConvGeom(Cb, Cs, 1, 1, 0, H, W) if K1 else ops.ConvGeom(Cb, Cs, 4, 2, 2, H, W)
    big = torch.randn(B, Cb, H, W, device='cuda'); small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    w = torch.randn(Cs, Cb, *((1, 1) if K1 else (4, 4)), device='cuda') * 0.05
    bias_s = torch.randn(Cs, device='cuda') * 0.1; bias_b = torch.randn(Cb, device='cuda') * 0.1
    bsc = torch.rand(Cb, device='cuda') + 0.5; bsh = torch.randn(Cb, device='cuda') * 0.1
    ssc = torch.rand(Cs, device='cuda') + 0.5; ssh = torch.randn(Cs, device='cuda') * 0.1
    nb = 8
    refd = F.conv2d(big[:nb].double(), w.double(), None) if K1 else F.conv2d(big[:nb].double(), w.double(), None, stride=2, padding=2)
    oph, opw = H - ((g.Hs - 1) * 2 - 4 + 4), W - ((g.Ws - 1) * 2 - 4 + 4)
    refu = F.conv_transpose2d(small[:nb].double(), w.double(), None) if K1 else F.conv_transpose2d(small[:nb].double(), w.double(), None, stride=2, padding=2, output_padding=(oph, opw))
    for mode in ('native', 'bf16x6'):
        ops.set_fp32_products(mode)
        sh = ops.conv_weight_shadow(g, w)
        kw = dict(w_shadow=sh) if sh is not None else {}
        line = f'{H}x{W} {Cb}->{Cs} {mode:7s}:'
        if WHAT in ('down', 'all'):
            o = ops.conv_down(g, big[:nb].contiguous(), w, None, ops.PGV_ACT_NONE, 0.0, **kw)
            err = ((o.double() - refd).norm() / refd.norm()).item()
            st = torch.zeros(ops.CLS_COPIES * 2 * Cs, device='cuda', dtype=torch.float64)
            t = timeit(lambda: ops.conv_down(g, big, w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=bsc, in_shift=bsh, stats=st, prezeroed=True, stats_copies=True, **kw))
            t2 = timeit(lambda: ops.conv_down(g, big, w, None, ops.PGV_ACT_NONE, 0.0, **kw))
            line += f'  down fwd {t:6.1f} us plain {t2:6.1f} us err {err:.2e}'
        if WHAT in ('up', 'all'):
            o = ops.conv_up(g, small[:nb].contiguous(), w, None, ops.PGV_ACT_NONE, 0.0, **kw)
            err = ((o.double() - refu).norm() / refu.norm()).item()
            st = torch.zeros(ops.CLS_COPIES * 2 * Cb, device='cuda', dtype=torch.float64)
            t = timeit(lambda: ops.conv_up(g, small, w, bias_b, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=ssc, in_shift=ssh, stats=st, prezeroed=True, stats_copies=True, **kw))
            t2 = timeit(lambda: ops.conv_up(g, small, w, None, ops.PGV_ACT_NONE, 0.0, **kw))
            line += f'  up fwd {t:6.1f} us plain {t2:6.1f} us err {err:.2e}'
        if WHAT in ('wgrad', 'all') and not K1:
            gw = torch.empty_like(w)
            wv = w.double().clone().requires_grad_(True)
            F.conv2d(big[:24].double(), wv, None, stride=2, padding=2).backward(small[:24].double())
            ops.conv_wgrad(g, big[:24].contiguous(), small[:24].contiguous(), gw)
            err = ((gw.double() - wv.grad).norm() / wv.grad.norm()).item()
            t = timeit(lambda: ops.conv_wgrad(g, big, small, gw, big_scale=bsc, big_shift=bsh))
            line += f'  wgrad {t:6.1f} us err {err:.2e}'
        if sh is not None:
            ts = timeit(lambda: ops.conv_weight_shadow(g, w))
            line += f'  shadow {ts:5.1f} us'
        print(line)
    ops.set_fp32_products('native')
