"""bf16-native deep kernels (conv_deep_bf16.hip, w_shadow=) against the fp32-image kernels in bf16 mode: exactness vs float64
on the rounded operands and time at batch 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib, ops
lib = _lib.load()
B = int(os.environ.get('B', 256))
WHAT = os.environ.get('WHAT', 'down,up').split(',')
ops.set_compute_dtype('bf16')
if os.environ.get('KNOB'):
    lib.pgv_dbg_set_deep_bf16_variant(int(os.environ['KNOB']))
def bf(t): return t.float().bfloat16().double()
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
CASES = {'129x174': (8, 16, 129, 174), '65x88': (16, 32, 65, 88), '33x45': (32, 64, 33, 45), '17x23': (64, 128, 17, 23), '9x12': (128, 256, 9, 12), '5x7': (256, 512, 5, 7), 'k1': (512, 2048, 3, 4)}
for which, (Cb, Cs, Hb, Wb) in CASES.items():
    if os.environ.get('ONLY') and which not in os.environ['ONLY'].split(','): continue
    KK, ST, PD = (1, 1, 0) if which == 'k1' else (4, 2, 2)
    g = ops.ConvGeom(Cb, Cs, KK, ST, PD, Hb, Wb)
    torch.manual_seed(1)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, KK, KK, device='cuda') * 0.05
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    bias_s = torch.randn(Cs, device='cuda') * 0.1; bias_b = torch.randn(Cb, device='cuda') * 0.1
    sc = torch.rand(Cb, device='cuda') + 0.5; sh = torch.randn(Cb, device='cuda') * 0.1
    ssc = torch.rand(Cs, device='cuda') + 0.5; ssh = torch.randn(Cs, device='cuda') * 0.1
    shadow = ops.conv_weight_shadow(g, w)
    t_sh = timeit(lambda: ops.conv_weight_shadow(g, w))
    nb = min(B, 19)
    if 'down' in WHAT:
        xin = torch.addcmul(sh.view(1, -1, 1, 1), big[:nb], sc.view(1, -1, 1, 1))
        ref = F.leaky_relu(F.conv2d(bf(xin), bf(w), bias_s.double(), stride=ST, padding=PD), 0.1)
        for name, kw in (('old', {}), ('new', dict(w_shadow=shadow))):
            st = torch.zeros(2 * Cs, device='cuda', dtype=torch.float64)
            got = ops.conv_down(g, big[:nb].contiguous(), w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc, in_shift=sh, stats=st, **kw)
            err = ((got.double() - ref).norm() / ref.norm()).item()
            serr = ((st - torch.cat([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))])).norm() / ref.pow(2).sum().sqrt()).item()
            t = timeit(lambda: ops.conv_down(g, big, w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc, in_shift=sh, **kw))
            if name == 'new' and os.environ.get('ABL'):
                for v in (1, 2, 4, 6, 7):
                    lib.pgv_dbg_set_deep_bf16_variant(v)
                    tv = timeit(lambda: ops.conv_down(g, big, w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc, in_shift=sh, **kw))
                    print(f'   ablation {v}: {tv:7.1f} us')
                lib.pgv_dbg_set_deep_bf16_variant(0)
            if name == 'new' and os.environ.get('STAMPS'):
                import ctypes
                stb = torch.zeros(32, device='cuda', dtype=torch.int64)
                lib.pgv_dbg_set_deep_bf16_stamps.argtypes = [ctypes.c_void_p]
                lib.pgv_dbg_set_deep_bf16_stamps(ctypes.c_void_p(stb.data_ptr()))
                ops.conv_down(g, big, w, bias_s, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=sc, in_shift=sh, stats=st, **kw)
                torch.cuda.synchronize()
                lib.pgv_dbg_set_deep_bf16_stamps(None)
                v = stb.cpu().tolist()
                for o in (0, 16):
                    print('   stamps', [v[o + i] - v[o] for i in range(8)], ' wg0->wg77 start', v[16] - v[0])
            print(f'down {which} {name}: {t:7.1f} us  err {err:.2e} stats {serr:.2e}   (shadow {t_sh:.1f} us)', flush=True)
    if 'up' in WHAT:
        sin = torch.addcmul(ssh.view(1, -1, 1, 1), small[:nb], ssc.view(1, -1, 1, 1))
        oph, opw = Hb - ((g.Hs - 1) * ST - 2 * PD + KK), Wb - ((g.Ws - 1) * ST - 2 * PD + KK)
        ref = F.leaky_relu(F.conv_transpose2d(bf(sin), bf(w), bias_b.double(), stride=ST, padding=PD, output_padding=(oph, opw)), 0.1)
        for name, kw in (('old', {}), ('new', dict(w_shadow=shadow))):
            st = torch.zeros(2 * Cb, device='cuda', dtype=torch.float64)
            got = ops.conv_up(g, small[:nb].contiguous(), w, bias_b, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=ssc, in_shift=ssh, stats=st, **kw)
            err = ((got.double() - ref).norm() / ref.norm()).item()
            serr = ((st - torch.cat([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))])).norm() / ref.pow(2).sum().sqrt()).item()
            t = timeit(lambda: ops.conv_up(g, small, w, bias_b, ops.PGV_ACT_LEAKY_RELU, 0.1, in_scale=ssc, in_shift=ssh, **kw))
            print(f'up   {which} {name}: {t:7.1f} us  err {err:.2e} stats {serr:.2e}', flush=True)
    if 'wgrad' in WHAT:
        for form in ('big', 'small'):
            kw = dict(big_scale=sc, big_shift=sh) if form == 'big' else dict(small_scale=ssc, small_shift=ssh)
            bb = torch.addcmul(sh.view(1, -1, 1, 1), big[:nb], sc.view(1, -1, 1, 1)) if form == 'big' else big[:nb]
            ss = torch.addcmul(ssh.view(1, -1, 1, 1), small[:nb], ssc.view(1, -1, 1, 1)) if form == 'small' else small[:nb]
            wv = w.double().clone().requires_grad_(True)
            F.conv2d(bf(bb), wv, None, stride=ST, padding=PD).backward(bf(ss))
            for name, v in (('old', 8), ('new', 0)) + ((('alt64', 64),) if which == '9x12' else ()) + ((('alt128', 128),) if which == '17x23' else ()):
                lib.pgv_dbg_set_deep_bf16_variant(v)
                gs = torch.empty_like(w)
                ops.conv_wgrad(g, big[:nb].contiguous(), small[:nb].contiguous(), gs, **kw)
                err = ((gs.double() - wv.grad).norm() / wv.grad.norm()).item()
                gw = torch.empty_like(w)
                t = timeit(lambda: ops.conv_wgrad(g, big, small, gw, **kw))
                if name == 'new' and os.environ.get('STAMPS') and form == 'big':
                    import ctypes
                    stb = torch.zeros(32, device='cuda', dtype=torch.int64)
                    lib.pgv_dbg_set_deep_bf16_stamps.argtypes = [ctypes.c_void_p]
                    lib.pgv_dbg_set_deep_bf16_stamps(ctypes.c_void_p(stb.data_ptr()))
                    ops.conv_wgrad(g, big, small, gw, **kw)
                    torch.cuda.synchronize()
                    lib.pgv_dbg_set_deep_bf16_stamps(None)
                    vv = stb.cpu().tolist()
                    for o in (0, 16):
                        print('   stamps', [vv[o + i] - vv[o] for i in range(6)])
                print(f'wgrad {which} {form:5s} {name}: {t:7.1f} us  err {err:.2e}', flush=True)
            lib.pgv_dbg_set_deep_bf16_variant(0)
    if 'dfuse' in WHAT:   # input-gradient form with the fused BatchNorm + activation backward of the block below
        for direction in ('down', 'up'):
            C = Cs if direction == 'down' else Cb
            a = (small if direction == 'down' else big) * 1.3 + 0.1
            coef = torch.cat([1.0 + 0.3 * torch.rand(C, device='cuda'), 0.05 * torch.randn(C, device='cuda'), 0.02 * torch.randn(C, device='cuda')])
            for name, kw in (('old', {}), ('new', dict(w_shadow=shadow))):
                gbc = torch.zeros(ops.CLS_COPIES * C, device='cuda')
                cls = torch.zeros(ops.CLS_COPIES * 4 * C, device='cuda') if direction == 'down' else None
                fz = (a, coef, gbc, ops.PGV_ACT_LEAKY_RELU, 0.1, cls, ops.CLS_COPIES)
                fn = (lambda: ops.conv_down(g, big, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fz, **kw)) if direction == 'down' else \
                     (lambda: ops.conv_up(g, small, w, None, ops.PGV_ACT_NONE, 0.0, bwd_fuse=fz, **kw))
                o = fn()
                t = timeit(fn)
                print(f'dfuse {direction} {which} {name}: {t:7.1f} us  checksum {o.double().sum().item():.6e}', flush=True)
