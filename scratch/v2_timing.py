"""In-kernel phase stamps of the conv_v2 kernels (debug library built by scratch/build_dbg.sh).
usage: v2_timing.py <down|up|wgrad> <enc2|enc3|enc4> [fwd|grad]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'libpgv_hip_dbg' + os.environ.get('PGV_DBG_SFX', '') + '.so')
from preset_gen_vae_amd import ops
lib = _lib.load()
lib.pgv_dbg_set_tlog_v2.argtypes = [ctypes.c_void_p]
LAYERS = {'enc2': (8, 16, 4, 129, 174), 'enc3': (16, 32, 4, 65, 88), 'enc4': (32, 64, 4, 33, 45)}
NAMES = ['prologue', 'pre-barrier', 'barrier wait', 'pre-loop', 'k-steps', 'epilogue', '-']
if len(sys.argv) > 4:
    lib.pgv_set_kernel_policy(int(sys.argv[4]))


def main():
    kind, layer = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else 'fwd'
    B = 256
    Cb, Cs, k, Hb, Wb = LAYERS[layer]
    g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda')
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
    sc_b, sh_b = torch.ones(Cb, device='cuda'), torch.zeros(Cb, device='cuda')
    sc_s, sh_s = torch.ones(Cs, device='cuda'), torch.zeros(Cs, device='cuda')
    bias_s, bias_b = torch.zeros(Cs, device='cuda'), torch.zeros(Cb, device='cuda')
    st_s = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
    st_b = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
    out_s, out_b, gw = torch.empty_like(small), torch.empty_like(big), torch.empty_like(w)
    if kind == 'down':
        fn = (lambda: ops.conv_down(g, big, w, bias_s, 1, 0.1, in_scale=sc_b, in_shift=sh_b, stats=st_s, out=out_s)) \
            if mode == 'fwd' else (lambda: ops.conv_down(g, big, w, None, 0, 0.0, out=out_s))
    elif kind == 'up':
        fn = (lambda: ops.conv_up(g, small, w, bias_b, 1, 0.1, in_scale=sc_s, in_shift=sh_s, stats=st_b, out=out_b)) \
            if mode == 'fwd' else (lambda: ops.conv_up(g, small, w, None, 0, 0.0, out=out_b))
    else:
        fn = lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc_b, big_shift=sh_b)
    tlog = torch.zeros(1 << 12, 8, 8, dtype=torch.int64, device='cuda')
    flush = torch.ones(64 << 20, device='cuda')
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.pgv_dbg_set_tlog_v2(ctypes.c_void_p(tlog.data_ptr()))
    flush.sum()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    lib.pgv_dbg_set_tlog_v2(None)
    raw = tlog.cpu().numpy()
    ok = raw[:, 0, 7] > 0
    d = raw[ok].astype(np.float64)
    n = d[:, :, 7]
    print(f"== {kind} {layer} {mode}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {ok.sum()} workgroups, items/wg {n.mean():.1f}; "
          f"cycles per workgroup (mean over workgroups; waves 0..3):")
    for i, nm in enumerate(NAMES[:6]):
        print(f"   {nm:14s} " + ' '.join(f"{d[:, wv, i].mean():9.0f}" for wv in range(4)) +
              f"   per item {d[:, 0, i].sum() / n[:, 0].sum():8.0f}")
    print(f"   sum            " + ' '.join(f"{d[:, wv, :6].sum(1).mean():9.0f}" for wv in range(4)))
    if d[:, 4, :6].sum() > 0:
        print("   loader waves 4..7 (cycles per workgroup):")
        for i, nm in ((0, 'commit'), (1, 'issue'), (2, 'barrier wait'), (3, 'sleep'), (5, 'load wait')):
            print(f"   {nm:14s} " + ' '.join(f"{d[:, wv, i].mean():9.0f}" for wv in range(4, 8)))


main()
