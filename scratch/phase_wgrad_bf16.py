"""Per-phase time of the band weight-gradient kernels in bf16 operand mode (debug library with -DPGV_PHASE_TIMING)."""
import os, sys
os.environ.setdefault('PGV_DBG_LIB', 'libpgv_hip_phase.so')
sys.argv = [sys.argv[0]]
import importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, 'scratch', 'phase_timing.py')).read()
head = src[:src.index("for which, (Cb, Cs, k, Hb, Wb) in {'L2'")]
exec(compile(head, 'phase_timing_head', 'exec'))
ops.set_compute_dtype(os.environ.get('DTYPE', 'bf16'))
lib.pgv_set_kernel_policy(3)
for which, (Cb, Cs, k, Hb, Wb) in {'L2': (8, 16, 4, 129, 174), 'L3': (16, 32, 4, 65, 88), 'L4': (32, 64, 4, 33, 45)}.items():
    g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda'); gw = torch.empty_like(w)
    sc = torch.ones(Cb, device='cuda'); sh = torch.zeros(Cb, device='cuda')
    run(f'wgrad {which} bf16 band', lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh), 512)
