"""Per-wave phase stamps of deep_down (variant library built with -DPGV_DEEP_STAMPS=<block>): cycles between the marks."""
import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from preset_gen_vae_amd import _lib, ops
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'libpgv_deep_st.so')
lib = _lib.load()
B = 256
for nm, (Cb, Cs, Hb, Wb) in {'G7': (256, 512, 5, 7), 'G5': (64, 128, 17, 23)}.items():
    g = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, 4, 4, device='cuda') * 0.05
    sc, sh = 1 + 0.1 * torch.randn(Cb, device='cuda'), 0.1 * torch.randn(Cb, device='cuda')
    st_s = torch.empty(2 * Cs, device='cuda', dtype=torch.float64); bs = torch.zeros(Cs, device='cuda')
    for _ in range(3):
        ops.conv_down(g, big, w, bs, 1, 0.1, in_scale=sc, in_shift=sh, stats=st_s)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 64 * 8))()
    lib.pgv_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.pgv_debug_read_stamps(buf, 8 * 64 * 8) == 0
    import numpy as np
    a = np.array(buf, dtype=np.int64).reshape(8, 64, 8)
    nsl = Cb // 4
    for wv in (0, 3):
        t = a[wv, :nsl, :6]
        d = np.diff(t, axis=1)            # 0->1 first half MFMA issue, 1->2 wait loads, 2->3 commit+issue, 3->4 second half, 4->5 barrier
        nxt = t[1:, 0] - t[:-1, 5]        # barrier exit -> next slab start
        per = t[1:, 0] - t[:-1, 0]
        print(nm, 'wave', wv, 'slab period', per[2:-2].mean().round(), 'phases', d[2:-2].mean(axis=0).round(), 'gap', nxt[2:-2].mean().round())
