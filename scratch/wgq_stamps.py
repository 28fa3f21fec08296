"""Phase stamps of conv_wgrad_split.hip (scratch/build_bigq_stamps.sh): workgroup 0, thread 0: kernel start, prologue done,
per unit (matrix start, matrix done, barrier passed), loop done, partial gradient written."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'libpgv_hip_stamps.so')
from preset_gen_vae_amd import ops
lib = _lib.load()
B = int(os.environ.get('B', 256))
for Cb, Cs, Hb, Wb in ((8, 16, 129, 174), (16, 32, 65, 88), (32, 64, 33, 45)):
    g = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    gw = torch.empty(Cs, Cb, 4, 4, device='cuda')
    ops.set_fp32_products('bf16x6')
    st = torch.zeros(64, device='cuda', dtype=torch.int64)
    for _ in range(3): ops.conv_wgrad(g, big, small, gw)
    lib.pgv_dbg_set_wgq_stamps(ctypes.c_void_p(st.data_ptr()))
    ops.conv_wgrad(g, big, small, gw)
    torch.cuda.synchronize()
    lib.pgv_dbg_set_wgq_stamps(None)
    s = st.cpu().tolist()
    t0 = s[0]
    units = [(s[2 + 3 * j] - t0, s[3 + 3 * j] - s[2 + 3 * j], s[4 + 3 * j] - s[3 + 3 * j]) for j in range(18) if s[2 + 3 * j]]
    print(f'{Hb}x{Wb} {Cb}<->{Cs}: prologue {s[1] - t0}, units (start, matrix, barrier wait): {units[:3]} ... {units[-2:]}, '
          f'loop done {s[60] - t0}, written {s[61] - t0}')
    if len(units) > 2:
        per = (units[-1][0] - units[1][0]) / (len(units) - 2)
        print(f'   per unit {per:.0f} ticks, {len(units)} units')
    ops.set_fp32_products('native')
