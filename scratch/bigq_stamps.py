"""Phase stamps of conv_big_split.hip (scratch/build_bigq_stamps.sh -> scratch/libpgv_hip_stamps.so): clock64() of workgroup 0,
first wave of each team, per trip: 0 matrix segment start, 1 loads issued, 2 matrix loop done, 4 tile written, 5 barrier passed,
6 loads arrived, 7 committed, 8 moved out, 9 barrier passed."""
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
which = sys.argv[1] if len(sys.argv) > 1 else 'down'
fused = len(sys.argv) > 2 and sys.argv[2] == 'fused'
for Cb, Cs, Hb, Wb in ((8, 16, 129, 174), (16, 32, 65, 88), (32, 64, 33, 45)):
    g = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
    torch.manual_seed(1)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    w = torch.randn(Cs, Cb, 4, 4, device='cuda') * 0.05
    ops.set_fp32_products('bf16x6')
    sh = ops.conv_weight_shadow(g, w)
    st = torch.zeros(2 * 64 * 16, device='cuda', dtype=torch.int64)
    C8 = ops.CLS_COPIES
    def run():
        if which == 'down':
            if fused:
                Co = Cs
                a = small * 1.3 + 0.1
                coef = torch.cat([torch.ones(Co, device='cuda'), torch.zeros(2 * Co, device='cuda')])
                ops.conv_down(g, big, w, None, 0, 0.0, bwd_fuse=(a, coef, torch.zeros(C8 * Co, device='cuda'), 1, 0.1, torch.zeros(C8 * 4 * Co, device='cuda'), C8), w_shadow=sh)
            else:
                ops.conv_down(g, big, w, None, 1, 0.1, w_shadow=sh)
        else:
            if fused:
                Co = Cb
                a = big * 1.3 + 0.1
                coef = torch.cat([torch.ones(Co, device='cuda'), torch.zeros(2 * Co, device='cuda')])
                ops.conv_up(g, small, w, None, 0, 0.0, bwd_fuse=(a, coef, torch.zeros(C8 * Co, device='cuda'), 1, 0.1, None, C8), w_shadow=sh)
            else:
                ops.conv_up(g, small, w, None, 1, 0.1, w_shadow=sh)
    for _ in range(3): run()
    lib.pgv_dbg_set_bigq_stamps(ctypes.c_void_p(st.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.pgv_dbg_set_bigq_stamps(None)
    s = st.cpu().view(2, 64, 16)
    print(f'{which} {"fused" if fused else "plain"} {Hb}x{Wb} {Cb}<->{Cs}')
    for team in range(2):
        rows = []
        for j in range(2, 8):
            r = s[team, j + 2]
            if r[0] == 0: continue
            d = [int(r[k] - r[0]) if r[k] else -1 for k in (10, 11, 12, 2, 4, 5, 6, 7, 8, 9)]
            rows.append(d)
        names = ['item0', 'item1', 'item2', 'mfma', 'tile', 'bar', 'arrived', 'commit', 'out', 'bar']
        for d in rows[:4]:
            print(f'  team {team}: ' + '  '.join(f'{n} {v}' for n, v in zip(names, d)))
    ops.set_fp32_products('native')
