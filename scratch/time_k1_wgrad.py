"""1x1 weight gradient (512 <-> 2048 on 3x4, B = 256): native fp32 kernel against the six-instruction form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
B = 256
flush = torch.empty(128 * 1024 * 1024, device='cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    tot = 0.0
    for _ in range(n):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1000
g = ops.ConvGeom(512, 2048, 1, 1, 0, 3, 4)
big = torch.randn(B, 512, 3, 4, device='cuda'); small = torch.randn(B, 2048, 3, 4, device='cuda')
sc, sh = torch.rand(512, device='cuda') + 0.5, torch.randn(512, device='cuda') * 0.1
gw = torch.zeros(2048, 512, 1, 1, device='cuda')
for mode in ('native', 'bf16x6'):
    ops.set_fp32_products(mode)
    t = timeit(lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh))
    print(f'{mode:7s}: 1x1 weight gradient {t:6.1f} us')
ops.set_fp32_products('native')

import ctypes
from preset_gen_vae_amd import _lib
lib = _lib.load()
ops.set_fp32_products('bf16x6')
st = torch.zeros(64, device='cuda', dtype=torch.int64)
lib.pgv_dbg_set_deep_bf16_stamps(ctypes.c_void_p(st.data_ptr()))
ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh)
torch.cuda.synchronize()
lib.pgv_dbg_set_deep_bf16_stamps(None)
v = st.cpu().tolist()
print('stamps (ticks from kernel start): prologue', v[1] - v[0], '| unit 2: matrix', v[3] - v[2], 'barrier', v[4] - v[3], 'commit+issue', v[5] - v[4],
      'barrier', v[6] - v[5], '| loop end', v[7] - v[0], 'stores done', v[8] - v[0])
ops.set_fp32_products('native')
