import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
g = torch.randn(256, 2048, 3, 4, device='cuda'); a = torch.randn_like(g); gy = torch.empty_like(g); gb = torch.zeros(2048, device='cuda')
def timeit(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
print('act_bn_bwd without BatchNorm [256, 2048, 3, 4]:', timeit(lambda: ops.act_bn_bwd(g, a, None, None, None, None, 1, 0.1, gy, gb, prezeroed=True)), 'us')
