import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for (B, C, H, W) in [(256, 512, 3, 4), (256, 256, 5, 7), (256, 128, 9, 12)]:
    g = torch.randn(B, C, H, W, device='cuda'); a = torch.randn(B, C, H, W, device='cuda')
    mean = a.mean(dim=(0, 2, 3)).contiguous(); rstd = (1.0 / torch.sqrt(a.var(dim=(0, 2, 3), unbiased=False) + 1e-5)).contiguous()
    scale = rstd.clone()
    red = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    gy = torch.empty_like(g); gb = torch.zeros(C, device='cuda'); gg = torch.empty(C, device='cuda'); gbt = torch.empty(C, device='cuda')
    def two():
        red.zero_()
        ops.bn_bwd_reduce(g, a, mean, rstd, red, prezeroed=True)
        ops.act_bn_bwd(g, a, scale, mean, rstd, red, 1, 0.1, gy, gb, ggamma=gg, gbeta=gbt, prezeroed=True)
    def one():
        ops.bn_act_bwd_fused(g, a, scale, mean, rstd, 1, 0.1, gy, gb, ggamma=gg, gbeta=gbt, prezeroed=True)
    two(); r2 = gy.clone(); one(); r1 = gy.clone()
    print((B, C, H, W), f'two passes {timeit(two):6.1f} us   fused {timeit(one):6.1f} us   max diff {(r1 - r2).abs().max().item():.2e}')
