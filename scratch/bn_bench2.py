import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from preset_gen_vae_amd import ops
B = 256
for C, HW in ((8, 129 * 174), (16, 65 * 88), (32, 33 * 45)):
    a = torch.randn(B, C, HW, device='cuda'); g = torch.randn(B, C, HW, device='cuda'); gy = torch.empty_like(g)
    mean, rstd, scale = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.ones(C, device='cuda')
    red = torch.zeros(2 * C, device='cuda', dtype=torch.float64); gb = torch.zeros(C, device='cuda')
    t1 = bench.time_kernel(lambda: ops.bn_bwd_reduce(g, a, mean, rstd, red, prezeroed=True), iters=5) * 1e3
    t2 = bench.time_kernel(lambda: ops.act_bn_bwd(g, a, scale, mean, rstd, red, 1, 0.1, gy, gb, prezeroed=True), iters=5) * 1e3
    print(f"C={C:5d} HW={HW:6d}: reduce {t1:6.1f} us ({2*a.numel()*4/t1/1e6:.2f} TB/s), act_bn_bwd {t2:6.1f} us ({3*a.numel()*4/t2/1e6:.2f} TB/s)")
