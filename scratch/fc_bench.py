"""Standalone timing of the six nn.Linear products of the step (graph-differential, cold cache)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from preset_gen_vae_amd import ops, _lib
B = 256
import os
if os.environ.get('FC_TILES'):
    _lib.load().pgv_dbg_set_gemm_tiles(int(os.environ['FC_TILES']))
if os.environ.get('FC_BF16'):
    ops.set_compute_dtype('bf16')
for dz, FEAT in ((64, 25024), (512, 12288)):
    x = torch.randn(B, FEAT, device='cuda'); We = torch.randn(2 * dz, FEAT, device='cuda') * 0.01; be = torch.zeros(2 * dz, device='cuda')
    gye = torch.randn(B, 2 * dz, device='cuda'); gWe = torch.empty_like(We)
    z = torch.randn(B, dz, device='cuda'); Wd = torch.randn(FEAT, dz, device='cuda') * 0.01; bd = torch.zeros(FEAT, device='cuda')
    gyd = torch.randn(B, FEAT, device='cuda'); gWd = torch.empty_like(Wd)
    for pol in (0, 2, 1):
        if pol == 2:   # second line: bf16 - every covered shape on gemm_frag.hip; fp32 - split-K jobs without the in-workgroup reduction
            pol = 0; _lib.load().pgv_dbg_set_gemm_variant(4096 if os.environ.get('FC_BF16') else 8192)
        if os.environ.get('FC_BF16') and pol == 1:
            pol = 0; _lib.load().pgv_dbg_set_gemm_variant(1024)   # third line: gemm.hip's bf16 MFMA tiles
        _lib.load().pgv_set_kernel_policy(pol)
        ts = [bench.time_kernel(f, iters=5) * 1e3 for f in (
            lambda: ops.linear_fwd(x, We, be), lambda: ops.linear_dgrad(gye, We), lambda: ops.linear_wgrad(gye, x, gWe),
            lambda: ops.linear_fwd(z, Wd, bd), lambda: ops.linear_dgrad(gyd, Wd), lambda: ops.linear_wgrad(gyd, z, gWd))]
        print(f"dz={dz} policy {pol}: enc fwd/dgrad/wgrad {ts[0]:.1f} {ts[1]:.1f} {ts[2]:.1f} | dec {ts[3]:.1f} {ts[4]:.1f} {ts[5]:.1f} | sum {sum(ts):.1f} us")
    _lib.load().pgv_set_kernel_policy(0); _lib.load().pgv_dbg_set_gemm_variant(0)
