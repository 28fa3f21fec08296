"""Tuning sweep of the fragment-streaming nn.Linear kernels (gemm_frag.hip) over pgv_dbg_set_gemm_variant."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from preset_gen_vae_amd import ops, _lib
B = 256
lib = _lib.load()
dz = int(os.environ.get('DZ', 64))
x = torch.randn(B, 25024, device='cuda'); We = torch.randn(2 * dz, 25024, device='cuda') * 0.01; be = torch.zeros(2 * dz, device='cuda')
gye = torch.randn(B, 2 * dz, device='cuda'); gWe = torch.empty_like(We)
z = torch.randn(B, dz, device='cuda'); Wd = torch.randn(25024, dz, device='cuda') * 0.01; bd = torch.zeros(25024, device='cuda')
gyd = torch.randn(B, 25024, device='cuda'); gWd = torch.empty_like(Wd)
ye = torch.zeros(B, 2 * dz, device='cuda'); gz = torch.zeros(B, dz, device='cuda')
fns = (lambda: ops.linear_fwd(x, We, be, out=ye), lambda: ops.linear_dgrad(gye, We), lambda: ops.linear_wgrad(gye, x, gWe),
       lambda: ops.linear_fwd(z, Wd, bd), lambda: ops.linear_dgrad(gyd, Wd, out=gz), lambda: ops.linear_wgrad(gyd, z, gWd))
def run(v, tag):
    lib.pgv_dbg_set_gemm_variant(v)
    ts = [bench.time_kernel(f, iters=5) * 1e3 for f in fns]
    print(f"{tag:40s} v={v:4d}: enc {ts[0]:5.1f} {ts[1]:5.1f} {ts[2]:5.1f} | dec {ts[3]:5.1f} {ts[4]:5.1f} {ts[5]:5.1f} | sum {sum(ts):6.1f}", flush=True)
for rep in range(4):
    run(1024, "OLD (gemm.hip LDS tiles)")
    run(2, "new stages 3 cap 1024")
    run(2 | 16, "new stages 3 cap 1024 jobs 2048")
    run(0 | 16, "new stages 4 cap 1024 jobs 2048")
lib.pgv_dbg_set_gemm_variant(0)
