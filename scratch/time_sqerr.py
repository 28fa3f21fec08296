"""sqerr_act_bwd (output layer criterion backward) with and without its single-address reduction outputs"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from preset_gen_vae_amd import ops
B = 256
dev = torch.device('cuda', 0)
xo, xt, gy = (torch.randn(B, 1, 257, 347, device=dev) for _ in range(3))
gl = torch.ones((), device=dev)
gb = torch.zeros(1, device=dev); la = torch.zeros((), device=dev); cls = torch.zeros(ops.CLS_COPIES * 4, device=dev)
for name, fn in (("cls + gbias + loss", lambda: ops.sqerr_act_bwd(xo, xt, gl, 1e-8, 2, 0.0, gy, gb, prezeroed=True, loss_acc=la, cls=cls)),
                 ("cls + gbias       ", lambda: ops.sqerr_act_bwd(xo, xt, gl, 1e-8, 2, 0.0, gy, gb, prezeroed=True, cls=cls)),
                 ("cls only          ", lambda: ops.sqerr_act_bwd(xo, xt, gl, 1e-8, 2, 0.0, gy, None, prezeroed=True, cls=cls)),
                 ("plain kernel, gbias", lambda: ops.sqerr_act_bwd(xo, xt, gl, 1e-8, 2, 0.0, gy, gb, prezeroed=True))):
    try:
        print(f"{name} {bench.time_kernel(fn, iters=5) * 1e3:7.1f} us", flush=True)
    except Exception as e:
        print(name, 'n/a', str(e)[:80])
