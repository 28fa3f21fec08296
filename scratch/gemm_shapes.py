import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preset_gen_vae_amd import ops
B, F, Z = 256, 25024, 64
x = torch.randn(B, F, device='cuda'); w1 = torch.randn(2 * Z, F, device='cuda') * 0.01; b1 = torch.zeros(2 * Z, device='cuda')
gy1 = torch.randn(B, 2 * Z, device='cuda')
z = torch.randn(B, Z, device='cuda'); w2 = torch.randn(F, Z, device='cuda') * 0.1; b2 = torch.zeros(F, device='cuda')
gy2 = torch.randn(B, F, device='cuda')
gw1 = torch.empty_like(w1); gw2 = torch.empty_like(w2)
def t(fn, n=20):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = [
 ('enc fwd  [256x25024]x[128x25024]^T', lambda: ops.linear_fwd(x, w1, b1), lambda: torch.addmm(b1, x, w1.t())),
 ('enc dgrad [256x128]x[128x25024]', lambda: ops.linear_dgrad(gy1, w1), lambda: gy1 @ w1),
 ('enc wgrad [128x256]x[256x25024]', lambda: ops.linear_wgrad(gy1, x, gw1), lambda: gy1.t() @ x),
 ('dec fwd  [256x64]x[25024x64]^T', lambda: ops.linear_fwd(z, w2, b2), lambda: torch.addmm(b2, z, w2.t())),
 ('dec dgrad [256x25024]x[25024x64]', lambda: ops.linear_dgrad(gy2, w2), lambda: gy2 @ w2),
 ('dec wgrad [25024x256]x[256x64]', lambda: ops.linear_wgrad(gy2, z, gw2), lambda: gy2.t() @ z),
]
for name, ours, ref in cases:
    try:
        a = t(ours)
    except Exception as e:
        a = float('nan'); print(e)
    print(f"{name:40s} ours {a:7.1f} us   torch {t(ref):7.1f} us")
