import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
from helpers import rel_l2
from preset_gen_vae_amd import ops
torch.manual_seed(0)
for (Cb, Cs, k, s, p, Hb, Wb) in [(64,128,4,2,2,17,23),(128,256,4,2,2,9,12),(256,512,4,2,2,5,7),(512,2048,1,1,0,3,4)]:
  for B in (2, 3):
    g = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
    sc_b = torch.rand(Cb, device='cuda') + 0.5; sh_b = torch.randn(Cb, device='cuda') * 0.1
    sc_s = torch.rand(Cs, device='cuda') + 0.5; sh_s = torch.randn(Cs, device='cuda') * 0.1
    outs = []
    for rep in range(6):
        st1 = torch.empty(2 * Cs, device='cuda', dtype=torch.float64); st2 = torch.empty(2 * Cb, device='cuda', dtype=torch.float64)
        d = ops.conv_down(g, big, w, None, 1, 0.1, in_scale=sc_b, in_shift=sh_b, stats=st1)
        u = ops.conv_up(g, small, w, None, 1, 0.1, in_scale=sc_s, in_shift=sh_s, stats=st2)
        gw = torch.empty_like(w); ops.conv_wgrad(g, big, small, gw, small_scale=sc_s, small_shift=sh_s)
        gw2 = torch.empty_like(w); ops.conv_wgrad(g, big, small, gw2, big_scale=sc_b, big_shift=sh_b)
        torch.cuda.synchronize()
        outs.append((d.clone(), u.clone(), gw.clone(), gw2.clone(), st1.clone(), st2.clone()))
    names = ['down', 'up', 'wgrad_s', 'wgrad_b', 'stats_d', 'stats_u']
    msg = []
    for i, n in enumerate(names):
        worst = max(rel_l2(o[i], outs[0][i]) for o in outs[1:])
        msg.append(f'{n} {worst:.1e}')
    print((Cb, Cs, k), 'B', B, ' '.join(msg))
