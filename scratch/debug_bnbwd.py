import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
from helpers import param_shapes, synth_input, rel_l2, synth_vec
from oracle import vae_oracle as vo
from preset_gen_vae_amd import ops
arch, dz = 'speccnn8l1_bn', 64
sd = vo.closed_form_state_dict(param_shapes(arch, dz, False), seed=1234, dtype=torch.float64)
x = synth_input(2); eps = synth_vec((2, dz), 1.2345, 0.4) * 1.3
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if vo.is_parameter_key(k)}
full = dict(sd); full.update(params)
taps = {}
zml, z, _, _, xo = vo.vae_forward(full, x, arch, dz, True, eps, None, None, None, taps)
total = F.mse_loss(xo, x) + 0.2 * vo.gaussian_dkl(zml[:, 0], zml[:, 1])
for name in ['dec7', 'dec6', 'dec5', 'dec4']:
    a, o = taps[name + '_act'], taps[name]
    g_a, g_o = torch.autograd.grad(total, [a, o], retain_graph=True)
    C = a.shape[1]
    mean = a.mean(dim=(0, 2, 3)); var = a.var(dim=(0, 2, 3), unbiased=False); rstd = 1 / torch.sqrt(var + 1e-5)
    gamma = [v for k, v in sd.items() if k.endswith(name + 'bn.weight')][0]
    scale = gamma * rstd
    dev = lambda t: t.detach().to('cuda', torch.float32).contiguous()
    red = torch.empty(2 * C, device='cuda', dtype=torch.float64)
    d_go, d_a, d_mean, d_rstd, d_scale = dev(g_o), dev(a), dev(mean), dev(rstd), dev(scale)
    ops.bn_bwd_reduce(d_go, d_a, d_mean, d_rstd, red)
    g_y = torch.empty_like(d_go); gb = torch.empty(C, device='cuda')
    ops.act_bn_bwd(d_go, d_a, d_scale, d_mean, d_rstd, red, 1, 0.1, g_y, gb)
    ref = g_a * torch.where(a > 0, 1.0, 0.1)
    # same formula in torch float32 on CPU
    a32, go32 = a.detach().float(), g_o.float()
    ah = (a32 - mean.float().view(1,-1,1,1)) * rstd.float().view(1,-1,1,1)
    c1 = go32.mean(dim=(0,2,3)).view(1,-1,1,1); c2 = (go32*ah).mean(dim=(0,2,3)).view(1,-1,1,1)
    ga32 = scale.float().view(1,-1,1,1) * (go32 - c1 - ah * c2)
    amp = ((g_o * scale.view(1,-1,1,1)).norm() / g_a.norm()).item()
    print(name, 'hip', rel_l2(g_y, ref), 'torch-f32 formula', rel_l2(ga32 * torch.where(a32 > 0, 1.0, 0.1), ref), 'amplification |scale*g_o|/|g_a| =', amp,
          'c1/|g|', (c1.abs().mean() / go32.abs().mean()).item())
print('---- wgrad / dgrad kernels on the real tensors')
prev = {'dec7': 'dec6', 'dec6': 'dec5', 'dec5': 'dec4', 'dec4': 'dec3'}
geoms = {'dec7': (8, 16, 129, 174), 'dec6': (16, 32, 65, 88), 'dec5': (32, 64, 33, 45), 'dec4': (64, 128, 17, 23)}
for name in ['dec7', 'dec6', 'dec5', 'dec4']:
    a, o = taps[name + '_act'], taps[name]
    g_a, = torch.autograd.grad(total, [a], retain_graph=True)
    g_y = (g_a * torch.where(a > 0, 1.0, 0.1)).detach()
    ap = taps[prev[name] + '_act'].detach(); op_ = taps[prev[name]]
    g_op, = torch.autograd.grad(total, [op_], retain_graph=True)
    Cb, Cs, Hb, Wb = geoms[name]
    geom = ops.ConvGeom(Cb, Cs, 4, 2, 2, Hb, Wb)
    C = ap.shape[1]
    mean = ap.mean(dim=(0, 2, 3)); var = ap.var(dim=(0, 2, 3), unbiased=False); rstd = 1 / torch.sqrt(var + 1e-5)
    gam = [v for k, v in sd.items() if k.endswith(prev[name] + 'bn.weight')][0]
    bet = [v for k, v in sd.items() if k.endswith(prev[name] + 'bn.bias')][0]
    sc = gam * rstd; sh = bet - mean * sc
    wkey = [k for k in params if k.endswith(name + 'tconv.weight')][0]
    gw_ref, = torch.autograd.grad(total, [params[wkey]], retain_graph=True)
    dev = lambda t: t.detach().to('cuda', torch.float32).contiguous()
    gw = torch.empty(tuple(gw_ref.shape), device='cuda')
    ops.conv_wgrad(geom, dev(g_y), dev(ap), gw, small_scale=dev(sc), small_shift=dev(sh))
    e1 = rel_l2(gw, gw_ref)
    ops.conv_wgrad(geom, dev(g_y), dev(op_), gw)
    e2 = rel_l2(gw, gw_ref)
    gx = ops.conv_down(geom, dev(g_y), dev(sd[wkey]), None, 0, 0.0)
    e3 = rel_l2(gx, g_op)
    # torch fp32 on CPU for the same ops
    gw32 = torch.nn.grad.conv2d_weight(g_y.float(), tuple(gw_ref.shape), op_.detach().float(), stride=2, padding=2)
    print(name, 'wgrad lazy', e1, 'wgrad materialised', e2, 'dgrad', e3, 'torch f32 wgrad', rel_l2(gw32, gw_ref), '|sh|max', sh.abs().max().item(), 'mean^2/var max', (mean*mean/var).max().item())
