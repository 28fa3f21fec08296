import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
from helpers import param_shapes, synth_input, rel_l2, synth_vec
from oracle import vae_oracle as vo
from preset_gen_vae_amd import config, ops
from preset_gen_vae_amd.model import build
arch, dz = 'speccnn8l1_bn', 64
sd = vo.closed_form_state_dict(param_shapes(arch, dz, False), seed=1234, dtype=torch.float64)
x = synth_input(2); eps = synth_vec((2, dz), 1.2345, 0.4) * 1.3
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if vo.is_parameter_key(k)}
full = dict(sd); full.update(params)
taps = {}
zml, z, _, _, xo = vo.vae_forward(full, x, arch, dz, True, eps, None, None, None, taps)
total = F.mse_loss(xo, x) + 0.2 * vo.gaussian_dkl(zml[:, 0], zml[:, 1])
# oracle intermediate grads for decoder blocks
names = ['dec7', 'dec6', 'dec5', 'dec4', 'dec3', 'dec2', 'dec1']
ref = {}
for n in names:
    ga, go = torch.autograd.grad(total, [taps[n + '_act'], taps[n]], retain_graph=True)
    ref[n] = (go.detach(), (ga * torch.where(taps[n + '_act'] > 0, 1.0, 0.1)).detach())
# product with recording
rec = []
orig = ops.act_bn_bwd
def patched(g_o, a, scale, mean, rstd, red, act, slope, g_y, gbias):
    gin = g_o.clone()
    orig(g_o, a, scale, mean, rstd, red, act, slope, g_y, gbias)
    rec.append((tuple(a.shape), gin, g_y.clone(), None if mean is None else (mean.clone(), rstd.clone(), scale.clone())))
ops.act_bn_bwd = patched
import preset_gen_vae_amd.model.layer as L
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture = arch; mc.input_tensor_size = (2,1,257,347); tc.latent_flow_input_regularization='none'
_, _, ae = build.build_ae_model(mc, tc)
ae.load_state_dict({k:(v if v.dtype==torch.long else v.float()) for k,v in sd.items()})
ae = ae.cuda().train()
c = lambda t: t.to('cuda', torch.float32).contiguous()
B = 2
out = ae(c(x), None, eps=c(eps), enc_dropout_mask=torch.ones(B, 24576, device='cuda'), dec_dropout_mask=torch.ones(B, 24576, device='cuda'))
from preset_gen_vae_amd.model import loss as LM
tot = LM.MSELoss()(out[4], c(x)) + ae.latent_loss(out[0]) * 0.2
tot.backward()
torch.cuda.synchronize()
print('loss', tot.item(), total.item())
# decoder blocks are recorded in order: dec8 (no bn), dec7, dec6, ...
for (shape, gin, gy, st), n in zip(rec[1:8], names):
    go_ref, gy_ref = ref[n]
    a = taps[n + '_act']
    mean = a.mean(dim=(0,2,3)); rstd = 1/torch.sqrt(a.var(dim=(0,2,3), unbiased=False) + 1e-5)
    print(n, shape, 'g_o err', rel_l2(gin, go_ref), 'g_y err', rel_l2(gy, gy_ref), 'mean err', rel_l2(st[0], mean), 'rstd err', rel_l2(st[1], rstd))
print('---- dec8 stage')
g_pre, g_xo = torch.autograd.grad(total, [taps['dec8_pre'], xo], retain_graph=True)
shape, gin, gy, st = rec[0]
print('dec8 g_in (dL/dx_out) err', rel_l2(gin, g_xo), 'g_y8 err', rel_l2(gy, g_pre), 'nonzero frac ref', (g_pre != 0).double().mean().item(), 'got', (gy != 0).double().mean().item())
d = (gy.double().cpu() - g_pre).abs()
print('max abs err', d.max().item(), 'ref max', g_pre.abs().max().item(), 'count err>1e-9:', (d > 1e-9).sum().item())
w8 = sd['decoder.single_ch_cnn.dec_nn.6.weight']
geom = ops.ConvGeom(1, 8, 5, 2, 2, 257, 347)
go7 = ops.conv_down(geom, c(g_pre), c(w8), None, 0, 0.0)
print('direct down on oracle g_y8 -> g_o7 err', rel_l2(go7, ref['dec7'][0]))
go7b = ops.conv_down(geom, gy, c(w8), None, 0, 0.0)
print('direct down on product g_y8 -> g_o7 err', rel_l2(go7b, ref['dec7'][0]))
xo_p = out[4].detach()
print('x_out err', rel_l2(xo_p, xo), 'clamped frac ref', (xo.abs() >= 1).double().mean().item(), 'got', (xo_p.abs() >= 1).double().mean().item())
flip = ((xo_p.abs().cpu() >= 1) != (xo.abs() >= 1))
print('gate flips:', flip.sum().item(), 'of', flip.numel())
