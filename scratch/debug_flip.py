import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import param_shapes, synth_input, rel_l2, synth_vec
from oracle import vae_oracle as vo
from preset_gen_vae_amd import config, ops
from preset_gen_vae_amd.model import build
from preset_gen_vae_amd.model import loss as LM
arch, dz, B = 'speccnn8l1_bn', 64, 2
sd = vo.closed_form_state_dict(param_shapes(arch, dz, False), seed=1234, dtype=torch.float64)
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture = arch; mc.input_tensor_size = (B,1,257,347); tc.latent_flow_input_regularization='none'
_, _, ae = build.build_ae_model(mc, tc)
ae.load_state_dict({k:(v if v.dtype==torch.long else v.float()) for k,v in sd.items()})
ae = ae.cuda().train()
c = lambda t: t.to('cuda', torch.float32).contiguous()
x = c(synth_input(2)); eps = c(synth_vec((2, dz), 1.2345, 0.4))
ones_e = torch.ones(B, 24576, device='cuda'); ones_d = torch.ones(B, 24576, device='cuda')
rec = []
orig = ops.act_bn_bwd
def patched(g_o, a, scale, mean, rstd, red, act, slope, g_y, gbias):
    gin = g_o.clone(); orig(g_o, a, scale, mean, rstd, red, act, slope, g_y, gbias)
    rec.append((a.clone(), gin, g_y.clone()))
ops.act_bn_bwd = patched
runs = []
for trial in range(6):
    rec.clear()
    for p in ae.parameters(): p.grad = None
    out = ae(x, None, eps=eps, enc_dropout_mask=ones_e, dec_dropout_mask=ones_d)
    tot = LM.MSELoss()(out[4], x) + ae.latent_loss(out[0]) * 0.2
    tot.backward(); torch.cuda.synchronize()
    runs.append([tuple(t.clone() for t in r) for r in rec])
for trial in range(1, 6):
    msgs = []
    for li, (r0, r1) in enumerate(zip(runs[0], runs[trial])):
        a0, gi0, gy0 = r0; a1, gi1, gy1 = r1
        flips = ((a0 > 0) != (a1 > 0)).sum().item()
        msgs.append(f'L{li}:{tuple(a0.shape)[1:]} a {rel_l2(a1,a0):.1e} flips {flips} gin {rel_l2(gi1,gi0):.1e} gy {rel_l2(gy1,gy0):.1e}')
    print(trial, ' | '.join(msgs[:9]))
