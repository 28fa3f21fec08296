import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import param_shapes, synth_input, rel_l2, synth_vec
from oracle import vae_oracle as vo
from preset_gen_vae_amd import config
from preset_gen_vae_amd.model import build
from preset_gen_vae_amd.model import loss as LM
arch, dz, B = sys.argv[1], 64, int(sys.argv[2])
sd = vo.closed_form_state_dict(param_shapes(arch, dz, False), seed=1234, dtype=torch.float64)
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture = arch; mc.input_tensor_size = (B,1,257,347); tc.latent_flow_input_regularization='none'
_, _, ae = build.build_ae_model(mc, tc)
ae.load_state_dict({k:(v if v.dtype==torch.long else v.float()) for k,v in sd.items()})
ae = ae.cuda().train()
c = lambda t: t.to('cuda', torch.float32).contiguous()
x = c(synth_input(2)).repeat(B//2,1,1,1); eps = c(synth_vec((2, dz), 1.2345, 0.4)).repeat(B//2,1)
F = ae.encoder.mlp[1].in_features
ones_e = torch.ones(B, F, device='cuda'); ones_d = torch.ones(B, ae.decoder.mlp[0].out_features, device='cuda')
base = None
for trial in range(8):
    for p in ae.parameters(): p.grad = None
    out = ae(x, None, eps=eps, enc_dropout_mask=ones_e, dec_dropout_mask=ones_d)
    tot = LM.MSELoss()(out[4], x) + ae.latent_loss(out[0]) * 0.2
    tot.backward(); torch.cuda.synchronize()
    g = {k: p.grad.clone() for k, p in ae.named_parameters()}
    if base is None: base = g; xo0 = out[4].clone(); continue
    worst = sorted(((rel_l2(g[k], base[k]), k) for k in g if base[k].abs().max() > 1e-9), reverse=True)[:3]
    print(trial, 'x_out diff', rel_l2(out[4], xo0), 'worst grad diffs', [(round(a, 9), b[-40:]) for a, b in worst])
