import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import param_shapes, synth_input, rel_l2, synth_vec, load_golden, unpack_mask
from oracle import vae_oracle as vo
from preset_gen_vae_amd import config
from preset_gen_vae_amd.model import build
from preset_gen_vae_amd.train_step import VAETrainStep
name = sys.argv[1] if len(sys.argv) > 1 else 'vae8l_b2.npz'
g = load_golden(name)
arch, dz, ob = str(g['meta/arch']), 64, bool(g['meta/output_bn'])
sd64 = vo.closed_form_state_dict(param_shapes(arch, dz, ob), seed=1234, dtype=torch.float64)
x2 = synth_input(2); eps = torch.tensor(g['in/eps']); em, dm = unpack_mask(g, 'enc'), unpack_mask(g, 'dec')
ora = vo.train_step(sd64, x2, arch, dz, eps, em, dm)
sd32 = {k:(v if v.dtype==torch.long else v.float()) for k,v in sd64.items()}
ora32 = vo.train_step(sd32, x2.float(), arch, dz, eps.float(), em.float(), dm.float())
c = lambda t: t.to('cuda', torch.float32).contiguous()
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture = arch; mc.input_tensor_size = (2,1,257,347); tc.latent_flow_input_regularization='bn' if ob else 'none'
_, _, ae = build.build_ae_model(mc, tc)
ae.load_state_dict(sd32)
ae = ae.cuda().train()
step = VAETrainStep(ae)
out = step.step(c(x2), inject={'eps': c(eps), 'enc_dropout_mask': c(em), 'dec_dropout_mask': c(dm)})
torch.cuda.synchronize()
rows = []
for k, p in ae.named_parameters():
    gr = ora['grads'][k]
    if gr.abs().max() < 1e-9: continue
    rows.append((rel_l2(p.grad, gr), rel_l2(ora32['grads'][k], gr), k))
for r in rows: print('%.2e  cpu32 %.2e  %s' % r)
