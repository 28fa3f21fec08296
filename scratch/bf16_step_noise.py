"""End-to-end bf16 train step against the bf16-mode oracle (B = 2, 8-layer z = 512) under several kernel routings: how the
scalar losses and the tensors scatter around the oracle when only the summation order of some layers changes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_vae as T
from helpers import synth_input, rel_l2
from oracle import vae_oracle as vo
from preset_gen_vae_amd import ops, _lib
from preset_gen_vae_amd.train_step import VAETrainStep
lib = _lib.load()
arch, dim_z, B = 'speccnn8l1_bn', 512, 2
x = synth_input(B)
eps = torch.sin(torch.arange(B * dim_z, dtype=torch.float64) * 0.37 + 0.2).reshape(B, dim_z) * 1.1
kw = dict(beta=0.2, lr=2e-4, weight_decay=1e-4)
ora = ora64 = None
for knob in (0, 32, 16, 48, 8, 56):
    ae = T._build(arch, dim_z, B, False, fc_dropout=0.0)
    sd64 = T._load_closed_form(ae, arch, dim_z, False, 4321)
    ae = ae.cuda().train()
    sd32 = {k: (v if v.dtype == torch.long else v.float()) for k, v in sd64.items()}
    if ora is None:
        with vo.operand_precision('bf16'):
            ora = vo.train_step(sd32, x.float(), arch, dim_z, eps.float(), None, None, **kw)
            ora64 = vo.train_step(sd64, x, arch, dim_z, eps, None, None, **kw)
        print('self noise: z', rel_l2(ora64['z_mu_logvar'], ora['z_mu_logvar']), 'x_out', rel_l2(ora64['x_out'], ora['x_out']),
              {k: abs(ora64[k].item() - ora[k].item()) / abs(ora[k].item()) for k in ('recons', 'latent', 'total')})
    lib.pgv_dbg_set_deep_bf16_variant(knob)
    ops.set_compute_dtype('bf16')
    try:
        step = VAETrainStep(ae, lr=2e-4, weight_decay=1e-4, beta=0.2, normalize_losses=True)
        out = step.step(T._cuda32(x), inject={'eps': T._cuda32(eps)})
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype('fp32')
    print(f'knob {knob:3d}: z', f"{rel_l2(out['z_mu_logvar'], ora['z_mu_logvar']):.2e}", 'x_out', f"{rel_l2(out['x_out'], ora['x_out']):.2e}",
          {k: f"{(out[k].item() - ora[k].item()) / abs(ora[k].item()):+.2e}" for k in ('recons', 'latent', 'total')}, flush=True)
lib.pgv_dbg_set_deep_bf16_variant(0)
