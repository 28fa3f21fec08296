"""8-layer z = 512 train step in bf16 operand mode at odd batch sizes: runs, finite, and close to the fp32-mode step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_vae as T
from helpers import synth_input
from preset_gen_vae_amd import ops
from preset_gen_vae_amd.train_step import VAETrainStep
arch, dim_z = 'speccnn8l1_bn', 512
for B in (2, 3, 19, 33, 257):
    res = {}
    for mode in ('fp32', 'bf16'):
        ae = T._build(arch, dim_z, B, False, fc_dropout=0.0)
        T._load_closed_form(ae, arch, dim_z, False, 4321)
        ae = ae.cuda().train()
        x = synth_input(B)
        eps = torch.sin(torch.arange(B * dim_z, dtype=torch.float64) * 0.37 + 0.2).reshape(B, dim_z) * 1.1
        ops.set_compute_dtype(mode)
        try:
            step = VAETrainStep(ae, lr=2e-4, weight_decay=1e-4, beta=0.2, normalize_losses=True)
            out = step.step(T._cuda32(x), inject={'eps': T._cuda32(eps)})
            torch.cuda.synchronize()
        finally:
            ops.set_compute_dtype('fp32')
        gn = sum(float(p.grad.double().pow(2).sum()) for p in ae.parameters() if p.grad is not None) ** 0.5
        res[mode] = (out['recons'].item(), out['latent'].item(), gn)
    r32, r16 = res['fp32'], res['bf16']
    ok = all(abs(a - b) <= 2e-2 * abs(a) + 1e-6 for a, b in zip(r32, r16)) and all(v == v for v in r16)
    print(f'B={B:4d}  fp32 {r32[0]:.6f} {r32[1]:.6f} |g| {r32[2]:.5f}   bf16 {r16[0]:.6f} {r16[1]:.6f} |g| {r16[2]:.5f}   {"ok" if ok else "MISMATCH"}', flush=True)
