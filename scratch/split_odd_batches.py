"""8-layer train step with fp32 products as six bf16 instructions at odd batch sizes: runs, finite, and equal to the native
fp32 step: losses to 1e-6; the gradient norm to 1e-3 only - two fp32 evaluations with different summation orders put a few
pre-activations on different sides of a LeakyReLU kink, which moves gradients by 4e-4 .. 5e-3 (tests/test_gpu_vae.py pins the
regions for its strict comparison; this script does not).  Measured: losses equal to 1 - 3e-7, |g| to 3e-6 .. 4.8e-4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_vae as T
from helpers import synth_input
from preset_gen_vae_amd import ops
from preset_gen_vae_amd.train_step import VAETrainStep
arch, dim_z = 'speccnn8l1_bn', 64
for B in (1, 2, 3, 7, 19, 33, 257):
    res = {}
    for mode in ('native', 'bf16x6'):
        ae = T._build(arch, dim_z, B, False, fc_dropout=0.0)
        T._load_closed_form(ae, arch, dim_z, False, 4321)
        ae = ae.cuda().train()
        x = synth_input(B)
        eps = torch.sin(torch.arange(B * dim_z, dtype=torch.float64) * 0.37 + 0.2).reshape(B, dim_z) * 1.1
        ops.set_fp32_products(mode)
        try:
            step = VAETrainStep(ae, lr=2e-4, weight_decay=1e-4, beta=0.2, normalize_losses=True)
            out = step.step(T._cuda32(x), inject={'eps': T._cuda32(eps)})
            torch.cuda.synchronize()
        finally:
            ops.set_fp32_products('native')
        gn = sum(float(p.grad.double().pow(2).sum()) for p in ae.parameters() if p.grad is not None) ** 0.5
        res[mode] = (out['recons'].item(), out['latent'].item(), gn)
    a, b = res['native'], res['bf16x6']
    ok = all(abs(u - v) <= 1e-6 * abs(u) + 1e-7 for u, v in zip(a[:2], b[:2])) and abs(a[2] - b[2]) <= 1e-3 * a[2] and all(v == v for v in b)
    print(f'B={B:4d}  native {a[0]:.7f} {a[1]:.7f} |g| {a[2]:.6f}   bf16x6 {b[0]:.7f} {b[1]:.7f} |g| {b[2]:.6f}   {"ok" if ok else "MISMATCH"}', flush=True)
