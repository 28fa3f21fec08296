"""B = 256 tiling property: per-parameter gradient deviation from the golden (fraction of the parameter's largest gradient)
in both product modes - which parameters sit near the 5e-3 line, and is that the mode or the golden's conditioning?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import ops
from preset_gen_vae_amd.train_step import VAETrainStep
import test_gpu_vae as T
from helpers import load_golden, synth_input, unpack_mask

for name in sys.argv[1:] or ['vae8l_b2_outbn.npz']:
    g = load_golden(name)
    arch, dim_z, B0, output_bn = str(g['meta/arch']), int(g['meta/dim_z']), int(g['meta/B']), bool(g['meta/output_bn'])
    reps = int(os.environ.get('REPS', 256 // B0))
    res = {}
    for mode in ('native', 'bf16x6'):
        ops.set_fp32_products(mode)
        ae = T._build(arch, dim_z, B0 * reps, output_bn)
        T._load_closed_form(ae, arch, dim_z, output_bn, int(g['meta/seed']))
        ae = ae.cuda().train()
        x = T._cuda32(synth_input(B0)).repeat(reps, 1, 1, 1)
        inject = {'eps': T._cuda32(torch.tensor(g['in/eps'])).repeat(reps, 1),
                  'enc_dropout_mask': T._cuda32(unpack_mask(g, 'enc')).repeat(reps, 1),
                  'dec_dropout_mask': T._cuda32(unpack_mask(g, 'dec')).repeat(reps, 1)}
        step = VAETrainStep(ae, lr=float(g['meta/lr']), weight_decay=float(g['meta/weight_decay']), beta=float(g['meta/beta']))
        step.step(x, inject=inject)
        for k, p in ae.named_parameters():
            cs = g['grad/' + k + '/checksum']
            if cs[2] < 1e-9:
                continue
            idx = torch.tensor(g['grad/' + k + '/sample_idx'])
            sample = torch.tensor(g['grad/' + k + '/sample'])
            got = p.grad.double().cpu().reshape(-1)[idx]
            res.setdefault(k, {})[mode] = ((got - sample).abs().max().item() / cs[2], cs[2])
            if k.endswith(os.environ.get('SHOW', 'dec2tconv.bias')):
                d = (got - sample).abs()
                i = int(d.argmax())
                print(mode, k, 'worst sampled element', int(idx[i]), 'got', float(got[i]), 'golden', float(sample[i]), 'n sampled', len(idx),
                      'elements off by > 2e-3 of max:', int((d > 2e-3 * cs[2]).sum()))
    ops.set_fp32_products('native')
    print(name)
    for k, v in sorted(res.items(), key=lambda kv: -max(kv[1]['native'][0], kv[1]['bf16x6'][0]))[:12]:
        print(f"  {k:55s} native {v['native'][0]:.2e}  bf16x6 {v['bf16x6'][0]:.2e}  (largest |g| {v['native'][1]:.2e})")
