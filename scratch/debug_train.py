import sys, os, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import param_shapes, synth_input, rel_l2, synth_vec
from oracle import vae_oracle as vo
from preset_gen_vae_amd import config, ops
from preset_gen_vae_amd.model import build, layer
arch = sys.argv[1] if len(sys.argv) > 1 else 'speccnn4l1_bn'
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture = arch; mc.input_tensor_size = (2,1,257,347); tc.latent_flow_input_regularization='none'
enc, dec, ae = build.build_ae_model(mc, tc)
sd64 = vo.closed_form_state_dict(param_shapes(arch, 64, False), seed=1234, dtype=torch.float64)
ae.load_state_dict({k: (v if v.dtype == torch.long else v.float()) for k, v in sd64.items()})
ae = ae.cuda().train()
x = synth_input(2)
eps = synth_vec((2, 64), 1.2345, 0.4) * 1.3
taps = {}
zml, z, _, _, xo = vo.vae_forward(sd64, x, arch, 64, True, eps=eps, taps=taps)
sd32 = {k: (v if v.dtype == torch.long else v.float()) for k, v in sd64.items()}
taps32 = {}
zml32, _, _, _, xo32 = vo.vae_forward(sd32, x.float(), arch, 64, True, eps=eps.float(), taps=taps32)
print('cpu fp32 vs fp64 zml', rel_l2(zml32, zml), 'xout', rel_l2(xo32, xo))
with torch.no_grad():
    h = x.float().cuda()
    blocks = enc._all_blocks()
    names = ['enc%d' % (i + 1) for i in range(len(blocks))]
    for blk, n in zip(blocks, names):
        a = taps[n + '_act']
        m = a.mean(dim=(0, 2, 3)); v = a.var(dim=(0, 2, 3), unbiased=False)
        h = layer.run_stack(h, [blk], True)
        print(n, 'chain', rel_l2(h, taps[n]), 'cpu32', rel_l2(taps32[n], taps[n]), 'max mean^2/var', (m * m / v).max().item(), 'min var', v.min().item())
    hf = layer.run_stack(x.float().cuda(), blocks, True)
    print('fused stack', rel_l2(hf, taps[names[-1]]))
    ones_e = torch.ones(2, enc.mlp[1].in_features, device='cuda')
    zz = ae.encoder(x.float().cuda(), dropout_mask=ones_e)
    print('zml', rel_l2(zz, zml))
    dblocks = dec._all_blocks()
    ones_d = torch.ones(2, dec.mlp[0].out_features, device='cuda')
    xo_p = ae.decoder(z.float().cuda(), dropout_mask=ones_d)
    print('decoder on oracle z', rel_l2(xo_p, xo))
