import sys, os, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import param_shapes, synth_input, rel_l2
from oracle import vae_oracle as vo
from preset_gen_vae_amd import config, ops
from preset_gen_vae_amd.model import build, layer
arch = sys.argv[1] if len(sys.argv) > 1 else 'speccnn4l1_bn'
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture = arch; mc.input_tensor_size = (2,1,257,347); tc.latent_flow_input_regularization='none'
enc, dec, ae = build.build_ae_model(mc, tc)
sd64 = vo.closed_form_state_dict(param_shapes(arch, 64, False), seed=1234, dtype=torch.float64)
ae.load_state_dict({k: (v if v.dtype == torch.long else v.float()) for k, v in sd64.items()})
ae = ae.cuda().eval()
x = synth_input(2)
taps = {}
zml, _, _, _, xo = vo.vae_forward(sd64, x, arch, 64, False, taps=taps)
# fp32 CPU oracle for noise floor
sd32 = {k: (v if v.dtype == torch.long else v.float()) for k, v in sd64.items()}
taps32 = {}
zml32, _, _, _, xo32 = vo.vae_forward(sd32, x.float(), arch, 64, False, taps=taps32)
print('cpu fp32 vs fp64 zml', rel_l2(zml32, zml), 'xout', rel_l2(xo32, xo))
with torch.no_grad():
    h = x.float().cuda()
    blocks = enc._all_blocks()
    names = ['enc%d' % (i + 1) for i in range(len(blocks))]
    for blk, n in zip(blocks, names):
        h = layer.run_stack(h, [blk], False)
        print(n, 'standalone chain', rel_l2(h, taps[n]), 'cpu32', rel_l2(taps32[n], taps[n]))
    hf = layer.run_stack(x.float().cuda(), blocks, False)
    print('fused stack', rel_l2(hf, taps[names[-1]]))
    z = ae.encoder(x.float().cuda())
    print('zml', rel_l2(z, zml), (z.double().cpu() - zml).abs().max().item())
    feat = taps[names[-1]].reshape(2, -1)
    lin = ops.linear_fwd(feat.float().cuda().contiguous(), ae.encoder.mlp[1].weight, ae.encoder.mlp[1].bias)
    print('fc on oracle feats', rel_l2(lin.view(2, 2, 64), zml))
    d = (z.double().cpu() - zml).abs().view(-1)
    print('worst idx', d.topk(5))
