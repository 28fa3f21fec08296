"""Are all gradient buckets launched from the gradient-ready hooks (before wait()) in the eager N > 1 mode?"""
import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preset_gen_vae_amd import config, parallel
from preset_gen_vae_amd.model import build as mbuild
from preset_gen_vae_amd.train_step import VAETrainStep
B = 8
for arch in ('speccnn4l1_bn', 'speccnn8l1_bn'):
    mc, tc = copy.copy(config.model), copy.copy(config.train)
    mc.encoder_architecture, mc.dim_z, mc.input_tensor_size = arch, 64, (B, 1, 257, 347)
    tc.latent_flow_input_regularization = 'bn'
    _, _, ae = mbuild.build_ae_model(mc, tc)
    ae = ae.cuda().train()
    step = VAETrainStep(ae, use_graph=False, grad_sync=lambda flat: parallel.GradAllReduce(flat, n_buckets=4))
    sync = step.grad_sync
    seen = []
    orig_wait = sync.wait
    def wait():
        seen.append((list(sync._launched), list(sync._done), list(sync._need)))
        return orig_wait()
    sync.wait = wait
    x = torch.randn(B, 1, 257, 347, device='cuda').clamp_(-1, 1)
    step.step(x); step.step(x)
    print(arch, 'before wait(): launched', seen[-1][0], 'done', seen[-1][1], 'need', seen[-1][2])
