"""Where do the device-to-device copies of a train step come from?  (eager step under torch.profiler, aten::copy_ / contiguous
call sites)"""
import sys, os, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from preset_gen_vae_amd import config
from preset_gen_vae_amd.model import build as mbuild
from preset_gen_vae_amd.train_step import VAETrainStep
arch = sys.argv[1] if len(sys.argv) > 1 else 'speccnn4l1_bn'
B = 32
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture, mc.dim_z, mc.input_tensor_size = arch, 64, (B, 1, 257, 347)
_, _, ae = mbuild.build_ae_model(mc, tc)
ae = ae.cuda().train()
step = VAETrainStep(ae, use_graph=False)
x = torch.randn(B, 1, 257, 347, device='cuda').clamp(-1, 1)
for _ in range(2):
    step.step(x)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.step(x)
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if any(k in ev.name for k in ('copy', 'Memcpy', 'clone', 'contiguous', 'fill_', 'zero_', 'cat')):
        st = [f.strip()[-80:] for f in (ev.stack or []) if 'site-packages/torch' not in f and 'dist-packages/torch' not in f][:4]
        key = (ev.name, str(ev.input_shapes)[:60], ' <- '.join(st))
        seen[key] = seen.get(key, 0) + 1
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, *k, sep=' | ')
