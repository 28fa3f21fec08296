"""Per-replay time of the captured default step: events around each of the first N replays after the capture."""
import copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
import bench
from preset_gen_vae_amd import config
from preset_gen_vae_amd.model import build as mbuild
from preset_gen_vae_amd.train_step import VAETrainStep
dev = torch.device('cuda:0')
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.input_tensor_size = (256, 1, 257, 347)
_, _, ae = mbuild.build_ae_model(mc, tc)
ae = ae.to(dev).train()
x = bench.synth_spectrograms(256, dev, 1)
step = VAETrainStep(ae, use_graph=True)
step.step(x)
x = step.static_input
torch.cuda.synchronize()
N = int(os.environ.get('N', 80))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    step.step(x)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print('first 12:', [round(t, 3) for t in ts[:12]])
print('mean 0-4 %.4f  5-24 %.4f  25-44 %.4f  45-79 %.4f' % (sum(ts[:5]) / 5, sum(ts[5:25]) / 20, sum(ts[25:45]) / 20, sum(ts[45:80]) / 35))
