"""eager-mode steps: allocator statistics and per-step host time, to find a per-step stall"""
import sys, os, time, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preset_gen_vae_amd import config
from preset_gen_vae_amd.model import build as mbuild
from preset_gen_vae_amd.train_step import VAETrainStep
B = 256
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture, mc.dim_z, mc.input_tensor_size = 'speccnn4l1_bn', 64, (B, 1, 257, 347)
tc.latent_flow_input_regularization = 'bn'
_, _, ae = mbuild.build_ae_model(mc, tc)
ae = ae.cuda().train()
step = VAETrainStep(ae, use_graph=False)
x = torch.randn(B, 1, 257, 347, device='cuda').clamp_(-1, 1)
for i in range(30):
    torch.cuda.synchronize(); s0 = torch.cuda.memory_stats(); t0 = time.perf_counter()
    step.step(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    if i >= 20 or i < 3:
        print(i, f"host {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms, device allocs {s1['num_device_alloc']-s0['num_device_alloc']}, "
                 f"frees {s1['num_device_free']-s0['num_device_free']}, retries {s1['num_alloc_retries']-s0['num_alloc_retries']}, "
                 f"reserved {s1['reserved_bytes.all.current']>>20} MB")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    step.step(x)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
