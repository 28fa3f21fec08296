"""Time every conv_wgrad call inside an eager train step, then again on clones of the same tensors."""
import sys, os, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from preset_gen_vae_amd import config, ops
from preset_gen_vae_amd.model import build
from preset_gen_vae_amd.train_step import VAETrainStep
B = 256
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture = 'speccnn4l1_bn'; mc.input_tensor_size = (B, 1, 257, 347)
enc, dec, ae = build.build_ae_model(mc, tc)
ae = ae.cuda().train()
ts = VAETrainStep(ae, use_graph=False)
x = torch.rand(B, 1, 257, 347, device='cuda') * 2 - 1
for _ in range(3): ts.step(x)
torch.cuda.synchronize()
orig = ops.conv_wgrad
log = []
def timed(geom, big, small, gw, **kw):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(geom, big, small, gw, **kw); e1.record(); torch.cuda.synchronize()
    t_in = e0.elapsed_time(e1) * 1e3
    b2, s2, g2 = big.clone(), small.clone(), torch.empty_like(gw)
    def tm(bb, ss, gg):
        torch.cuda.synchronize()
        tot = 0
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); orig(geom, bb, ss, gg, **kw); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1) * 1e3
        return round(tot / 3, 1)
    log.append((geom.Cb, geom.Cs, geom.Hb, 'in-step', round(t_in, 1), 'orig', tm(big, small, gw), 'clone big', tm(b2, small, gw),
                'clone small', tm(big, s2, gw), 'clone gw', tm(big, small, g2), 'all clones', tm(b2, s2, g2),
                hex(big.data_ptr()), hex(small.data_ptr()), hex(gw.data_ptr()), hex(b2.data_ptr()), hex(s2.data_ptr()), hex(g2.data_ptr())))
    return r
ops.conv_wgrad = timed
import preset_gen_vae_amd.model.layer as L
L.ops.conv_wgrad = timed
ts.step(x)
for l in log: print(l)
