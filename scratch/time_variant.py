"""A/B timing of one launch label of bench.launch_table under a debug knob of the library:
usage: time_variant.py <setter symbol> <label substring> [values...]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from preset_gen_vae_amd import _lib, config
from preset_gen_vae_amd.model import build as mbuild
setter, pats = sys.argv[1], sys.argv[2].split(',')
vals = [int(v) for v in sys.argv[3:]] or [0, 1]
arch = os.environ.get('ARCH', 'speccnn4l1_bn')
B = 256
lib = _lib.load()
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture, mc.dim_z, mc.input_tensor_size = arch, 64, (B, 1, 257, 347)
_, _, ae = mbuild.build_ae_model(mc, tc)
ae = ae.cuda().train()
table = bench.launch_table(ae, B, torch.device('cuda', 0))
for rep in range(3):
    for label, fn, byt, fl in table:
        if not any(p in label for p in pats):
            continue
        ts = []
        for v in vals:
            getattr(lib, setter)(v)
            ts.append(bench.time_kernel(fn, iters=5) * 1e3)
        getattr(lib, setter)(0)
        print(f"{label:36s} " + "  ".join(f"v{v}: {t:6.1f}" for v, t in zip(vals, ts)), flush=True)
