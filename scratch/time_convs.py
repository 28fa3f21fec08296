"""A/B timing of conv launches: second-generation kernels (policy 0) vs the band kernels (policy 3), measured the way
bench.py measures its roofline table (graph replay, cold operands).  usage: time_convs.py [substring ...]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from preset_gen_vae_amd import _lib, config  # noqa: E402
from preset_gen_vae_amd.model import build as mbuild  # noqa: E402


def main():
    pats = [a for a in sys.argv[1:] if not a.startswith('--')]
    arch = 'speccnn8l1_bn' if '--8l' in sys.argv else 'speccnn4l1_bn'
    B = 256
    lib = _lib.load()
    from preset_gen_vae_amd import ops
    ops.set_compute_dtype(os.environ.get('DTYPE', 'fp32'))
    mc, tc = copy.copy(config.model), copy.copy(config.train)
    mc.encoder_architecture, mc.dim_z, mc.input_tensor_size = arch, 64, (B, 1, 257, 347)
    tc.latent_flow_input_regularization = 'none'
    _, _, ae = mbuild.build_ae_model(mc, tc)
    ae = ae.cuda().train()
    dev = torch.device('cuda', 0)
    table = bench.launch_table(ae, B, dev)
    print(f"{'launch':28s} {'v2 us':>9s} {'band us':>9s} {'roof us':>9s} {'frac v2':>8s}")
    for label, fn, byt, fl in table:
        if pats and not any(p in label for p in pats):
            continue
        res = []
        for pol in (0, 3):
            lib.pgv_set_kernel_policy(pol)
            res.append(bench.time_kernel(fn, iters=5) * 1e3)
        lib.pgv_set_kernel_policy(0)
        roof = max(byt / 8e12, fl / 157.3e12) * 1e6
        print(f"{label:28s} {res[0]:9.1f} {res[1]:9.1f} {roof:9.1f} {roof / res[0]:8.3f}", flush=True)


if __name__ == '__main__':
    main()
