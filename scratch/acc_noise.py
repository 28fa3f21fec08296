"""fp32 accumulation noise of the deep-layer products on the kernel tests' structured inputs: rel-L2 against float64 and
the same error relative to the convolution of the absolute values (the scale rounding errors are proportional to)."""
import sys, os, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_kernels as tk
from preset_gen_vae_amd import _lib, ops
from helpers import rel_l2
lib = _lib.load()
for case in tk.CONV_CASES[4:7]:
    Cb, Cs, k, s, p, Hb, Wb, B = case
    big, small, w, *_ = tk._conv_inputs(case)
    geom = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    ref = F.conv2d(big, w, None, stride=s, padding=p)
    refabs = F.conv2d(big.abs(), w.abs(), None, stride=s, padding=p)
    for pol in (0, 3, 2, 1):
        lib.pgv_set_kernel_policy(pol)
        got = ops.conv_down(geom, big.float().cuda(), w.float().cuda(), None, 0, 0.0).double().cpu()
        e = (got - ref).norm().item()
        print(case[:7], 'policy', pol, f'rel_l2 {e / ref.norm().item():.2e}  err/|abs conv| {e / refabs.norm().item():.2e}  |ref|/|abs| {ref.norm().item() / refabs.norm().item():.3f}')
lib.pgv_set_kernel_policy(0)
