import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, torch.nn.functional as F
import preset_gen_vae_amd
from preset_gen_vae_amd import ops
import test_gpu_kernels as T
case = (16, 32, 4, 2, 2, 65, 88, 3)
big, small, w, bias_s, bias_b, sc_b, sh_b, sc_s, sh_s, Hs, Ws = T._conv_inputs(case)
dev = T.dev
g = ops.ConvGeom(16, 32, 4, 2, 2, 65, 88)
print('small', small.abs().max().item(), small.dtype, 'w', w.abs().max().item(), 'sc', sc_s.min().item(), sc_s.max().item())
def run(tag, **kw):
    use_aff = kw.get('aff', False); use_bias = kw.get('bias', False); act = kw.get('act', 0)
    x = T._affine_fma(small, sc_s, sh_s).double() if use_aff else small.double()
    ref = F.conv_transpose2d(x, w.double(), bias_b.double() if use_bias else None, stride=2, padding=2, output_padding=(1,0))
    if act: ref = F.leaky_relu(ref, 0.1)
    args = dict(in_scale=dev(sc_s), in_shift=dev(sh_s)) if use_aff else {}
    res = {}
    for mode in ('native', 'bf16x6'):
        ops.set_fp32_products(mode)
        sh = ops.conv_weight_shadow(g, dev(w))
        kk = dict(w_shadow=sh) if sh is not None else {}
        o = ops.conv_up(g, dev(small), dev(w), dev(bias_b) if use_bias else None, ops.PGV_ACT_LEAKY_RELU if act else ops.PGV_ACT_NONE, 0.1 if act else 0.0, **args, **kk)
        res[mode] = ((o.double().cpu()-ref).norm()/ref.norm()).item()
    ops.set_fp32_products('native')
    print(tag, res)
run('plain')
run('aff', aff=True)
run('bias', bias=True)
run('act', act=1)
run('all', aff=True, bias=True, act=1)
