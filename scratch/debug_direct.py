import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F, numpy as np
from helpers import synth_vec, rel_l2
from preset_gen_vae_amd import ops
dev = lambda t: t.to('cuda', torch.float32).contiguous()
for B in (2, 5):
    Cb, Cs, k, s, p, Hb, Wb = 1, 8, 5, 2, 2, 257, 347
    g = ops.ConvGeom(Cb, Cs, k, s, p, Hb, Wb)
    big = synth_vec((B, Cb, Hb, Wb), 0.9137, 0.3) * 1.5
    small = synth_vec((B, Cs, g.Hs, g.Ws), 0.7719, 1.1)
    w = synth_vec((Cs, Cb, k, k), 0.6180, 0.7) * 0.4
    bs = synth_vec((Cs,), 1.37, 0.2) * 0.1; bb = synth_vec((Cb,), 1.73, 0.5) * 0.1
    scs, shs = 1.0 + 0.2 * synth_vec((Cs,), 3.1, 0.4), 0.3 * synth_vec((Cs,), 3.7, 0.9)
    scb, shb = 1.0 + 0.2 * synth_vec((Cb,), 2.1, 0.1), 0.3 * synth_vec((Cb,), 2.9, 0.6)
    aff = lambda t, sc, sh: t * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = F.leaky_relu(F.conv2d(big, w, bs, stride=2, padding=2), 0.1)
    got = ops.conv_down(g, dev(big), dev(w), dev(bs), 1, 0.1)
    d = (got.double().cpu() - ref).abs(); print(B, 'down direct', rel_l2(got, ref), d.max().item(), np.unravel_index(d.argmax().item(), d.shape))
    ref = F.conv2d(aff(big, scb, shb), w, None, stride=2, padding=2)
    got = ops.conv_down(g, dev(big), dev(w), None, 0, 0.0, in_scale=dev(scb), in_shift=dev(shb))
    print(B, 'down direct affine', rel_l2(got, ref))
    ref = F.hardtanh(F.conv_transpose2d(aff(small, scs, shs), w, bb, stride=2, padding=2))
    got = ops.conv_up(g, dev(small), dev(w), dev(bb), 2, 0.0, in_scale=dev(scs), in_shift=dev(shs))
    d = (got.double().cpu() - ref).abs(); print(B, 'up direct', rel_l2(got, ref), d.max().item(), np.unravel_index(d.argmax().item(), d.shape))
    ref = F.conv_transpose2d(small, w, None, stride=2, padding=2)
    got = ops.conv_up(g, dev(small), dev(w), None, 0, 0.0)
    print(B, 'up direct plain', rel_l2(got, ref))
    wv = w.clone().requires_grad_(True); y = F.conv2d(aff(big, scb, shb), wv, None, stride=2, padding=2); y.backward(aff(small, scs, shs))
    gw = torch.empty((Cs, Cb, k, k), device='cuda')
    ops.conv_wgrad(g, dev(big), dev(small), gw, big_scale=dev(scb), big_shift=dev(shb), small_scale=dev(scs), small_shift=dev(shs))
    print(B, 'wgrad direct', rel_l2(gw, wv.grad))
