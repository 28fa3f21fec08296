"""Timing of the launches of the pass-free BatchNorm backward as the 4-layer step issues them (graph replay, cold
operands, bench.time_kernel): every fused input-gradient call next to its plain form, tap / class sums, coefficient
kernel.  usage: time_fused.py [substring ...] [--policy N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from preset_gen_vae_amd import _lib, ops  # noqa: E402

# (name of the lower block, Cb, Cs, k, Hb, Wb, consumer is a ConvTranspose2d)
LAYERS = [('enc1<-enc2', 8, 16, 4, 129, 174, False), ('enc2<-enc3', 16, 32, 4, 65, 88, False),
          ('enc3<-enc4', 32, 64, 4, 33, 45, False), ('dec5<-dec6', 16, 32, 4, 65, 88, True),
          ('dec6<-dec7', 8, 16, 4, 129, 174, True), ('dec7<-dec8', 1, 8, 5, 257, 347, True)]


def main():
    pats = [a for a in sys.argv[1:] if not a.startswith('--')]
    pol = int(sys.argv[sys.argv.index('--policy') + 1]) if '--policy' in sys.argv else 0
    B = 256
    lib = _lib.load()
    lib.pgv_set_kernel_policy(pol)
    dev = torch.device('cuda', 0)
    print(f"{'launch':34s} {'us':>8s} {'MB alg':>8s} {'TB/s':>6s}")
    for name, Cb, Cs, k, Hb, Wb, up in LAYERS:
        if pats and not any(p in name for p in pats):
            continue
        geom = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
        big = torch.randn(B, Cb, Hb, Wb, device=dev)
        small = torch.randn(B, Cs, geom.Hs, geom.Ws, device=dev)
        w = torch.randn(Cs, Cb, k, k, device=dev) * 0.05
        gw = torch.randn_like(w)
        gy, lower = (big, small) if up else (small, big)     # consumer's output gradient / lower block's tensor
        C = lower.shape[1]
        a = torch.randn_like(lower)
        out = torch.empty_like(lower)
        coef = torch.cat([torch.ones(C, device=dev), 0.01 * torch.randn(2 * C, device=dev)])
        gb = torch.zeros(C, device=dev)
        fuse = (a, coef, gb, 1, 0.1)
        conv = ops.conv_down if up else ops.conv_up
        m = 2 if up else 1
        cls = torch.zeros(gy.shape[1] * m * m * (ops.CLS_COPIES if up else 1), device=dev)
        T = torch.zeros(gy.shape[1] * k * k, device=dev, dtype=torch.float64)
        sc, sh, mu, rs = (torch.ones(C, device=dev) for _ in range(4))
        gg, gbt = torch.empty(C, device=dev), torch.empty(C, device=dev)
        nbytes = lambda *ts: sum(t.numel() * 4 for t in ts)
        rows = [('dgrad plain', lambda: conv(geom, gy, w, None, 0, 0.0, out=out), nbytes(gy, out)),
                ('dgrad fused', lambda: conv(geom, gy, w, None, 0, 0.0, out=out, bwd_fuse=fuse), nbytes(gy, out, a)),
                ('dgrad fused, no bias grad', lambda: conv(geom, gy, w, None, 0, 0.0, out=out, bwd_fuse=(a, coef, None, 1, 0.1)), nbytes(gy, out, a)),
                ('dgrad fused + class sums', lambda: conv(geom, gy, w, None, 0, 0.0, out=out,
                                                          bwd_fuse=fuse + (torch.zeros(ops.CLS_COPIES * 4 * C, device=dev),)), nbytes(gy, out, a)),
                ('act_bwd_coef pass', lambda: ops.act_bwd_coef(out, a, coef, 1, 0.1, out, gb, prezeroed=True), nbytes(out, out, a)),
                ('class_sums', lambda: ops.conv_class_sums(geom, gy, up, cls, prezeroed=True), nbytes(gy)),
                ('tap_sums border', lambda: ops.conv_tap_sums(geom, gy, up, T, prezeroed=True, cls=cls), 0),
                ('tap_sums full', lambda: ops.conv_tap_sums(geom, gy, up, T, prezeroed=True), nbytes(gy)),
                ('bn_bwd_coef', lambda: ops.bn_bwd_coef(geom, B, not up, w, gw, T, sc, sh, mu, rs, lower.numel() // C, coef, gg, gbt), 0)]
        for label, fn, byt in rows:
            us = bench.time_kernel(fn, iters=5) * 1e3
            print(f"{name + ' ' + label:34s} {us:8.1f} {byt / 1e6:8.1f} {byt / max(us, 1e-3) / 1e6:6.2f}", flush=True)
    lib.pgv_set_kernel_policy(0)


if __name__ == '__main__':
    main()
