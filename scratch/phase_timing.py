"""In-kernel phase stamps of the band kernels (debug library built by scratch/build_dbg.sh)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', os.environ.get('PGV_DBG_LIB', 'libpgv_hip_dbg.so'))
from preset_gen_vae_amd import ops
lib = _lib.load()
lib.pgv_dbg_set_tlog.argtypes = [ctypes.c_void_p]
lib.pgv_dbg_set_tlog_band.argtypes = [ctypes.c_void_p]
B = 256
tlog = torch.zeros(1 << 16, 4, 8, dtype=torch.int64, device='cuda')


def run(name, fn, n_wg):
    lib.pgv_dbg_set_tlog(None); lib.pgv_dbg_set_tlog_band(None)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tlog.zero_()
    lib.pgv_dbg_set_tlog(ctypes.c_void_p(tlog.data_ptr())); lib.pgv_dbg_set_tlog_band(ctypes.c_void_p(tlog.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    lib.pgv_dbg_set_tlog(None); lib.pgv_dbg_set_tlog_band(None)
    raw = tlog[:n_wg].cpu().numpy()
    if raw[0, 0, 7] > 0 and raw[0, 0, 7] < 10000:  # band kernels: accumulated durations, slot 7 = items
        okb = raw[:, 0, 7] > 0
        d = raw[okb].astype(np.float64); n = d[:, :, 7:8]; d = d[:, :, :7] / 100.0
        names = ['wait-barrier', 'commit(LDS write)', 'weights', 'issue next', 'barrier', 'MFMA', 'epilogue']
        print(f"== {name}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {okb.sum()} persistent workgroups, "
              f"items/wg {n.mean():.2f}; per-item mean us (wave 0..3):")
        for i, nm in enumerate(names):
            print(f"   {nm:18s} " + ' '.join(f"{(d[:, wv, i] / n[:, wv, 0]).mean():6.2f}" for wv in range(4)) +
                  f"   total/wg {d[:, 0, i].mean():7.2f}")
        print(f"   sum per item {(d[:, 0, :].sum(1) / n[:, 0, 0]).mean():.2f} us, per wg {d[:, 0, :].sum(1).mean():.1f} us")
        return
    t = raw.astype(np.float64) / 100.0  # us, [wg][wave][stamp]
    ok = t[:, 0, 0] > 0
    if not ok.any():
        print(f'== {name}: kernel {e0.elapsed_time(e1)*1e3:.1f} us (no stamps)'); return
    t = t[ok]
    t0 = t[:, :, 0].min()
    raw = tlog[:n_wg].cpu().numpy()[ok.nonzero()[0] if False else slice(None)]
    nph = int((t[0, 0, :7] > 0).sum())
    hw = tlog[:n_wg].cpu().numpy()[:, 0, 7]
    hw = hw[tlog[:n_wg, 0, 0].cpu().numpy() > 0]
    if hw.any():
        xcc = (hw >> 32) & 0xf; lo = hw & 0xffffffff
        cu = (lo >> 8) & 0xf; sh = (lo >> 12) & 1; se = (lo >> 13) & 7
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        ids, counts = np.unique(cuid, return_counts=True)
        print(f"   {len(ids)} distinct CUs, WGs per CU min {counts.min()} max {counts.max()}")
        # phase relation of co-resident workgroups: for each CU sort its WGs by start, print a few timelines
        for c in ids[:3]:
            sel = np.where(cuid == c)[0]
            order = sel[np.argsort(t[sel, 0, 0])]
            print("   CU", c, "wg ids", order[:6].tolist())
            for w in order[:6]:
                print("      ", ' '.join(f"{x - t0:7.2f}" for x in t[w, 0, :nph]))

    print(f"== {name}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {ok.sum()} workgroups stamped, {nph} stamps")
    for i in range(1, nph):
        dph = t[:, :, i] - t[:, :, i - 1]
        print(f"   phase {i-1}->{i}: per wave mean " + ' '.join(f"{dph[:, wv].mean():6.2f}" for wv in range(4)) +
              f"   all p10 {np.percentile(dph,10):6.2f} p90 {np.percentile(dph,90):6.2f}")
    life = t[:, :, nph - 1].max(1) - t[:, :, 0].min(1)
    print(f"   lifetime mean {life.mean():.2f} us; last end {t[:, :, nph-1].max()-t0:.1f} us")


for which, (Cb, Cs, k, Hb, Wb) in {'L2': (8, 16, 4, 129, 174), 'L3': (16, 32, 4, 65, 88), 'L4': (32, 64, 4, 33, 45)}.items():
    g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda'); w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    out = torch.empty_like(small); bias = torch.zeros(Cs, device='cuda'); gw = torch.empty_like(w)
    st = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
    sc = torch.ones(Cb, device='cuda'); sh = torch.zeros(Cb, device='cuda')
    run(f'down {which}', lambda: ops.conv_down(g, big, w, bias, 1, 0.1, in_scale=sc, in_shift=sh, stats=st, out=out), 1 << 16)
    run(f'wgrad {which}', lambda: ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh), 512)

print("---- up kernels (dgrad form)")
for which, (Cb, Cs, k, Hb, Wb) in {'L2': (8, 16, 4, 129, 174), 'L3': (16, 32, 4, 65, 88), 'L4': (32, 64, 4, 33, 45)}.items():
    g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
    w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda'); outb = torch.empty(B, Cb, Hb, Wb, device='cuda')
    run(f'up {which}', lambda: ops.conv_up(g, small, w, None, 0, 0.0, out=outb), 512)
