"""Launch one conv kernel of the bench configuration a few times (for `rocprofv3 --pmc ...` passes).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 profiles/pmc_driver.py wgrad enc2
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 profiles/pmc_driver.py wgrad enc2
    python3 profiles/pmc_driver.py --summarise gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r1_traffic.json
"""
import csv
import glob
import re
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LAYERS = {'enc1': (1, 8, 5, 257, 347), 'enc2': (8, 16, 4, 129, 174), 'enc3': (16, 32, 4, 65, 88),
          'enc4': (32, 64, 4, 33, 45)}


def run(kind, layer, B=256, reps=5):
    import torch
    from preset_gen_vae_amd import ops
    Cb, Cs, k, Hb, Wb = LAYERS[layer]
    g = ops.ConvGeom(Cb, Cs, k, 2, 2, Hb, Wb)
    big = torch.randn(B, Cb, Hb, Wb, device='cuda')
    small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
    w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
    sc, sh = torch.ones(Cb, device='cuda'), torch.zeros(Cb, device='cuda')
    bias = torch.zeros(Cs, device='cuda')
    st = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
    out_s, gw = torch.empty_like(small), torch.empty_like(w)
    for _ in range(reps):
        if kind == 'wgrad':
            ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh)
        elif kind == 'down':
            ops.conv_down(g, big, w, bias, 1, 0.1, in_scale=sc, in_shift=sh, stats=st, out=out_s)
        elif kind == 'up':
            ops.conv_up(g, small, w, None, 0, 0.0, out=big)
        else:
            raise SystemExit(kind)
    torch.cuda.synchronize()


def summarise(dirs):
    res = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                name = r['Kernel_Name']
                if 'conv_' not in name and '_c1_kernel' not in name:
                    continue
                m = re.search(r'(\w+_kernel(?:<[^>]*>)?)', name)
                key = m.group(1) if m else name[:60]
                res.setdefault(key, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    out = {}
    for k, c in res.items():
        fetch = sum(c.get('FETCH_SIZE', [0])) / max(1, len(c.get('FETCH_SIZE', [0])))
        write = sum(c.get('WRITE_SIZE', [0])) / max(1, len(c.get('WRITE_SIZE', [0])))
        # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced
        # streaming reads -> doubled; WRITE_SIZE is exact for 16-byte streaming stores and float atomics.
        out[k] = {'FETCH_SIZE_KiB': fetch, 'WRITE_SIZE_KiB': write,
                  'hbm_bytes_per_launch': (2.0 * fetch + write) * 1024.0}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == '--summarise':
        summarise(sys.argv[2:])
    else:
        run(sys.argv[1], sys.argv[2])
