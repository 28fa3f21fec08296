"""Launch the deep-layer conv kernels (enc5..enc8 / dec1..dec4 shapes, B = 256) a few times for rocprofv3 PMC passes:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma_bf16 \
        --output-format csv -- python3 profiles/pmc_deep.py bf16
    python3 profiles/pmc_deep.py --summarise gpurun_out/pmc_mfma_bf16 gpurun_out/pmc_mfma_fp32 > profiles/r1_mfma_util.json
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = {'G5': (64, 128, 4, 17, 23), 'G6': (128, 256, 4, 9, 12), 'G7': (256, 512, 4, 5, 7), 'G8': (512, 2048, 1, 3, 4),
          'L2': (8, 16, 4, 129, 174), 'L3': (16, 32, 4, 65, 88), 'L4': (32, 64, 4, 33, 45)}


def run(dtype, B=256, reps=3):
    import torch
    from preset_gen_vae_amd import ops
    ops.set_compute_dtype(dtype)
    for nm, (Cb, Cs, k, Hb, Wb) in SHAPES.items():
        g = ops.ConvGeom(Cb, Cs, k, 2 if k == 4 else 1, 2 if k == 4 else 0, Hb, Wb)
        big = torch.randn(B, Cb, Hb, Wb, device='cuda')
        small = torch.randn(B, Cs, g.Hs, g.Ws, device='cuda')
        w = torch.randn(Cs, Cb, k, k, device='cuda') * 0.05
        sc, sh = torch.ones(Cb, device='cuda'), torch.zeros(Cb, device='cuda')
        st = torch.empty(2 * Cs, device='cuda', dtype=torch.float64)
        bs = torch.zeros(Cs, device='cuda')
        gw = torch.empty_like(w)
        for _ in range(reps):
            ops.conv_down(g, big, w, bs, 1, 0.1, in_scale=sc, in_shift=sh, stats=st)
            ops.conv_up(g, small, w, None, 0, 0.0)
            ops.conv_wgrad(g, big, small, gw, big_scale=sc, big_shift=sh)
    torch.cuda.synchronize()


def summarise(dirs):
    out = {}
    for d in dirs:
        mode = 'bf16' if 'bf16' in d else 'fp32'
        acc = {}
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r['Kernel_Name']
                m = re.search(r'((?:deep_|k1_|conv_)\w+_kernel<[^>]*>)', name)
                if not m:
                    continue
                acc.setdefault(m.group(1), {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        for k, c in acc.items():
            mean = {n: sum(v) / len(v) for n, v in c.items()}
            busy, gui = mean.get('SQ_VALU_MFMA_BUSY_CYCLES'), mean.get('GRBM_GUI_ACTIVE')
            rec = {n: round(v, 1) for n, v in mean.items()}
            if busy is not None and gui:
                # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 256 CUs x 4 SIMDs (checked: = MFMA count x 16 cycles for
                # v_mfma_f32_16x16x16_bf16, x 32 for v_mfma_f32_16x16x4_f32); GRBM_GUI_ACTIVE is summed over the 8 XCDs
                rec['mfma_util'] = round(busy / (gui / 8.0 * 256 * 4), 4)
            out.setdefault(mode, {})[k] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    if sys.argv[1] == '--summarise':
        summarise(sys.argv[2:])
    else:
        run(sys.argv[1])
