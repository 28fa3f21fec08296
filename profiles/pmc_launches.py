"""HBM traffic / counters per launch of the train step, keyed by the labels of bench.launch_table (rounds 2-3).

Every launch of the table is issued REPS times behind a sentinel kernel (torch erfinv_ on a 1-element tensor), so the
dispatch stream of one process splits into one segment per label; all kernels of a segment (e.g. the wgrad kernel AND its
reduce pass) are added up and divided by REPS.  Counters are collected in separate rocprofv3 passes (--kernel-trace and
--pmc only):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 profiles/pmc_launches.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 profiles/pmc_launches.py run
    python3 profiles/pmc_launches.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r3_traffic.json
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma ... run
    python3 profiles/pmc_launches.py mfma gpurun_out/pmc_mfma > profiles/r3_mfma_util.json
(scratch/profile_final.sh runs all of it)
"""
import csv
import re
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 3
LABELS_FILE = os.path.join(ROOT, 'gpurun_out', os.environ.get('PMC_LABELS', 'pmc_labels.json'))
SENTINEL = 'erfinv'   # substring of the sentinel's kernel name (at::native::erfinv_kernel_cuda ...)


def build_table():
    import copy
    import torch
    import bench
    from preset_gen_vae_amd import config
    from preset_gen_vae_amd.model import build as mbuild
    from preset_gen_vae_amd.utils import audio
    B = 256
    # PMC_ARCH / PMC_DZ / PMC_DTYPE select the configuration (round 4: the 8-layer and the bf16 z = 512 steps too)
    from preset_gen_vae_amd import ops
    ops.set_compute_dtype(os.environ.get('PMC_DTYPE', 'fp32'))
    ops.set_fp32_products(os.environ.get('PMC_FP32_PRODUCTS', 'native'))   # 'bf16x6': PGV_COMPUTE_F32_SPLIT
    mc, tc = copy.copy(config.model), copy.copy(config.train)
    mc.encoder_architecture = os.environ.get('PMC_ARCH', 'speccnn4l1_bn')
    mc.dim_z, mc.input_tensor_size = int(os.environ.get('PMC_DZ', 64)), (B, 1, 257, 347)
    _, _, ae = mbuild.build_ae_model(mc, tc)
    dev = torch.device('cuda', 0)
    ae = ae.to(dev).train()
    fe = audio.MelSpectrogram(1024, 256, -120.0, 257, 22050, device=dev)
    return bench.launch_table(ae, B, dev, frontend=fe), bench


def run():
    import torch
    table, _ = build_table()
    one = torch.full((1,), 0.5, device='cuda')
    labels = []
    for label, fn, byt, fl in table:
        one.erfinv_()
        for _ in range(REPS):
            fn()
        labels.append({'launch': label, 'algorithmic_bytes': byt, 'algorithmic_flops': fl})
    one.erfinv_()
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(LABELS_FILE), exist_ok=True)
    json.dump(labels, open(LABELS_FILE, 'w'))


def segments(d):
    """[(label dict, {counter: sum over the segment's kernels / REPS}, [kernel names])] of one rocprofv3 output dir."""
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    labels = json.load(open(LABELS_FILE))
    segs, cur, seen = [], None, set()
    last_id = None
    for r in rows:
        name = r['Kernel_Name']
        if SENTINEL in name.lower():
            if r['Dispatch_Id'] != last_id:      # one row per counter per dispatch
                if cur is not None:
                    segs.append(cur)
                cur = ({}, [])
            last_id = r['Dispatch_Id']
            continue
        if cur is None:
            continue
        cur[0][r['Counter_Name']] = cur[0].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        key = (r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            m = re.search(r'(\w+_kernel\w*|__amd_\w+|\w+Kernel\w*)', name)
            short = m.group(1) if m else name[:48]
            if short not in cur[1]:
                cur[1].append(short)
    assert len(segs) == len(labels), (len(segs), len(labels))
    return [(lab, {k: v / REPS for k, v in c.items()}, names) for lab, (c, names) in zip(labels, segs)]


def traffic(fetch_dir, write_dir):
    out = {'_about': 'HBM traffic per launch (all kernels of the launch) from rocprofv3 PMC passes, FETCH_SIZE and '
                     'WRITE_SIZE in separate passes with --kernel-trace only, profiles/pmc_launches.py; KiB units; '
                     'FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section: gfx950 reports half of the bytes of wide '
                     'coalesced streaming reads). Keys = bench.py launch labels; bench.py reports the entry of its roofline '
                     'kernel as roofline.traffic. Batch 256, operands resident and re-used across the 3 repetitions of a '
                     'launch (small tensors may be served from the 256 MB MALL).'}
    f = segments(fetch_dir)
    w = segments(write_dir)
    for (lab, cf, names), (_, cw, _) in zip(f, w):
        fetch, write = cf.get('FETCH_SIZE', 0.0), cw.get('WRITE_SIZE', 0.0)
        hbm = (2.0 * fetch + write) * 1024.0
        out[lab['launch']] = {'FETCH_SIZE_KiB': round(fetch, 1), 'WRITE_SIZE_KiB': round(write, 1),
                              'hbm_bytes_per_launch': hbm, 'algorithmic_bytes': lab['algorithmic_bytes'],
                              'traffic_over_algorithmic': round(hbm / max(1, lab['algorithmic_bytes']), 3),
                              'kernels': names}
    print(json.dumps(out, indent=1))


def mfma(d):
    out = {'_about': 'matrix-pipe utilisation per launch: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / '
                     '(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs '
                     '(1.6 M for a 75 us kernel). profiles/pmc_launches.py, batch 256, configuration ' + os.environ.get('PMC_ARCH', 'speccnn4l1_bn') + ' z=' + os.environ.get('PMC_DZ', '64') + ' ' + os.environ.get('PMC_DTYPE', 'fp32') + ' fp32 products ' + os.environ.get('PMC_FP32_PRODUCTS', 'native')}
    for lab, c, names in segments(d):
        if lab['algorithmic_flops'] <= 0:
            continue
        busy, act = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), c.get('GRBM_GUI_ACTIVE', 0.0)
        out[lab['launch']] = {k: round(v, 1) for k, v in c.items()}
        out[lab['launch']]['mfma_busy_frac'] = round(busy / (act / 8.0 * 1024.0), 4) if act else None
        out[lab['launch']]['kernels'] = names
    print(json.dumps(out, indent=1))


def counters(dirs):
    """Raw SQ counters per launch label from one or more passes (round 4: SQ_INSTS_VALU / SQ_WAIT_INST_LDS / SQ_BUSY_CYCLES
    ... behind DESIGN.md 3.9), plus two ratios: wave cycles parked (s_waitcnt / barrier) and VALU instructions per MFMA."""
    out = {'_about': 'SQ counters per launch (sum over the launch\'s kernels, per repetition), rocprofv3 --kernel-trace --pmc '
                     'passes of profiles/pmc_launches.py; configuration ' + os.environ.get('PMC_ARCH', 'speccnn4l1_bn') +
                     ' z=' + os.environ.get('PMC_DZ', '64') + ' ' + os.environ.get('PMC_DTYPE', 'fp32') + ' fp32 products ' + os.environ.get('PMC_FP32_PRODUCTS', 'native') +
                     '.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles '
                     '(MI355X_MICROARCH.md).'}
    merged = {}
    for d in dirs:
        for lab, c, names in segments(d):
            e = merged.setdefault(lab['launch'], {'kernels': names})
            e.update({k: round(v, 1) for k, v in c.items()})
    for k, e in merged.items():
        if e.get('SQ_WAVE_CYCLES'):
            e['parked_frac_of_wave_cycles'] = round(e.get('SQ_WAIT_ANY', 0.0) / e['SQ_WAVE_CYCLES'], 4)
            e['issue_stall_frac_of_wave_cycles'] = round(e.get('SQ_WAIT_INST_ANY', 0.0) / e['SQ_WAVE_CYCLES'], 4)
        if e.get('SQ_INSTS_VALU_MFMA_MOPS_F32') and e.get('SQ_INSTS_VALU'):
            # MOPS_F32 counts 512 MACs per unit: one v_mfma_f32_16x16x4_f32 = 2 units
            e['valu_insts_per_f32_mfma'] = round(e['SQ_INSTS_VALU'] / (e['SQ_INSTS_VALU_MFMA_MOPS_F32'] / 2.0), 3)
        out[k] = e
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'counters':
        counters(sys.argv[2:])
    elif sys.argv[1] == 'run':
        run()
    elif sys.argv[1] == 'traffic':
        traffic(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == 'mfma':
        mfma(sys.argv[2])
