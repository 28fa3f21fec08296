"""HBM traffic of the front-end kernel (STFT -> mel -> dB of 256 waveforms, BASELINE config 5) from rocprofv3 PMC passes.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fe_fetch --output-format csv -- python3 profiles/pmc_frontend.py run
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_fe_write --output-format csv -- python3 profiles/pmc_frontend.py run
  python3 profiles/pmc_frontend.py traffic gpurun_out/pmc_fe_fetch gpurun_out/pmc_fe_write > profiles/r6_traffic_frontend.json

Same units and gfx950 correction as pmc_launches.py (KiB; FETCH_SIZE doubled)."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, N = 256, 88576


def run():
    sys.path.insert(0, ROOT)
    import torch
    import preset_gen_vae_amd  # noqa: F401
    from preset_gen_vae_amd.utils.audio import MelSpectrogram
    mel = MelSpectrogram(1024, 256, -120.0, 257, 22050)
    x = torch.randn(B, N, device='cuda') * 0.1
    out = torch.empty(B, 1, 257, 347, device='cuda')
    for _ in range(4):
        mel.batch(x, out=out)
    torch.cuda.synchronize()


def per_dispatch(d, counter):
    vals = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'stft_mel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                vals[r['Dispatch_Id']] = vals.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    v = list(vals.values())[1:]          # (the first launch also sets the kernel's LDS attribute / loads code)
    return sum(v) / len(v), len(v)


def traffic(fd, wd):
    fetch, n = per_dispatch(fd, 'FETCH_SIZE')
    write, _ = per_dispatch(wd, 'WRITE_SIZE')
    alg = B * (N + 257 * 347) * 4
    hbm = (2.0 * fetch + write) * 1024.0
    print(json.dumps({'_about': 'front-end kernel stft_mel_g8_kernel, 256 waveforms of 88 576 samples -> [256, 1, 257, 347]; '
                                'profiles/pmc_frontend.py, mean of %d launches' % n,
                      'stft_mel': {'FETCH_SIZE_KiB': round(fetch, 1), 'WRITE_SIZE_KiB': round(write, 1), 'hbm_bytes_per_launch': hbm,
                                   'algorithmic_bytes': alg, 'traffic_over_algorithmic': round(hbm / alg, 3)}}, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    else:
        traffic(sys.argv[2], sys.argv[3])
