"""Cold-operand timing of selected launches of the train step (bench.launch_table), optionally with a variant library.

    python scratch/time_launches.py [--lib scratch/libX.so] [--match REGEX] [--products bf16x6|native] [--dtype fp32|bf16]
                                    [--arch speccnn4l1_bn] [--reps 5] [--tag NAME]

Prints one line per launch: label, median ms of `reps` 5-launch averages, bytes / ms as TB/s.  A/B on one box: call it
twice in the same gpurun command with different --lib.
"""
import argparse
import copy
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--match", default=".")
ap.add_argument("--products", default="bf16x6")
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--arch", default="speccnn4l1_bn")
ap.add_argument("--dim-z", type=int, default=64)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tag", default="")
ap.add_argument("--json", default=None)
args = ap.parse_args()

import preset_gen_vae_amd  # noqa: E402,F401
from preset_gen_vae_amd import _lib  # noqa: E402

if args.lib:
    _lib.LIB_PATH = os.path.join(ROOT, args.lib)
import torch  # noqa: E402

import bench  # noqa: E402
from preset_gen_vae_amd import config, ops  # noqa: E402
from preset_gen_vae_amd.model import build as mbuild  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
ops.set_compute_dtype(args.dtype)
if args.dtype == 'fp32':
    ops.set_fp32_products(args.products)
mc, tc = copy.copy(config.model), copy.copy(config.train)
mc.encoder_architecture, mc.dim_z, mc.input_tensor_size = args.arch, args.dim_z, (args.batch, 1, 257, 347)
_, _, ae = mbuild.build_ae_model(mc, tc)
ae = ae.to(dev).train()
pat = re.compile(args.match)
rows = []
tot = 0.0
for label, fn, bytes_, flops in bench.launch_table(ae, args.batch, dev):
    if not pat.search(label):
        continue
    ms = sorted(bench.time_kernel(fn, iters=5) for _ in range(args.reps))[args.reps // 2]
    tot += ms
    rows.append({'launch': label, 'ms': ms, 'bytes': bytes_, 'flops': flops})
    print(f"{args.tag:10s} {label:44s} {ms * 1e3:8.1f} us  {bytes_ / (ms * 1e-3) / 1e12 if ms > 0 else 0:6.2f} TB/s"
          f"  {flops / (ms * 1e-3) / 1e12 if ms > 0 else 0:7.1f} TF/s", flush=True)
print(f"{args.tag:10s} {'SUM':44s} {tot * 1e3:8.1f} us")
if args.json:
    with open(os.path.join(ROOT, args.json), 'w') as f:
        json.dump(rows, f, indent=1)
