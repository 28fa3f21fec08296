"""First-step exchanged gradients of the 2-rank launch modes against each other, bucket by bucket, with a repeat of the
same mode as the noise floor (tests/test_gpu_distributed.py's worker)."""
import os, sys, tempfile, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_distributed as T
from pathlib import Path
tmp = Path(tempfile.mkdtemp())
worker = tmp / "worker.py"
worker.write_text(T.WORKER)
grads = {}
for mode in ("two-graph", "two-graph-again", "bucket-graphs", "eager-rng", "eager-rng-again"):
    out = str(tmp / ("state_" + mode))
    launch = "eager-rng" if mode.startswith("eager-rng") else mode
    env = dict(os.environ, PGV_ROOT=ROOT, PGV_OUT=out, PGV_SHARDS="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(T._free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PGV_LAUNCH=launch, PGV_PRODUCTS=os.environ.get("PRODUCTS", "native"))
    T._run_two_ranks(worker, env)
    grads[mode] = torch.load(out + ".grad0")
rg = grads["two-graph"]["ranges"]
for a, b in itertools.combinations(grads, 2):
    errs = []
    for lo, hi in rg:
        x, y = grads[a]["flat_grad"][lo:hi].double(), grads[b]["flat_grad"][lo:hi].double()
        errs.append(((x - y).norm() / y.norm()).item())
    print(f"{a:16s} vs {b:16s}: " + "  ".join(f"{e:.2e}" for e in errs))
