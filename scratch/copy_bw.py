"""Practical ceiling for a large-plane launch: pgv_copy (16-byte grid-stride streaming copy) over the same byte volume,
cold operands, and the read-only probe."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
lib = _lib.load()
flush = torch.empty(128 * 1024 * 1024, device='cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    tot = 0.0
    for _ in range(n):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1000
st = torch.cuda.current_stream().cuda_stream
for mb in (139, 71, 37, 230):
    n = mb * 1000 * 1000 // 4
    a, b = torch.randn(n, device='cuda'), torch.empty(n, device='cuda')
    t = timeit(lambda: lib.pgv_copy(a.data_ptr(), b.data_ptr(), n, st))
    t2 = timeit(lambda: b.copy_(a))
    sink = torch.zeros(1, device='cuda')
    t3 = timeit(lambda: lib.pgv_probe_read(a.data_ptr(), n, sink.data_ptr(), st))
    print(f'{2*mb} MB moved (read {mb} + write {mb}): pgv_copy {t:6.1f} us = {2*mb/t/1e3*1e3:.2f} GB/ms, torch copy_ {t2:6.1f} us = {2*mb/t2:.2f} TB/s... | read-only {mb} MB: {t3:6.1f} us = {mb/t3:.2f} TB/s')
