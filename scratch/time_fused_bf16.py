"""time_fused.py for one layer in bf16 operand mode and fp32, policies 0 and 3 (the 129x174 input gradient of enc2)"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from preset_gen_vae_amd import ops
for dt in ('fp32', 'bf16'):
    for pol in (0, 3):
        ops.set_compute_dtype(dt)
        print('==', dt, 'policy', pol, flush=True)
        sys.argv = ['time_fused.py', 'enc1<-enc2', '--policy', str(pol)]
        runpy.run_path(os.path.join(ROOT, 'scratch', 'time_fused.py'), run_name='__main__')
ops.set_compute_dtype('fp32')
