import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
mode = sys.argv[1]
if mode == 'build_first':
    g.build()
import torch
print('cuda available', torch.cuda.is_available(), 'count', torch.cuda.device_count())
maps = open('/proc/self/maps').read()
libs = sorted({l.split()[-1] for l in maps.split('\n') if 'amdhip' in l or 'libhsa' in l or 'libpgv' in l})
print('\n'.join(libs))
from preset_gen_vae_amd import ops
x = torch.zeros(1024, device='cuda')
ops.fill(x, 3.0)
torch.cuda.synchronize()
print('fill ok', float(x.sum()))
geom = ops.ConvGeom(1, 8, 5, 2, 2, 257, 347)
big = torch.zeros(2, 1, 257, 347, device='cuda'); w = torch.zeros(8, 1, 5, 5, device='cuda')
try:
    ops.conv_down(geom, big, w, None, 0, 0.0)
    print('conv ok')
except Exception as e:
    print('ERR', e)
