"""A/B of conv_up launches with an alternative library build (env PGV_ALT_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
if os.environ.get('PGV_ALT_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'scratch', os.environ['PGV_ALT_LIB'])
sys.argv = [sys.argv[0]] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(ROOT, 'scratch', 'time_convs.py'), run_name='__main__')
