"""bench.py with an alternative library build (env PGV_ALT_LIB = file under scratch/), for same-box A/B runs."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import preset_gen_vae_amd  # noqa
from preset_gen_vae_amd import _lib
if os.environ.get('PGV_ALT_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'scratch', os.environ['PGV_ALT_LIB'])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
