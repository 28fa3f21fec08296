import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_products_baseline():
    """The library default is fp32 products as six bf16 instructions (ops.DEFAULT_FP32_PRODUCTS).  The suite covers BOTH
    forms explicitly: tests run on the native instruction unless they switch to 'bf16x6' themselves (and switch back in
    their ``finally``), so the session starts from 'native'.  test_default_fp32_products_* check the default itself."""
    sys.path.insert(0, ROOT)
    import preset_gen_vae_amd  # noqa: F401
    from preset_gen_vae_amd import ops
    assert ops.fp32_products() == ops.DEFAULT_FP32_PRODUCTS == 'bf16x6'
    ops.set_fp32_products('native')
    yield
    ops.set_fp32_products(None)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
