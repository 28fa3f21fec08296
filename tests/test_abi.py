"""CPU: the C-ABI shared library loads and exports every symbol include/pgv_hip.h declares (no compute calls)."""
import os
import re

import pytest

from helpers import ROOT


def _header_functions():
    text = open(os.path.join(ROOT, "include", "pgv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgv_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from preset_gen_vae_amd import _lib
    names = _header_functions()
    assert len(names) >= 30
    assert sorted(_lib.SIGNATURES.keys()) == names


def test_library_exports_every_symbol():
    from preset_gen_vae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpgv_hip.so not built (run python __graft_entry__.py build)")
    lib = _lib.load()
    for name in _header_functions():
        assert hasattr(lib, name), name
    assert lib.pgv_abi_version() == 11
    assert lib.pgv_set_kernel_policy(0) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from preset_gen_vae_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_argument_errors_do_not_need_a_gpu():
    """Validation happens before any launch: bad descriptors come back as PGV_E_INVALID with a message."""
    import ctypes
    from preset_gen_vae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpgv_hip.so not built")
    lib = _lib.load()
    d = _lib.ConvDesc(1, 2, 9, 9, 3, 7, 5, 4, 4, 2, 2)     # Hs/Ws inconsistent with the geometry
    rc = lib.pgv_conv_down(ctypes.byref(d), 16, None, None, 16, None, 0, 0.0, 16, None, None)
    assert rc == -1 and b"inconsistent" in lib.pgv_last_error()
    assert lib.pgv_stft_mel(16, 1, 1000, 512, 256, 4, 16, 1.0, None, None, None, 0, 1e-6, 1.0, 0.0, 16, None) == -1
    assert b"n_fft=1024" in lib.pgv_last_error()
