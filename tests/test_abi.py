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
    assert lib.pgv_abi_version() == 16
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


def test_weight_shadow_sizes_and_descriptor_layout_do_not_need_a_gpu():
    """ABI v11: ``pgv_conv_desc`` ends in the ``w_shadow`` pointer (twelve int32 then one pointer: 56 bytes), and
    ``pgv_conv_weight_shadow_bytes`` answers from the descriptor alone - two bf16 layouts (4 bytes per weight) for the
    layers that have bf16-native kernels in bf16 operand mode, 0 for every other layer."""
    import ctypes
    from preset_gen_vae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpgv_hip.so not built")
    lib = _lib.load()
    assert ctypes.sizeof(_lib.ConvDesc) == 56 and _lib.ConvDesc.w_shadow.offset == 48
    BF16 = 2   # PGV_COMPUTE_BF16

    def nbytes(Cb, Cs, k, s, p, Hb, Wb, flags=BF16):
        Hs, Ws = (Hb + 2 * p - k) // s + 1, (Wb + 2 * p - k) // s + 1
        d = _lib.ConvDesc(1, Cb, Hb, Wb, Cs, Hs, Ws, k, k, s, p, flags, None)
        return lib.pgv_conv_weight_shadow_bytes(ctypes.byref(d))

    for Cb, Cs, k, s, p, Hb, Wb in [(64, 128, 4, 2, 2, 17, 23), (128, 256, 4, 2, 2, 9, 12), (256, 512, 4, 2, 2, 5, 7),
                                    (512, 2048, 1, 1, 0, 3, 4), (32, 64, 4, 2, 2, 33, 45), (16, 32, 4, 2, 2, 65, 88),
                                    (8, 16, 4, 2, 2, 129, 174)]:
        assert nbytes(Cb, Cs, k, s, p, Hb, Wb) == 4 * Cb * Cs * k * k
    # the 1-channel layers keep their direct kernels; no layer has a shadow for the deep layers' shapes at other plane sizes
    for case in [(1, 8, 5, 2, 2, 257, 347), (64, 128, 4, 2, 2, 19, 23), (3, 5, 4, 2, 2, 10, 13), (8, 16, 4, 2, 2, 128, 174)]:
        assert nbytes(*case) == 0
    # PGV_COMPUTE_F32_SPLIT (fp32 products as six bf16 instructions): three bf16 planes per direction, 12 bytes per weight, for
    # every k4 s2 layer of the stacks from 129x174 down to 5x7 and the 1x1 layers; ignored together with PGV_COMPUTE_BF16
    SPLIT = 8
    for Cb, Cs, k, s, p, Hb, Wb in [(64, 128, 4, 2, 2, 17, 23), (128, 256, 4, 2, 2, 9, 12), (256, 512, 4, 2, 2, 5, 7),
                                    (512, 2048, 1, 1, 0, 3, 4), (32, 64, 4, 2, 2, 33, 45), (16, 32, 4, 2, 2, 65, 88),
                                    (8, 16, 4, 2, 2, 129, 174)]:
        assert nbytes(Cb, Cs, k, s, p, Hb, Wb, flags=SPLIT) == 12 * Cb * Cs * k * k
        assert nbytes(Cb, Cs, k, s, p, Hb, Wb, flags=SPLIT | BF16) == 4 * Cb * Cs * k * k
        assert nbytes(Cb, Cs, k, s, p, Hb, Wb, flags=0) == 0
    assert nbytes(1, 8, 5, 2, 2, 257, 347, flags=SPLIT) == 0 and nbytes(16, 32, 4, 2, 2, 66, 88, flags=SPLIT) == 0
    # writing a shadow for a layer that has none is an argument error, reported before any launch
    d = _lib.ConvDesc(1, 1, 257, 347, 8, 129, 174, 5, 5, 2, 2, BF16, None)
    assert lib.pgv_conv_weight_shadow(ctypes.byref(d), 16, 16, None) == -1
    assert b"no weight shadow" in lib.pgv_last_error()
