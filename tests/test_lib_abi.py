"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares, and its
argument validation fails loudly (no kernel is launched here)."""
import ctypes
import os

import pytest

from coskad_amd import _lib


def test_library_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.lib()
    assert lib.coskad_abi_version() == 1


def test_every_header_symbol_is_exported():
    lib = _lib.lib()
    syms = _lib.header_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_exported_symbols_are_declared():
    """no stray extern "C" entry point without a header declaration"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("coskad_")}
    assert exported == set(_lib.header_symbols())


def test_argument_errors_are_reported_without_a_gpu():
    null = ctypes.c_void_p(0)
    with pytest.raises(_lib.CoskadHipError, match="null pointer"):
        _lib.call("coskad_gcn_f32", null, null, null, null, _lib.i32(4), _lib.i32(12), _lib.i32(17), _lib.i32(0), null)
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    with pytest.raises(_lib.CoskadHipError, match="unsupported"):
        _lib.call("coskad_gcn_f32", p, p, p, p, _lib.i32(4), _lib.i32(11), _lib.i32(17), _lib.i32(0), null)
    with pytest.raises(_lib.CoskadHipError, match="latent"):
        _lib.call("coskad_btlnk_fwd_f32", p, p, p, null, p, _lib.i32(4), _lib.i32(816), _lib.i32(64), null)


def test_size_queries():
    lib = _lib.lib()
    lib.coskad_stat_floats.restype = ctypes.c_int
    assert lib.coskad_stat_floats(32, 64) == 2 * 32 + 2 * 64 * 32 + 4 * 64
    lib.coskad_head_slots.restype = ctypes.c_int
    assert lib.coskad_head_slots() == 19
    lib.coskad_train_stats_ws_bytes.restype = ctypes.c_size_t
    assert lib.coskad_train_stats_ws_bytes(32) > 512 * 2 * (32 * 32 + 32) * 4


def test_python_layer_fits_agrees_with_the_library():
    """`ST_GCNN_layer.is_wide` decides in Python (no native call while a model is built); same answer as coskad_layer_fits."""
    from coskad_amd import ops
    from coskad_amd.models.graph_layers.stsgcn import layer_fits
    for V in (14, 17, 18, 25):
        for Ci in (1, 2, 3, 4, 8, 16, 32, 48, 64, 65, 128):
            for Co in (2, 16, 32, 64, 65, 256):
                assert layer_fits(Ci, Co, 12, V) == ops.layer_fits(Ci, Co, 12, V), (Ci, Co, V)
