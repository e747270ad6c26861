"""CPU: the C-ABI shared library builds for gfx950 without a GPU, loads, and exports every symbol include/gq_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "gq_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gq_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from guidedquant_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("gq_anyprec_gemv", "gq_anyprec_dequant", "gq_lutgemm_gemv", "gq_qtip_matvec", "gq_hadamard", "gq_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in _declared():
        assert hasattr(lib, name), name


def test_python_binding_list_matches_header():
    from guidedquant_amd import _lib
    assert set(_lib.EXPORTS) == set(_declared())


def test_version_and_error_string_without_gpu(lib):
    lib.gq_version.restype = ctypes.c_int
    assert lib.gq_version() >= 100
    lib.gq_last_error.restype = ctypes.c_char_p
    assert lib.gq_last_error() is not None


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from guidedquant_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.lib()


def test_qtip_split_planner_runs_without_a_gpu(lib):
    """gq_qtip_plan_ksplit is host logic (launches nothing; 256 compute units assumed when no device is present): the split it
    suggests is one gq_qtip_linear_in accepts -- 1..max, every K range at least 16 four-tile-block chunks when it splits"""
    import ctypes
    lib.gq_qtip_plan_ksplit.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.c_int]
    for Ms, K in (([4096], 4096), ([4096, 4096, 4096], 4096), ([11008, 11008], 4096), ([4096], 11008), ([8192], 28672), ([64], 128), ([32], 32)):
        for mx in (1, 2, 4):
            ks = lib.gq_qtip_plan_ksplit(len(Ms), (ctypes.c_uint32 * len(Ms))(*Ms), K, mx)
            assert 1 <= ks <= mx
            assert ks == 1 or (K // 32) // 4 // ks >= 16
    # one linear with 128 bands on 256 units: two K ranges; q / k / v (3 x 128 bands): whole bands
    assert lib.gq_qtip_plan_ksplit(1, (ctypes.c_uint32 * 1)(4096), 4096, 4) == 2
    assert lib.gq_qtip_plan_ksplit(3, (ctypes.c_uint32 * 3)(4096, 4096, 4096), 4096, 4) == 1
    # bad arguments fall back to 1
    assert lib.gq_qtip_plan_ksplit(0, None, 4096, 4) == 1
