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
