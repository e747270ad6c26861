"""CPU: pin the oracle's QTIP decode and Hadamard restatements against golden vectors produced by the reference's own
decode_compressed / quantlut_sym / matmul_hadU (tests/make_golden.py)."""
import numpy as np
import pytest

from conftest import golden_files


@pytest.mark.parametrize("path", golden_files("qtip_R"))
def test_qtip_decode_matches_reference(oracle, path):
    g = np.load(path)
    R, m, k = int(g["R"]), int(g["m"]), int(g["k"])
    W = oracle.qtip_decode(g["compressed"], g["tlut"], m, k, R)
    assert np.array_equal(W.view(np.uint16), g["W"].view(np.uint16))
    y = oracle.qtip_matvec(g["compressed"], g["tlut"], g["x"], m, k, R)
    np.testing.assert_allclose(y, g["y64"], rtol=1e-12, atol=1e-12)


def test_quantlut_sym_matches_reference(oracle):
    g = np.load(golden_files("qtip_quantlut_sym")[0])
    got = oracle.quantlut_sym(g["tlut"])
    want = g["expanded"].astype(np.float16)
    assert np.array_equal(got.astype(np.float32), want.astype(np.float32))


@pytest.mark.parametrize("path", golden_files("had_n"))
def test_hadamard_matches_reference(oracle, path):
    g = np.load(path)
    hk = g["hadK"].astype(np.float32)
    Y = oracle.matmul_hadU(g["X"], hk if hk.size else None)
    Yt = oracle.matmul_hadU(g["X"], hk if hk.size else None, transpose=True)
    np.testing.assert_allclose(Y, g["Y"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(Yt, g["Yt"], rtol=2e-5, atol=2e-5)
