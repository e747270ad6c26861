"""CPU: pin the oracle's Any-Precision restatement against golden vectors produced by the
reference's own pack.py / finetune_utils.py (tests/make_golden.py)."""
import numpy as np
import pytest

from conftest import golden_files

AP = golden_files("ap_b")


def test_have_goldens():
    assert len(AP) >= 10


def test_half_rounding_matches_numpy(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.normal(0, 1, 20000), rng.normal(0, 1e-5, 20000), rng.normal(0, 3e4, 5000),
        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 2.0**-24, 2.0**-25, 2.0**-25 * 1.0000001, 6.1e-5, 1e-8]),
        # exact ties on the fp16 grid
        (np.arange(1, 4000) + 0.5) * 2.0**-24, (2048 + np.arange(0, 2000) + 0.5) * 2.0**-10,
    ])
    for v in vals:
        got = L.gq_oracle_d2h(float(v))
        want = int(np.float64(v).astype(np.float16).view(np.uint16))
        assert got == want, (v, got, want)
    hs = rng.integers(0, 1 << 16, 20000, dtype=np.uint16)
    for h in hs:
        d = L.gq_oracle_h2d(int(h))
        w = float(np.uint16(h).view(np.float16))
        assert (np.isnan(d) and np.isnan(w)) or d == w


@pytest.mark.parametrize("path", AP)
def test_pack_matches_reference(oracle, path):
    g = np.load(path)
    bits = int(g["bits"])
    q = oracle.ap_pack(g["codes"], bits)
    assert q.dtype == np.int32 and q.shape == g["qweight"].shape
    assert np.array_equal(q, g["qweight"])
    # the independent numpy byte-route statement agrees too
    assert np.array_equal(oracle.ap_pack_np(g["codes"], bits), g["qweight"])


@pytest.mark.parametrize("path", AP)
def test_unpack_and_dequant_match_reference(oracle, path):
    g = np.load(path)
    bits = int(g["bits"])
    assert np.array_equal(oracle.ap_unpack(g["qweight"], bits), g["codes"])
    W = oracle.ap_dequant(g["qweight"], g["lut"], bits)
    assert np.array_equal(W.view(np.uint16), g["W"].view(np.uint16))


@pytest.mark.parametrize("path", AP)
def test_any_precision_prefix_property(oracle, path):
    """First b' planes of a b-bit tensor are the b'-bit codes (MSB-first planes, pack.py:103-107)."""
    g = np.load(path)
    bits = int(g["bits"])
    for b2 in range(1, bits + 1):
        assert np.array_equal(oracle.ap_unpack(g["qweight"], b2), g["codes"] >> (bits - b2))


@pytest.mark.parametrize("path", AP)
def test_gemv_f64_matches_reference_matmul(oracle, path):
    g = np.load(path)
    y = oracle.ap_gemv_f64(g["x"], g["qweight"], g["lut"], int(g["bits"]))
    np.testing.assert_allclose(y[0], g["y64"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("path", AP)
def test_gemv_f16_c_vs_numpy_and_error_envelope(oracle, path):
    g = np.load(path)
    bits = int(g["bits"])
    if bits < 2:
        pytest.skip("kernel supports 2..8")
    K = g["codes"].shape[1]
    y16 = oracle.ap_gemv_f16(g["x"], g["qweight"], g["lut"], bits)
    ksplit = K > 4096 and bits >= 7
    if not ksplit:
        ynp = oracle.ap_gemv_f16_np(g["x"], g["qweight"], g["lut"], bits)
        assert np.array_equal(y16.view(np.uint16), ynp.view(np.uint16))
    # fp16-accumulated result stays inside the fp16 error envelope of the exact value
    scale = np.abs(g["W"].astype(np.float64)) @ np.abs(g["x"].astype(np.float64))
    err = np.abs(y16[0].astype(np.float64) - g["y64"])
    assert (err <= 4e-3 * scale + 1e-6).all(), (err / scale).max()


def test_gemv_f16_multi_m_is_per_row(oracle):
    g = np.load(AP[0])
    bits = int(g["bits"])
    K = g["codes"].shape[1]
    rng = np.random.default_rng(5)
    X = rng.normal(0, 1, (3, K)).astype(np.float16)
    Y = oracle.ap_gemv_f16(X, g["qweight"], g["lut"], bits)
    for m in range(3):
        y1 = oracle.ap_gemv_f16(X[m:m + 1], g["qweight"], g["lut"], bits)
        assert np.array_equal(Y[m].view(np.uint16), y1[0].view(np.uint16))


@pytest.mark.parametrize("path", golden_files("ap_b"))
def test_native_float_gemv_against_goldens(oracle, path):
    """the native-float CPU baseline (oracle.ap_gemv_f32) is within fp32-accumulation distance of the reference-generated
    exact product on every fixture, and agrees with the oracle's own fp64 statement"""
    g = np.load(path)
    bits = int(g["bits"])
    got = oracle.ap_gemv_f32(g["x"], g["qweight"], g["lut"], bits)[0].astype(np.float64)
    scale = np.abs(g["W"].astype(np.float64)) @ np.abs(g["x"].astype(np.float64))
    assert (np.abs(got - g["y64"]) <= 2.0**-11 * 1.001 * np.abs(g["y64"]) + 1e-5 * scale + 1e-7).all()
    y64 = oracle.ap_gemv_f64(g["x"], g["qweight"], g["lut"], bits)[0]
    assert (np.abs(got - y64) <= 2.0**-11 * 1.001 * np.abs(y64) + 1e-5 * scale + 1e-7).all()
