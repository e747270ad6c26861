"""GPU parity: the HIP Any-Precision GEMV / dequant, called through the C ABI (ctypes -> libgq_hip.so) behind the
reference's `ap_gemv` module surface, against the CPU oracle.  The bar is BIT-EXACT fp16 (the kernel reproduces
the reference's fp16 accumulation order), so the north-star tolerance (1e-3 rel-fp16) is met with zero error."""
import os

import numpy as np
import pytest

from ap_helpers import _check_fast, _fast
from conftest import golden_files

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

REL_TOL = 1e-3  # north_star tolerance ("within 1e-3 rel-fp16 of the CUDA reference")


@pytest.fixture(autouse=True)
def _exact_mode():
    """Everything in this file not marked otherwise runs the bit-exact mode of the GEMV."""
    from guidedquant_amd import _lib
    _lib.check(_lib.lib().gq_set_ap_mode(1), "gq_set_ap_mode")
    yield
    _lib.lib().gq_set_ap_mode(-1)
    for k in ("GQ_PL_MIN_MWEIGHTS", "GQ_PL_MAX_BITS", "GQ_PL_LOCAL"):
        os.environ.pop(k, None)
    _lib.lib().gq_reset_env_cache()


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _run_gemv(x, q, lut, bits, M=1):
    from guidedquant_amd import ap_gemv
    d = _dev()
    K = q.shape[2] * 32
    N = q.shape[1]
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float16).reshape(M, 1, K)).to(d)
    qt = torch.from_numpy(np.ascontiguousarray(q)).to(d)
    lt = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float16)).to(d)
    out = torch.full((M, 1, N), float("nan"), dtype=torch.float16, device=d)
    ap_gemv.anyprec_gemv(xt, out, qt, lt, bits)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(M, N)


@pytest.mark.parametrize("path", golden_files("ap_b"))
def test_gemv_goldens_bit_exact(oracle, path):
    g = np.load(path)
    bits = int(g["bits"])
    want = oracle.ap_gemv_f16(g["x"], g["qweight"], g["lut"], bits)
    got = _run_gemv(g["x"], g["qweight"], g["lut"], bits)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    # and it is inside the fp16 envelope of the reference-generated exact product
    scale = np.abs(g["W"].astype(np.float64)) @ np.abs(g["x"].astype(np.float64))
    assert (np.abs(got[0].astype(np.float64) - g["y64"]) <= 4e-3 * scale + 1e-6).all()


@pytest.mark.parametrize("path", golden_files("ap_b"))
def test_dequant_goldens_bit_exact(path):
    from guidedquant_amd import ap_gemv
    g = np.load(path)
    bits = int(g["bits"])
    d = _dev()
    W = ap_gemv.anyprec_dequant(torch.from_numpy(g["qweight"]).to(d), torch.from_numpy(g["lut"]).to(d), bits)
    assert W.dtype == torch.float16 and tuple(W.shape) == g["W"].shape
    assert np.array_equal(W.cpu().numpy().view(np.uint16), g["W"].view(np.uint16))


@pytest.mark.parametrize("bits", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("N,K", [(64, 4096), (36, 1024), (20, 1152), (12, 11008), (8, 14336), (16, 2048), (8, 96),
                                 (4, 8192), (4, 5120)])
def test_gemv_random_bit_exact(oracle, bits, N, K):
    rng = np.random.default_rng(bits * 7919 + N * 131 + K)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = (rng.normal(0, 1, (N, 1 << bits)) * 10.0**rng.integers(-5, 1, (N, 1))).astype(np.float16)
    x = (rng.normal(0, 1, K) * 10.0**rng.integers(-3, 2, K)).astype(np.float16)
    want = oracle.ap_gemv_f16(x, q, lut, bits)
    got = _run_gemv(x, q, lut, bits)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (3072, 2048),
                                 (10240, 8192), (8192, 28672)])
def test_gemv_full_size_sampled_rows(oracle, bits, N, K):
    """Full BASELINE shapes (Llama-3-8B, 3.2-1B, 3.3-70B slices): launch the real grid, check a sample of rows
    (first/last blocks + random) bit-for-bit against the oracle evaluated on just those rows."""
    from guidedquant_amd import pack
    rng = np.random.default_rng(bits + N + K)
    q = pack.random_planes(N, K, bits, seed=bits * 31 + N)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    x = rng.normal(0, 1, K).astype(np.float16)
    got = _run_gemv(x, q, lut, bits)[0]
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(N - 40, N), rng.integers(0, N, 64)]))
    want = oracle.ap_gemv_f16(x, np.ascontiguousarray(q[:, rows, :]), lut[rows], bits)[0]
    assert np.array_equal(got[rows].view(np.uint16), want.view(np.uint16))
    assert np.isfinite(got.astype(np.float32)).all()


@pytest.mark.parametrize("bits", [2, 4])
def test_fast_path_equals_generic_path_on_device(bits):
    """size-independent property: the wave-64 quad kernel and the 32-lane generic kernel agree everywhere."""
    import ctypes
    from guidedquant_amd import _lib, pack
    N, K = 4096, 4096
    rng = np.random.default_rng(3)
    q = pack.random_planes(N, K, bits, seed=5)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    x = rng.normal(0, 1, K).astype(np.float16)
    fast = _run_gemv(x, q, lut, bits)
    os.environ["GQ_AP_FORCE_GENERIC"] = "1"
    _lib.lib().gq_reset_env_cache()
    try:
        slow = _run_gemv(x, q, lut, bits)
    finally:
        del os.environ["GQ_AP_FORCE_GENERIC"]
        _lib.lib().gq_reset_env_cache()
    assert np.array_equal(fast.view(np.uint16), slow.view(np.uint16))


@pytest.mark.parametrize("M", [2, 5, 8])
def test_gemv_multi_batch(oracle, M):
    bits, N, K = 3, 64, 4096
    rng = np.random.default_rng(M)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = rng.normal(0, 0.05, (N, 1 << bits)).astype(np.float16)
    X = rng.normal(0, 1, (M, K)).astype(np.float16)
    want = oracle.ap_gemv_f16(X, q, lut, bits)
    got = _run_gemv(X, q, lut, bits, M=M)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_any_precision_parent_tensor(oracle):
    """A 4-bit parent tensor served at 2 and 3 bits with its own LUTs (first b planes, pack.py:103-107)."""
    N, K = 32, 4096
    rng = np.random.default_rng(11)
    codes = rng.integers(0, 16, (N, K), dtype=np.uint8)
    q4 = oracle.ap_pack(codes, 4)
    x = rng.normal(0, 1, K).astype(np.float16)
    for b in (2, 3, 4):
        lut = rng.normal(0, 0.05, (N, 1 << b)).astype(np.float16)
        want = oracle.ap_gemv_f16(x, oracle.ap_pack(codes >> (4 - b), b), lut, b)
        got = _run_gemv(x, q4, lut, b)
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_validation_errors():
    from guidedquant_amd import ap_gemv
    d = _dev()
    x = torch.zeros(1, 1, 128, dtype=torch.float16, device=d)
    out = torch.zeros(1, 1, 8, dtype=torch.float16, device=d)
    q = torch.zeros(2, 8, 4, dtype=torch.int32, device=d)
    lut = torch.zeros(8, 4, dtype=torch.float16, device=d)
    ap_gemv.anyprec_gemv(x, out, q, lut, 2)
    with pytest.raises(RuntimeError, match="Bitwidth must be between 2 and 8"):
        ap_gemv.anyprec_gemv(x, out, q, lut, 9)
    with pytest.raises(RuntimeError, match="lut tensor must be of shape"):
        ap_gemv.anyprec_gemv(x, out, q, lut, 3)
    with pytest.raises(RuntimeError, match="qweight tensor must be of type int"):
        ap_gemv.anyprec_gemv(x, out, q.float(), lut, 2)
    with pytest.raises(RuntimeError, match="Mismatched data types"):
        ap_gemv.anyprec_gemv(x.float(), out, q, lut, 2)
    with pytest.raises(RuntimeError, match="Only sequence length of 1"):
        ap_gemv.anyprec_gemv(torch.zeros(1, 2, 128, dtype=torch.float16, device=d), out, q, lut, 2)
    with pytest.raises(RuntimeError, match="must be on GPU"):
        ap_gemv.anyprec_gemv(x.cpu(), out.cpu(), q, lut, 2)
    with pytest.raises(RuntimeError, match="contiguous"):
        ap_gemv.anyprec_gemv(x, out, q.transpose(1, 2).contiguous().transpose(1, 2), lut, 2)


def test_aplinear_module_and_custom_op(oracle):
    from guidedquant_amd.APLinear import APLinear
    d = _dev()
    bits, N, K = 2, 256, 4096
    rng = np.random.default_rng(2)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = rng.normal(0, 0.03, (N, 1 << bits)).astype(np.float16)
    lin = APLinear(K, N, bits, device=d)
    assert lin.qweight.shape == (bits, N, K // 32) and lin.qweight.dtype == torch.int32
    assert lin.lut.shape == (N, 1 << bits) and lin.lut.dtype == torch.float16
    lin.load_state_dict({"qweight": torch.from_numpy(q), "lut": torch.from_numpy(lut)})
    x = rng.normal(0, 1, (1, 1, K)).astype(np.float16)
    y = lin(torch.from_numpy(x).to(d))
    assert y is lin.output and tuple(y.shape) == (1, 1, N)
    want = oracle.ap_gemv_f16(x.reshape(K), q, lut, bits)
    assert np.array_equal(y.cpu().numpy().reshape(1, N).view(np.uint16), want.view(np.uint16))
    # the op is visible under the reference's name
    torch.ops.plugin.anyprec_gemv(torch.from_numpy(x).to(d), lin.qweight, lin.lut, lin.output, bits)
    # prefill branch: dequant + matmul
    xs = torch.from_numpy(rng.normal(0, 1, (1, 5, K)).astype(np.float16)).to(d)
    ys = lin(xs)
    W = oracle.ap_dequant(q, lut, bits).astype(np.float32)
    ref = xs.float().cpu().numpy()[0] @ W.T
    np.testing.assert_allclose(ys.float().cpu().numpy()[0], ref, rtol=2e-2, atol=2e-2)


# ----------------------------------------------------------------------------- fast (plane-MFMA) mode
@pytest.mark.parametrize("path", [p for p in golden_files("ap_b") if int(np.load(p)["bits"]) in (2, 3, 4)])
@pytest.mark.parametrize("local", [1, 0])
def test_fast_mode_goldens(oracle, path, local):
    g = np.load(path)
    bits = int(g["bits"])
    _fast(local=local)
    got = _run_gemv(g["x"], g["qweight"], g["lut"], bits)[0]
    _check_fast(got, g["x"], g["qweight"], g["lut"], bits, oracle)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(64, 4096), (36, 1024), (20, 1280), (12, 11008), (8, 14336), (16, 2048), (4, 8192),
                                 (200, 256), (17, 5120)])
@pytest.mark.parametrize("local", [1, 0])
def test_fast_mode_random(oracle, bits, N, K, local):
    rng = np.random.default_rng(bits * 7919 + N * 131 + K + 1)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    x = (rng.normal(0, 1, K) * np.where(rng.random(K) < 0.02, 40.0, 1.0)).astype(np.float16)
    _fast(local=local)
    got = _run_gemv(x, q, lut, bits)[0]
    _check_fast(got, x, q, lut, bits, oracle)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (3072, 2048),
                                 (10240, 8192), (8192, 28672)])
@pytest.mark.parametrize("local", [1, 0])
def test_fast_mode_full_size_sampled_rows(oracle, bits, N, K, local):
    from guidedquant_amd import pack
    rng = np.random.default_rng(bits + N + K)
    q = pack.random_planes(N, K, bits, seed=bits * 31 + N)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    x = rng.normal(0, 1, K).astype(np.float16)
    _fast(local=local)
    got = _run_gemv(x, q, lut, bits)[0]
    assert np.isfinite(got.astype(np.float32)).all()
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(N - 40, N), rng.integers(0, N, 64)]))
    _check_fast(got, x, q, lut, bits, oracle, rows=rows)


@pytest.mark.parametrize("local", [1, 0])
def test_fast_mode_tiny_and_huge_activations(oracle, local):
    """block scaling of the activation pieces: vectors of very small / very large / mixed magnitude stay accurate"""
    bits, N, K = 2, 48, 4096
    rng = np.random.default_rng(9)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    _fast(local=local)
    for mag in (1e-4, 1.0, 3e2):
        x = (rng.normal(0, mag, K)).astype(np.float16)
        got = _run_gemv(x, q, lut, bits)[0]
        _check_fast(got, x, q, lut, bits, oracle)
    x = (rng.normal(0, 1, K) * 10.0**rng.integers(-4, 2, K)).astype(np.float16)
    got = _run_gemv(x, q, lut, bits)[0]
    _check_fast(got, x, q, lut, bits, oracle)
    x = np.zeros(K, dtype=np.float16)
    assert (_run_gemv(x, q, lut, bits)[0] == 0).all()


def test_default_dispatch_is_hybrid(oracle):
    """Default fast mode without lifted thresholds: a small matrix is served by the exact-order kernel (bit-identical
    to the reference order), the 8B gate/up matrix by the plane-MFMA kernel (fp32-class accuracy)."""
    from guidedquant_amd import pack
    rng = np.random.default_rng(21)
    _fast(force_plane=False)
    for N, K, plane in ((2048, 4096, False), (4096, 4096, True), (28672, 4096, True)):
        q = pack.random_planes(N, K, 2, seed=N)
        lut = np.sort(rng.normal(0, 0.02, (N, 4)).astype(np.float16), axis=1)
        x = rng.normal(0, 1, K).astype(np.float16)
        got = _run_gemv(x, q, lut, 2)[0]
        rows = np.unique(rng.integers(0, N, 96))
        if plane:
            _check_fast(got, x, q, lut, 2, oracle, rows=rows)
        else:
            want = oracle.ap_gemv_f16(x, np.ascontiguousarray(q[:, rows, :]), lut[rows], 2)[0]
            assert np.array_equal(got[rows].view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("bits,N,K", [(2, 28672, 4096), (2, 4096, 14336), (3, 4096, 8192), (4, 2048, 4096), (2, 6144, 4096), (2, 4096, 4096),
                                       (3, 4096, 14336), (2, 4096, 4352)])
def test_fast_mode_is_deterministic(bits, N, K):
    """size-independent property: the plane-MFMA kernel has no order-dependent reductions (fixed DPP trees, ordered LDS
    sums, no atomics on data) -- 100 launches on the same inputs give bit-identical outputs (also a race detector for
    the direct-to-LDS stream / LDS-counter synchronisation)"""
    from guidedquant_amd import ap_gemv, pack
    d = _dev()
    rng = np.random.default_rng(bits + N)
    q = torch.from_numpy(pack.random_planes(N, K, bits, seed=3)).to(d)
    lut = torch.from_numpy(np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)).to(d)
    x = torch.from_numpy(rng.normal(0, 1, (1, 1, K)).astype(np.float16)).to(d)
    _fast()
    outs = []
    for i in range(100):
        out = torch.full((1, 1, N), float("nan"), dtype=torch.float16, device=d)
        ap_gemv.anyprec_gemv(x, out, q, lut, bits)
        outs.append(out)
    torch.cuda.synchronize()
    ref = outs[0].view(torch.int16)
    assert bool(torch.isfinite(outs[0].float()).all())
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), ref)


@pytest.mark.parametrize("N,K", [(64, 4096), (32, 28672)])
def test_fast_mode_multi_batch(oracle, N, K):
    """M > 1 on the plane kernel (one block row per batch entry), including the two-launch K split"""
    bits, M = 2, 3
    rng = np.random.default_rng(N + K)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    X = rng.normal(0, 1, (M, K)).astype(np.float16)
    _fast()
    got = _run_gemv(X, q, lut, bits, M=M)
    for mm in range(M):
        _check_fast(got[mm], X[mm], q, lut, bits, oracle)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(4, 28672, 4096), (2, 6144, 4096), (3, 4096, 4096), (8, 512, 4096), (7, 1024, 2048), (2, 1024, 8192), (4, 256, 4352)])
def test_fast_mode_rows_share_one_pass(oracle, monkeypatch, bits, M, N, K):
    """M = 2 .. 8 batch rows, up to 4 per pass over the planes (the reference's multi_row kernel, anyprec.cu:381,425,494-506): one
    image per row in MFMA columns 4 mm .. 4 mm + 3.  The arithmetic of a row is that of the one-row launch: results bit-identical
    to GQ_PL_ONEPASS=0 (one block row per batch row, same image builders), row by row within the fast-mode envelope; one row carries massive channels
    (extraction list entries tagged with their row).  Shapes whose images do not fit (K = 8192 at 3 / 4 bits) fall back."""
    from guidedquant_amd import _lib, pack
    rng = np.random.default_rng(bits * 31 + M + N + K)
    q = pack.random_planes(N, K, bits, seed=M + K)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    X = rng.normal(0, 1, (M, K))
    hot = rng.choice(K, 3, replace=False)
    X[M - 1, hot] *= 2.0**11
    X[0, hot[0]] *= 2.0**9
    X = X.astype(np.float16)
    _fast()
    got = _run_gemv(X, q, lut, bits, M=M)
    monkeypatch.setenv("GQ_PL_ONEPASS", "0")
    # the same kernel and image builders for the reference run: the local-image variant (own scale per wave) and the late-wave image
    # helpers (sum(x) partials split differently) give last-bit differences
    monkeypatch.setenv("GQ_PL_HIMG", "0")
    _fast(local=0)
    ref = _run_gemv(X, q, lut, bits, M=M)
    monkeypatch.delenv("GQ_PL_HIMG")
    _fast()
    ref_dflt = _run_gemv(X, q, lut, bits, M=M)  # (what a shape whose images do not fit falls back to: the default one-row dispatch)
    monkeypatch.delenv("GQ_PL_ONEPASS")
    _lib.lib().gq_reset_env_cache()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)) or np.array_equal(got.view(np.uint16), ref_dflt.view(np.uint16))
    rows = np.unique(np.concatenate([np.arange(0, 24), np.arange(N - 24, N), rng.integers(0, N, 48)]))
    for mm in (0, M - 1):
        _check_fast(got[mm], X[mm], q, lut, bits, oracle, rows=rows)


@pytest.mark.parametrize("bits,N,K", [(2, 28672, 4096), (2, 4096, 14336), (3, 28672, 4096), (4, 28672, 4096)])
def test_fast_mode_within_north_star_tolerance_of_reference_order(oracle, bits, N, K):
    """north_star: "outputs within 1e-3 rel-fp16 of the CUDA reference".  Norm-wise, the plane-MFMA result is within
    REL_TOL of the reference-order (fp16-accumulated) result -- and that distance is the reference's own accumulation
    error: the exact product is just as far from it."""
    from guidedquant_amd import ap_gemv, pack
    d = _dev()
    rng = np.random.default_rng(bits + N)
    q = pack.random_planes(N, K, bits, seed=7)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    x = rng.normal(0, 1, K).astype(np.float16)
    rows = np.unique(rng.integers(0, N, 384))
    qs, ls = np.ascontiguousarray(q[:, rows, :]), lut[rows]
    ref = oracle.ap_gemv_f16(x, qs, ls, bits)[0].astype(np.float64)
    y64 = oracle.ap_gemv_f64(x, qs, ls, bits)[0]
    _fast()
    g = _run_gemv(x, q, lut, bits)[0][rows].astype(np.float64)
    rel = np.linalg.norm(g - ref) / np.linalg.norm(ref)
    assert rel <= REL_TOL, rel
    assert rel <= 1.1 * np.linalg.norm(y64 - ref) / np.linalg.norm(ref) + 1e-5


@pytest.mark.parametrize("pt", ["0", "2"])
@pytest.mark.parametrize("M", [1, 3])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (1000, 8192), (520, 11008), (64, 28672), (36, 4096), (300, 2048)])
def test_two_bit_exact_kernels_agree_with_the_oracle_on_both_decodes(oracle, N, K, M, pt):
    """exact mode, 2 bits: the v_perm byte-pool kernel (GQ_AP_PT=0) and the LDS pair-table kernel (GQ_AP_PT=2: every shape it serves --
    rows of 4096 weights or >= 8192; 2048 and ragged widths fall back to the first) both reproduce the reference's fp16 order bit for
    bit: rows x widths incl. a tail chunk (11008), the 70B down projection's width, several batch rows; plus the fused prologues /
    epilogues on the pair-table kernel against the same chain on the v_perm kernel."""
    from guidedquant_amd import _lib, pack
    from ap_helpers import run_fused
    L = _lib.lib()
    os.environ["GQ_AP_PT"] = pt
    L.gq_reset_env_cache()
    try:
        bits = 2
        rng = np.random.default_rng(N + K + M)
        q = pack.random_planes(N, K, bits, seed=N + K)
        lut = (rng.normal(0, 1, (N, 4)) * 10.0**rng.integers(-4, 1, (N, 1))).astype(np.float16)
        x = (rng.normal(0, 1, (M, K)) * 10.0**rng.integers(-2, 2, (M, K))).astype(np.float16)
        got = _run_gemv(x, q, lut, bits, M=M)
        rows = np.unique(np.concatenate([np.arange(0, min(N, 24)), np.arange(max(0, N - 24), N), rng.integers(0, N, 32)]))
        for m in range(M):
            want = oracle.ap_gemv_f16(x[m], np.ascontiguousarray(q[:, rows, :]), lut[rows], bits)[0]
            assert np.array_equal(got[m][rows].view(np.uint16), want.view(np.uint16)), (m, pt)
        if M == 1 and N % 2 == 0 and K in (4096, 8192, 14336):  # (widths whose row step is even: the pair epilogue's condition, both kernels)
            nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
            res = rng.normal(0, 1, N).astype(np.float16)
            outs = {}
            for p2 in ("0", pt):
                os.environ["GQ_AP_PT"] = p2
                L.gq_reset_env_cache()
                outs[p2] = (run_fused(x[0], q, lut, bits, norm_weight=nw, eps=1e-5), run_fused(x[0], q, lut, bits, norm_weight=nw, eps=1e-5, flags=4, out_elems=N // 2),
                            run_fused(x[0], q, lut, bits, residual=res, flags=1))
            for a, b in zip(outs["0"], outs[pt]):
                assert np.array_equal(a.view(np.uint16), b.view(np.uint16))
    finally:
        os.environ.pop("GQ_AP_PT", None)
        L.gq_reset_env_cache()
