"""GPU parity: the fused-dequant prefill GEMM (gq_anyprec_gemm, csrc/ap_gemm.hip) -- the seq_len > 1 branch of
APLinear.forward (inference/APLinear.py:35-50: anyprec_dequant + matmul) -- against the oracle's dequantised matrix times x
in float64.  fp32 accumulation, one fp16 rounding: |got - exact| <= 2^-11 |exact| + 1e-5 * sum|x||w| (tolerance stated here)."""
import os

import numpy as np
import pytest

from conftest import golden_files

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _gemm(x, q, lut, bits):
    from guidedquant_amd import ap_gemv
    d = torch.device("cuda:0")
    out = ap_gemv.anyprec_gemm(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float16)).to(d), torch.from_numpy(np.ascontiguousarray(q)).to(d),
                               torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float16)).to(d), bits)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _check(got, x, q, lut, bits, oracle, rows=None):
    if rows is not None:
        q, lut, got = np.ascontiguousarray(q[:, rows, :]), lut[rows], got[:, rows]
    W = oracle.ap_dequant(q, lut, bits).astype(np.float64)
    X = np.asarray(x, dtype=np.float16).astype(np.float64)
    ref = X @ W.T
    scale = np.abs(X) @ np.abs(W).T
    err = np.abs(got.astype(np.float64) - ref)
    assert np.isfinite(got.astype(np.float32)).all()
    assert (err <= 2.0**-11 * 1.001 * np.abs(ref) + 1e-5 * scale + 1e-7).all(), (err / (scale + 1e-30)).max()


@pytest.mark.parametrize("path", [p for p in golden_files("ap_b") if int(np.load(p)["bits"]) in (2, 3, 4) and int(np.load(p)["qweight"].shape[2]) % 2 == 0])
def test_gemm_goldens(oracle, path):
    """the reference-generated fixtures (pack.py planes, _dequantize_weight W): x = the fixture vector stacked with noise rows"""
    g = np.load(path)
    bits = int(g["bits"])
    K = g["qweight"].shape[2] * 32
    rng = np.random.default_rng(1)
    X = np.concatenate([g["x"].reshape(1, K), rng.normal(0, 1, (6, K)).astype(np.float16)])
    got = _gemm(X, g["qweight"], g["lut"], bits)
    ref = X.astype(np.float64) @ g["W"].astype(np.float64).T
    scale = np.abs(X.astype(np.float64)) @ np.abs(g["W"].astype(np.float64)).T
    assert (np.abs(got.astype(np.float64) - ref) <= 2.0**-11 * 1.001 * np.abs(ref) + 1e-5 * scale + 1e-7).all()


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K,S", [(256, 4096, 5), (128, 1024, 128), (200, 1088, 33), (130, 2048, 129), (64, 11008, 77), (36, 64, 3),
                                   (100, 1984, 260)])
def test_gemm_random(oracle, bits, N, K, S):
    """row / token / K tails: N not a multiple of 128 (or 4), S not a multiple of 128, tail chunks of 64 .. 960 weights"""
    rng = np.random.default_rng(bits * 977 + N + K + S)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = (rng.normal(0, 1, (N, 1 << bits)) * 10.0**rng.integers(-3, 1, (N, 1))).astype(np.float16)
    X = (rng.normal(0, 1, (S, K)) * np.where(rng.random((S, K)) < 0.02, 20.0, 1.0)).astype(np.float16)
    _check(_gemm(X, q, lut, bits), X, q, lut, bits, oracle)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K,S", [(6144, 4096, 512), (28672, 4096, 256), (4096, 14336, 384), (4096, 11008, 130)])
def test_gemm_full_size_sampled_rows(oracle, bits, N, K, S):
    from guidedquant_amd import pack
    rng = np.random.default_rng(bits + N + K)
    q = pack.random_planes(N, K, bits, seed=bits * 31 + N)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    X = rng.normal(0, 1, (S, K)).astype(np.float16)
    got = _gemm(X, q, lut, bits)
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(N - 40, N), rng.integers(0, N, 64)]))
    _check(got, X, q, lut, bits, oracle, rows=rows)


@pytest.fixture
def _shape_env():
    from guidedquant_amd import _lib
    yield
    os.environ.pop("GQ_GEMM_SHAPE", None)
    _lib.lib().gq_reset_env_cache()


@pytest.mark.parametrize("shape", [0, 14, 24, 18])
@pytest.mark.parametrize("bits", [2, 3, 4])
def test_gemm_every_tile_shape(oracle, _shape_env, bits, shape):
    """the dispatcher picks a wave / block tile per problem (csrc/ap_gemm.hip::launch_gemm); here every one of them is forced
    (GQ_GEMM_SHAPE) on problems with row, token and K tails (K % 256 == 0: the pipelined kernel; 4-bit 24 falls back to 14)"""
    from guidedquant_amd import _lib
    os.environ["GQ_GEMM_SHAPE"] = str(shape)
    _lib.lib().gq_reset_env_cache()
    for N, K, S in [(520, 2304, 300), (260, 4096, 257), (96, 768, 130)]:
        rng = np.random.default_rng(bits * 131 + shape + N)
        codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
        q = oracle.ap_pack(codes, bits)
        lut = (rng.normal(0, 1, (N, 1 << bits)) * 10.0**rng.integers(-3, 1, (N, 1))).astype(np.float16)
        X = (rng.normal(0, 1, (S, K)) * np.where(rng.random((S, K)) < 0.02, 20.0, 1.0)).astype(np.float16)
        _check(_gemm(X, q, lut, bits), X, q, lut, bits, oracle)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K,S", [(4096, 4096, 128), (4096, 14336, 100), (1000, 2048, 33), (6144, 4096, 512)])
def test_gemm_split_k_on_short_grids(oracle, bits, N, K, S):
    """fewer output tiles than compute units: the Python op hands gq_anyprec_gemm_ws a workspace and K is split over fp32
    partial sums (the planner must actually split here); same tolerance as the single pass, and within fp16 rounding of it"""
    from guidedquant_amd import _lib, pack
    assert _lib.lib().gq_anyprec_gemm_ws_bytes(S, N, K, bits) >= 2 * S * N * 4
    rng = np.random.default_rng(bits + N + K + S)
    q = pack.random_planes(N, K, bits, seed=bits * 17 + N)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    X = rng.normal(0, 1, (S, K)).astype(np.float16)
    got = _gemm(X, q, lut, bits)
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(N - 40, N), rng.integers(0, N, 64)]))
    _check(got, X, q, lut, bits, oracle, rows=rows)
    d = torch.device("cuda:0")
    xt, qt, lt = (torch.from_numpy(np.ascontiguousarray(a)).to(d) for a in (X, q, lut))
    one = torch.empty(S, N, dtype=torch.float16, device=d)
    _lib.check(_lib.lib().gq_anyprec_gemm(xt.data_ptr(), one.data_ptr(), qt.data_ptr(), lt.data_ptr(), S, N, K, bits, None), "gq_anyprec_gemm")
    torch.cuda.synchronize()
    diff = np.abs(got.astype(np.float32) - one.cpu().numpy().astype(np.float32))
    assert (diff <= 2.0**-10 * np.abs(got.astype(np.float32)) + 1e-4).all()


def test_gemm_is_deterministic_and_row_independent(oracle):
    """size-independent properties: replays are bit-identical; a token's output row does not depend on what else is in the batch"""
    from guidedquant_amd import pack
    bits, N, K = 2, 1024, 4096
    rng = np.random.default_rng(5)
    q = pack.random_planes(N, K, bits, seed=9)
    lut = np.sort(rng.normal(0, 0.02, (N, 4)).astype(np.float16), axis=1)
    X = rng.normal(0, 1, (200, K)).astype(np.float16)
    a, b = _gemm(X, q, lut, bits), _gemm(X, q, lut, bits)
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))
    c = _gemm(X[37:150], q, lut, bits)
    assert np.array_equal(a[37:150].view(np.uint16), c.view(np.uint16))


def test_modules_take_the_fused_gemm_for_prefill_rows(oracle):
    """APLinear / AnyPrecisionLinear with seq_len > 1: the fused GEMM and the reference's dequant + matmul branch agree"""
    from guidedquant_amd.APLinear import APLinear
    d = torch.device("cuda:0")
    bits, N, K = 3, 512, 2048
    rng = np.random.default_rng(2)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q, lut = oracle.ap_pack(codes, bits), rng.normal(0, 0.03, (N, 1 << bits)).astype(np.float16)
    lin = APLinear(K, N, bits, device=d)
    lin.load_state_dict({"qweight": torch.from_numpy(q), "lut": torch.from_numpy(lut)})
    xs = torch.from_numpy(rng.normal(0, 1, (1, 70, K)).astype(np.float16)).to(d)
    os.environ["GQ_PREFILL_FUSED"] = "1"  # (auto sends only large matrices with short prompts to the fused kernel)
    try:
        y = lin(xs)
    finally:
        del os.environ["GQ_PREFILL_FUSED"]
    assert tuple(y.shape) == (1, 70, N) and y.dtype == torch.float16
    _check(y.cpu().numpy()[0], xs.cpu().numpy()[0], q, lut, bits, oracle)
    os.environ["GQ_PREFILL_FUSED"] = "0"
    try:
        y0 = lin(xs)
    finally:
        del os.environ["GQ_PREFILL_FUSED"]
    assert float((y.float() - y0.float()).abs().max()) <= 2e-2 * float(y0.float().abs().max())


@pytest.mark.parametrize("path", golden_files("ap_b"))
def test_device_packer_goldens_bit_exact(oracle, path):
    """gq_anyprec_pack against the reference-generated planes (pack.py pack_single_weight goldens, bits 2..8, tail chunks)"""
    from guidedquant_amd import ap_gemv
    g = np.load(path)
    bits = int(g["bits"])
    codes = oracle.ap_unpack(g["qweight"], bits)
    q = ap_gemv.anyprec_pack(torch.from_numpy(codes).cuda(), bits)
    assert np.array_equal(q.cpu().numpy(), g["qweight"])
    # any-precision prefix property: packing the parent codes and keeping the first b planes == packing codes >> (bits - b)
    for b in range(1, bits):
        qb = ap_gemv.anyprec_pack(torch.from_numpy(codes >> (bits - b)).cuda(), b)
        assert torch.equal(qb, q[:b])


def test_device_packer_full_size_round_trip():
    """size-independent property at a BASELINE shape: pack on the device -> dequantise on the device == lut[codes]; and equals
    the host packer"""
    from guidedquant_amd import ap_gemv, pack
    d = torch.device("cuda:0")
    N, K, bits = 4096, 14336, 3
    g = torch.Generator(device=d)
    g.manual_seed(4)
    codes = torch.randint(0, 1 << bits, (N, K), dtype=torch.uint8, device=d, generator=g)
    lut = torch.randn(N, 1 << bits, device=d, generator=g).half()
    q = ap_gemv.anyprec_pack(codes, bits)
    W = ap_gemv.anyprec_dequant(q, lut, bits)
    assert torch.equal(W, torch.gather(lut, 1, codes.long()))
    rows = [0, 1, 777, N - 1]
    assert np.array_equal(q[:, rows].cpu().numpy(), pack.pack_codes(codes[rows].cpu().numpy(), bits))
