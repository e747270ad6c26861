"""GPU: the decode-to-fp16 matrix-core GEMV (csrc/ap_gemv.hip::ap_gemv_dq_kernel, round 6; the fast mode's 4-bit kernel for the large
matrices, replaces anyprec.cu:372-542 for M = 1) -- exact fp16 x fp16 products, fp32 accumulation, one fp16 rounding: the fast-mode
envelope against the oracle's fp64 product (ap_helpers._check_fast: never farther from the reference-order result than the correctly
rounded exact result, + 2 ulp), and the fused prologues / epilogues against the reference's rounding points (inference/model.py:259-
266, 281-292, 311-313).  Every bit width the kernel is compiled for, not only the ones the default dispatch sends to it."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from ap_helpers import _check_fast, half_add, rmsnorm_ref, run_fused, silu_mul_ref  # noqa: E402

EPS = 1e-5


@pytest.fixture(autouse=True)
def _dq_everywhere():
    from guidedquant_amd import _lib
    L = _lib.lib()
    os.environ["GQ_DQ"] = "7"
    os.environ["GQ_DQ_MIN_MWEIGHTS"] = "0"
    L.gq_reset_env_cache()
    L.gq_set_ap_mode(0)
    yield
    os.environ.pop("GQ_DQ", None)
    os.environ.pop("GQ_DQ_MIN_MWEIGHTS", None)
    L.gq_reset_env_cache()
    L.gq_set_ap_mode(-1)


def _layer(N, K, bits, seed):
    from guidedquant_amd import pack
    rng = np.random.default_rng(seed)
    return rng, pack.random_planes(N, K, bits, seed=seed), np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)


def _rows(rng, N, n=48):
    return np.unique(np.concatenate([np.arange(0, min(24, N)), np.arange(max(0, N - 24), N), rng.integers(0, N, n)]))


# 8B wqkv / wo / w1w3 / w2, a 1B width, a 70B width, ragged row counts (the last 16-row group partly / almost empty), the 70B down
# projection's width (more staging items than staging threads: the re-read path of stage_x_dqv)
SHAPES = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (3072, 2048), (10240, 8192), (1000, 4096), (4097, 1024), (520, 28672)]


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", SHAPES)
def test_dq_kernel_plain_residual_rmsnorm_pairs(oracle, bits, N, K):
    rng, q, lut = _layer(N, K, bits, 11 * bits + N + K)
    x = rng.normal(0, 1, K)
    x[rng.choice(K, 4, replace=False)] *= 30.0   # massive channels: nothing special happens to them here (fp16 operands)
    x = x.astype(np.float16)
    rows = _rows(rng, N)
    got = run_fused(x, q, lut, bits)
    _check_fast(got, x, q, lut, bits, oracle, rows=rows)
    res = rng.normal(0, 1, N).astype(np.float16)
    got_r = run_fused(x, q, lut, bits, residual=res, flags=1)
    assert np.array_equal(got_r.view(np.uint16), half_add(res, got).view(np.uint16))
    nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
    got_n = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS)
    _check_fast(got_n, rmsnorm_ref(x, nw, EPS), q, lut, bits, oracle, rows=rows)
    if N % 2 == 0:
        pairs = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS, flags=4, out_elems=N // 2)
        assert np.array_equal(pairs.view(np.uint16), silu_mul_ref(got_n[0::2], got_n[1::2]).view(np.uint16))


@pytest.mark.parametrize("bits", [3, 4])
def test_dq_kernel_silu_prologue(oracle, bits):
    N, K = 4096, 14336
    rng, q, lut = _layer(N, K, bits, 17 + bits)
    gu = rng.normal(0, 1, 2 * K).astype(np.float16)
    got = run_fused(gu, q, lut, bits, flags=2)
    _check_fast(got, silu_mul_ref(gu[:K], gu[K:]), q, lut, bits, oracle, rows=_rows(rng, N))


def test_default_dispatch_sends_the_large_4_bit_matrices_here_and_nothing_at_2_bits(oracle):
    """the same launch with the kernel switched off (GQ_DQ=0) runs the plane kernel: other summation order, so the two fast-mode results
    differ in the last bits for a 4-bit 25 M-weight (and a 3-bit 16 M-weight) matrix under the default dispatch -- and are identical where the default does not use it"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    res = {}
    for bits, N, K in ((4, 6144, 4096), (2, 6144, 4096), (4, 2048, 4096), (3, 4096, 4096)):
        rng, q, lut = _layer(N, K, bits, 5 + bits + N)
        x = rng.normal(0, 1, K).astype(np.float16)
        for dq in (None, "0"):
            os.environ.pop("GQ_DQ_MIN_MWEIGHTS", None)
            if dq is None:
                os.environ.pop("GQ_DQ", None)
            else:
                os.environ["GQ_DQ"] = dq
            L.gq_reset_env_cache()
            res[(bits, N, dq)] = run_fused(x, q, lut, bits)
        if N * K >= 16e6:  # (the 8 M-weight 4-bit matrix runs the exact-order kernel under the default dispatch: another envelope)
            _check_fast(res[(bits, N, None)], x, q, lut, bits, oracle, rows=_rows(rng, N))
    assert not np.array_equal(res[(4, 6144, None)].view(np.uint16), res[(4, 6144, "0")].view(np.uint16))
    assert np.array_equal(res[(2, 6144, None)].view(np.uint16), res[(2, 6144, "0")].view(np.uint16))
    assert np.array_equal(res[(4, 2048, None)].view(np.uint16), res[(4, 2048, "0")].view(np.uint16))
    assert not np.array_equal(res[(3, 4096, None)].view(np.uint16), res[(3, 4096, "0")].view(np.uint16))  # (3-bit wo: here since round 6)
