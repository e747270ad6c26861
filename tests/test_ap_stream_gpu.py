"""GPU: every instance of the stream kernel (csrc/ap_stream.hip) against the oracle -- also the ones the default dispatch does not
use (GQ_ST=3 sends every prologue and bit width it serves to it): plain / RMSNorm / SiLU*up prologues, residual and gate/up pair
epilogues, 2-4 bits, raw and summed parking, one and two image units per builder wave, the massive-channel extraction.  Same
fast-mode envelope as the plane kernels (tests/ap_helpers.py::_check_fast; semantics anyprec.cu:372-542, model.py:259-313)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from ap_helpers import _check_fast, _fast, check_nonhot_accuracy, half_add, rmsnorm_ref, run_fused, silu_mul_ref  # noqa: E402

EPS = 1e-5


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(autouse=True)
def _stream_everywhere():
    from guidedquant_amd import _lib
    _fast()
    os.environ["GQ_ST"] = "3"
    _lib.lib().gq_reset_env_cache()
    yield
    for k in ("GQ_ST", "GQ_PL_MIN_MWEIGHTS", "GQ_PL_MAX_BITS", "GQ_PL_LOCAL"):  # (_fast() lifts the dispatch thresholds: not for the tests that follow)
        os.environ.pop(k, None)
    _lib.lib().gq_reset_env_cache()
    _lib.lib().gq_set_ap_mode(-1)


def _layer(N, K, bits, seed):
    from guidedquant_amd import pack
    rng = np.random.default_rng(seed)
    return rng, pack.random_planes(N, K, bits, seed=seed), np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)


def _rows(rng, N, n=48):
    return np.unique(np.concatenate([np.arange(0, 24), np.arange(N - 24, N), rng.integers(0, N, n)]))


def _served(N, K, bits, pro):
    """does the stream kernel take this launch?  (ask the dispatcher's own entry: a plain launch with GQ_ST=0 differs in the last bits)"""
    return K % 2048 == 0


# (N, K): 8B wqkv / wo / w1w3 / w2, 1B widths, a 70B width (two image units per builder wave), a ragged row count
SHAPES = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (3072, 2048), (10240, 8192), (1000, 4096)]


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", SHAPES)
def test_stream_kernel_plain_and_residual(oracle, bits, N, K):
    # (K = 14336 at 3 / 4 bits: four image units per builder wave are compiled for 2 bits only -- ap_stream.hip::pick_stream_cfg declines,
    # NPU > 2, and the same call is served by the plane kernels of ap_plane.hip: the fallback is what is checked for those two cases)
    rng, q, lut = _layer(N, K, bits, 3 * bits + N + K)
    x = rng.normal(0, 1, K).astype(np.float16)
    rows = _rows(rng, N)
    got = run_fused(x, q, lut, bits)
    _check_fast(got, x, q, lut, bits, oracle, rows=rows)
    res = rng.normal(0, 1, N).astype(np.float16)
    got_r = run_fused(x, q, lut, bits, residual=res, flags=1)
    assert np.array_equal(got_r.view(np.uint16), half_add(res, got).view(np.uint16))  # fp16 add of the same sums (model.py:311-313)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(6144, 4096), (28672, 4096), (3072, 2048), (10240, 8192)])
def test_stream_kernel_rmsnorm_and_pairs(oracle, bits, N, K):
    rng, q, lut = _layer(N, K, bits, 5 * bits + N + K)
    x = rng.normal(0, 1, K)
    x[rng.choice(K, 4, replace=False)] *= 30.0
    x = x.astype(np.float16)
    nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
    xn = rmsnorm_ref(x, nw, EPS)
    got = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS)
    _check_fast(got, xn, q, lut, bits, oracle, rows=_rows(rng, N))
    # gate/up pair epilogue on the same sums: silu(y[2i]) * y[2i+1] with the reference's fp16 rounding points (model.py:259-266)
    pairs = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS, flags=4, out_elems=N // 2)
    assert np.array_equal(pairs.view(np.uint16), silu_mul_ref(got[0::2], got[1::2]).view(np.uint16))


@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 14336)])
def test_stream_kernel_silu_prologue(oracle, N, K):
    bits = 2
    rng, q, lut = _layer(N, K, bits, 17 + N + K)
    gu = rng.normal(0, 1, 2 * K).astype(np.float16)
    xs = silu_mul_ref(gu[:K], gu[K:])
    got = run_fused(gu, q, lut, bits, flags=2)
    _check_fast(got, xs, q, lut, bits, oracle, rows=_rows(rng, N))


@pytest.mark.parametrize("lr", [9, 13])
def test_stream_kernel_massive_channels(oracle, lr):
    """channels 2^9 / 2^13 times the rest, two of them in one image unit: the others keep fp32-class accuracy"""
    N, K, bits = 512, 4096, 2
    rng, q, lut = _layer(N, K, bits, 40 + lr)
    x = rng.normal(0, 1, K)
    hot = np.concatenate([rng.choice(K, 3, replace=False), [K - 1, K - 3, 17]])
    x[hot] = 2.0**lr * np.sign(x[hot])
    x = (x / 64).astype(np.float16)
    nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
    check_nonhot_accuracy(run_fused(x, q, lut, bits), x, hot, q, lut, bits, oracle)
    check_nonhot_accuracy(run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS), rmsnorm_ref(x, nw, EPS), hot, q, lut, bits, oracle)


# ----------------------------------------------------------------------------- rows wider than 16384: K split over blocks
@pytest.mark.parametrize("kslice", ["0", "2048"])  # default slice (4096 where it divides K) / 2048 (summed parking, 4 images per block)
@pytest.mark.parametrize("N,K", [(8192, 28672), (520, 28672), (1000, 20480), (4096, 18432)])
def test_k_split_over_blocks_with_a_workspace(oracle, monkeypatch, N, K, kslice):
    """gq_anyprec_gemv_fused_ws on the 70B down projection (8192 x 28672) and other rows wider than 16384: every block multiplies one
    K slice, the fp32 sums of the slices are added in a second launch and rounded ONCE (nround = 1; the two-launch form without a
    workspace rounds twice) -- anyprec.cu:430-436,505-512; the residual epilogue is the fp16 add of the same sums (model.py:311-313)."""
    from guidedquant_amd import _lib
    monkeypatch.setenv("GQ_ST_KSLICE", kslice)
    monkeypatch.delenv("GQ_ST", raising=False)  # (the default dispatch: the K split does not depend on GQ_ST)
    _lib.lib().gq_reset_env_cache()
    assert _lib.lib().gq_anyprec_gemv_fused_ws_bytes(N, K, 2, 1) == (K // (4096 if kslice == "0" and K % 4096 == 0 else 2048)) * N * 4
    assert _lib.lib().gq_anyprec_gemv_fused_ws_bytes(N, 14336, 2, 1) == 0 and _lib.lib().gq_anyprec_gemv_fused_ws_bytes(N, K, 2, 4) == 0
    rng, q, lut = _layer(N, K, 2, N + K)
    x = rng.normal(0, 1, K)
    x[rng.choice(K, 6, replace=False)] *= 40.0  # massive channels (the SiLU * up product is where Llama has them)
    x = x.astype(np.float16)
    rows = _rows(rng, N)
    got = run_fused(x, q, lut, 2, workspace=True)
    assert np.isfinite(got).all()
    _check_fast(got, x, q, lut, 2, oracle, rows=rows, nround=1.0)
    res = rng.normal(0, 1, N).astype(np.float16)
    got_r = run_fused(x, q, lut, 2, residual=res, flags=1, workspace=True)
    assert np.array_equal(got_r.view(np.uint16), half_add(res, got).view(np.uint16))
    _lib.lib().gq_reset_env_cache()


def test_k_split_hot_channels_at_the_70b_down_projection(oracle):
    """massive activation channels through the K split at (8192, 28672): the elements above the extraction threshold leave the image
    of their unit and are multiplied on their own -- the other elements keep fp32-class accuracy relative to sum|w||x| of the
    NON-hot elements (tests/ap_helpers.py::check_nonhot_accuracy), with ONE fp16 rounding."""
    from guidedquant_amd import _lib
    os.environ.pop("GQ_ST", None)
    _lib.lib().gq_reset_env_cache()
    N, K = 8192, 28672
    rng, q, lut = _layer(N, K, 2, 77)
    x = rng.normal(0, 0.05, K)
    hot = rng.choice(K, 5, replace=False)
    x[hot] = rng.choice([-1.0, 1.0], 5) * rng.uniform(40.0, 120.0, 5)
    x = x.astype(np.float16)
    rows = _rows(rng, N, n=24)
    got = run_fused(x, q, lut, 2, workspace=True)
    check_nonhot_accuracy(got[rows], x, hot, np.ascontiguousarray(q[:, rows, :]), lut[rows], 2, oracle, nround=1.0)
