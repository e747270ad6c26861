"""GPU parity of the kernels the benchmarked decode step actually runs: `gq_anyprec_gemv_fused` in the DEFAULT (fast,
plane-MFMA) arithmetic mode with the RMSNorm prologue (wqkv, w1w3: `ap_plane_kernel<b, PRO_RMSNORM, *>`), the residual
epilogue (wo, w2: `ap_plane_local_kernel` / `ap_plane_kernel`), the SiLU*up prologue, the gate/up pair epilogue and the
two-launch K split (K = 28672), on the Llama-3.1-8B / 3.2-1B / 3.3-70B layer shapes at 2, 3 and 4 bits.

Oracle side: the element-wise op is restated in numpy with the reference's rounding points (inference/model.py:281-292
RMSNorm: fp32 norm -> fp16 -> fp16 multiply by the weight; :259-266 fp16 silu * up; :311-313 fp16 residual add), the
resulting fp16 vector goes through `oracle.ap_gemv_f64 / ap_gemv_f16` and the fast-mode envelope of `_check_fast` is
asserted (tests/ap_helpers.py): at least as close to the reference-order result as the correctly rounded exact product.
"""
import os

import numpy as np
import pytest

from ap_helpers import _check_fast, _fast, half_add, lnq_like_layer, rmsnorm_ref, run_fused, silu_mul_ref

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GQ_EPI_RESIDUAL, GQ_PRO_SILU_MUL, GQ_EPI_SILU_PAIRS = 1, 2, 4
EPS = 1e-5


@pytest.fixture(autouse=True)
def _restore_mode():
    yield
    from guidedquant_amd import _lib
    _lib.lib().gq_set_ap_mode(-1)
    for k in ("GQ_PL_MIN_MWEIGHTS", "GQ_PL_MAX_BITS", "GQ_PL_LOCAL"):
        os.environ.pop(k, None)
    _lib.lib().gq_reset_env_cache()


def _layer(N, K, bits, seed):
    from guidedquant_amd import pack
    rng = np.random.default_rng(seed)
    q = pack.random_planes(N, K, bits, seed=seed)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    return rng, q, lut


def _rows(rng, N, n=64):
    return np.unique(np.concatenate([np.arange(0, 40), np.arange(N - 40, N), rng.integers(0, N, n)]))


def _hidden(rng, K):
    """a hidden-state-like vector: unit-scale noise with a few large channels"""
    x = rng.normal(0, 1, K)
    x[rng.choice(K, 4, replace=False)] *= 30.0
    return x.astype(np.float16)


# wqkv / w1w3 of Llama-3.1-8B, 3.2-1B, 3.3-70B: the RMSNorm prologue always runs the shared-image plane kernel
@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(6144, 4096), (28672, 4096), (3072, 2048), (16384, 2048), (10240, 8192), (57344, 8192)])
def test_rmsnorm_prologue_fast_mode(oracle, bits, N, K):
    if N == 57344 and bits != 2:
        pytest.skip("70B gate/up at 3/4 bits: same kernel instance as 28672x4096, 2-bit covers the grid size")
    rng, q, lut = _layer(N, K, bits, bits * 101 + N + K)
    x = _hidden(rng, K)
    nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
    _fast()
    got = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS)
    assert np.isfinite(got.astype(np.float32)).all()
    xn = rmsnorm_ref(x, nw, EPS)
    _check_fast(got, xn, q, lut, bits, oracle, rows=_rows(rng, N))


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(6144, 4096), (28672, 4096)])
def test_rmsnorm_prologue_default_dispatch_equals_unfused_chain(oracle, bits, N, K):
    """default thresholds (what bench.py runs): the fused RMSNorm -> GEMV equals the plain GEMV of the same build fed
    with the torch-normalised vector, except where a last-bit difference of the fp32 statistic flips an fp16 rounding of
    the normalised vector (reduction order of the sum of squares); at most a handful of outputs may move, by <= 1 ulp"""
    rng, q, lut = _layer(N, K, bits, bits * 7 + N)
    x = _hidden(rng, K)
    nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
    _fast(force_plane=False)
    fused = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS)
    xt = torch.from_numpy(x).cuda().float()
    xn = ((xt * torch.rsqrt((xt * xt).mean() + EPS)).half() * torch.from_numpy(nw).cuda()).cpu().numpy()
    assert (xn.view(np.uint16) != rmsnorm_ref(x, nw, EPS).view(np.uint16)).sum() <= 4  # torch on the GPU vs the numpy restatement
    plain = run_fused(xn, q, lut, bits)
    rows = _rows(rng, N, 32)
    # which kernel family the default dispatch picks (ap_gemv.hip): behind the RMSNorm prologue the plane-MFMA kernel from 20 M weights
    # at every width (round 5); the plain launch from 20 M at 2 bits, 32 M at 3 / 4 bits -- below that the exact kernel
    # (round 6: at 4 bits every launch from 20 M weights runs the decode-to-fp16 matrix-core kernel, ap_gemv_dq_kernel -- a fast-mode kernel)
    # (and from 16 M weights at 3 and 4 bits -- 3 bits: below 32 M, where the plane kernel takes over: fast-mode kernels either way)
    fused_fast, plain_fast = N * K >= 20 * 1000000, N * K >= (20 if bits == 2 else 16) * 1000000
    if fused_fast == plain_fast:
        diff = fused.view(np.uint16) != plain.view(np.uint16)
        assert diff.mean() <= 0.02, diff.mean()
        ulp = np.abs(np.spacing(plain)).astype(np.float64)
        # (round 4: the RMSNorm launches of the 2-bit 8B shapes run the stream kernel, the plain launch the round-3 kernels: two fp32
        # summation orders, so an output that nearly cancels may move by more than its own ulps -- by fp32 noise of sum|w||x|)
        slack = 2e-6 * (np.abs(lut.astype(np.float64)).max(axis=1) * np.abs(xn.astype(np.float64)).sum())
        assert (np.abs(fused.astype(np.float64) - plain.astype(np.float64)) <= 2 * ulp + slack).all()
    if fused_fast:
        _check_fast(fused, rmsnorm_ref(x, nw, EPS), q, lut, bits, oracle, rows=rows)
    if not plain_fast:  # the exact kernel: the reference's fp16 order on the normalised vector, bit for bit
        want = oracle.ap_gemv_f16(xn, np.ascontiguousarray(q[:, rows, :]), lut[rows], bits)[0]
        assert np.array_equal(plain[rows].view(np.uint16), want.view(np.uint16))
        if not fused_fast:
            assert (fused[rows].view(np.uint16) != want.view(np.uint16)).mean() <= 0.1


# wo / w2: residual epilogue, on the local-image kernel (2/3-bit default) and on the shared-image kernel
@pytest.mark.parametrize("local", [1, 0])
@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 14336), (2048, 2048), (2048, 8192), (8192, 8192)])
def test_residual_epilogue_fast_mode(oracle, bits, N, K, local):
    rng, q, lut = _layer(N, K, bits, bits * 13 + N + K)
    x = rng.normal(0, 1, K).astype(np.float16)
    resid = _hidden(rng, N)
    _fast(local=local)
    plain = run_fused(x, q, lut, bits)
    got = run_fused(x, q, lut, bits, residual=resid, flags=GQ_EPI_RESIDUAL)
    # the epilogue is one fp16 add behind the fp16-rounded GEMV result (model.py:311-313): bit-identical to the two ops
    assert np.array_equal(got.view(np.uint16), half_add(resid, plain).view(np.uint16))
    rows = _rows(rng, N)
    _check_fast(plain, x, q, lut, bits, oracle, rows=rows)
    # and against the oracle directly: |got - (resid + exact)| <= rounding of y + rounding of the sum (+ fp32-class slack)
    y64 = oracle.ap_gemv_f64(x, np.ascontiguousarray(q[:, rows, :]), lut[rows], bits)[0]
    e = resid[rows].astype(np.float64) + y64
    W = np.abs(oracle.ap_dequant(np.ascontiguousarray(q[:, rows, :]), lut[rows], bits).astype(np.float64))
    scale = W @ np.abs(x.astype(np.float64))
    assert (np.abs(got[rows].astype(np.float64) - e) <= 2.0**-11 * 1.002 * (np.abs(y64) + np.abs(e)) + 1e-5 * scale + 1e-7).all()


@pytest.mark.parametrize("bits", [2, 3])
def test_residual_epilogue_two_launch_k_split(oracle, bits):
    """K = 28672 (the 70B down projection): two launches over K-halves, the second adding to the first one's fp16 result
    through the residual epilogue (resid == out), here with an external residual on top"""
    N, K = 8192, 28672
    rng, q, lut = _layer(N, K, bits, bits + 5)
    x = rng.normal(0, 1, K).astype(np.float16)
    resid = _hidden(rng, N)
    _fast()
    got = run_fused(x, q, lut, bits, residual=resid, flags=GQ_EPI_RESIDUAL)
    plain = run_fused(x, q, lut, bits)
    rows = _rows(rng, N, 24)
    _check_fast(plain, x, q, lut, bits, oracle, rows=rows)
    k1 = ((K // 2 + 1023) // 1024) * 1024
    qs, ls = np.ascontiguousarray(q[:, rows, :]), lut[rows]
    y1 = oracle.ap_gemv_f64(x[:k1], np.ascontiguousarray(qs[:, :, :k1 // 32]), ls, bits)[0]
    y2 = oracle.ap_gemv_f64(x[k1:], np.ascontiguousarray(qs[:, :, k1 // 32:]), ls, bits)[0]
    r = resid[rows].astype(np.float64)
    W = np.abs(oracle.ap_dequant(qs, ls, bits).astype(np.float64))
    scale = W @ np.abs(x.astype(np.float64))
    # out = fp16(fp16(resid + fp16(y1)) + fp16(y2)): four roundings, each <= 2^-11 of the value rounded
    tol = 2.0**-11 * 1.002 * (np.abs(y1) + np.abs(y2) + np.abs(r + y1) + np.abs(r + y1 + y2)) + 1e-5 * scale + 1e-7
    assert (np.abs(got[rows].astype(np.float64) - (r + y1 + y2)) <= tol).all()


@pytest.mark.parametrize("local", [1, 0])
@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(4096, 14336), (2048, 8192)])
def test_silu_mul_prologue_fast_mode(oracle, bits, N, K, local):
    """w2 of the unpaired path (GQ_NATIVE_PAIRS=0): x holds [gate | up], the GEMV input is silu(gate) * up, + residual"""
    rng, q, lut = _layer(N, K, bits, bits * 17 + N + K)
    gu = rng.normal(0, 1.5, 2 * K).astype(np.float16)
    resid = _hidden(rng, N)
    _fast(local=local)
    got = run_fused(gu, q, lut, bits, flags=GQ_PRO_SILU_MUL)
    h = silu_mul_ref(gu[:K], gu[K:])
    rows = _rows(rng, N)
    _check_fast(got, h, q, lut, bits, oracle, rows=rows)
    both = run_fused(gu, q, lut, bits, residual=resid, flags=GQ_PRO_SILU_MUL | GQ_EPI_RESIDUAL)
    assert np.array_equal(both.view(np.uint16), half_add(resid, got).view(np.uint16))


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(28672, 4096), (16384, 2048)])
def test_rmsnorm_plus_pair_epilogue_fast_mode(oracle, bits, N, K):
    """exactly the w1w3 launch of the decode step: RMSNorm prologue + gate/up pair epilogue on the row-interleaved tensor"""
    rng, q, lut = _layer(N, K, bits, bits * 19 + N)
    x = _hidden(rng, K)
    nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
    half = N // 2
    perm = np.stack((np.arange(half), np.arange(half, N)), axis=1).reshape(-1)
    qp, lp = np.ascontiguousarray(q[:, perm, :]), np.ascontiguousarray(lut[perm])
    _fast()
    y = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS)
    o = run_fused(x, qp, lp, bits, norm_weight=nw, eps=EPS, flags=GQ_EPI_SILU_PAIRS, out_elems=half)
    want = silu_mul_ref(y[:half], y[half:])
    diff = np.abs(o.astype(np.float64) - want.astype(np.float64))
    assert (diff <= 2.0**-10 * np.abs(want.astype(np.float64)) + 1e-7).all()  # the only freedom: the last bit of exp()
    assert (o.view(np.uint16) != want.view(np.uint16)).mean() < 0.02
    _check_fast(y, rmsnorm_ref(x, nw, EPS), q, lut, bits, oracle, rows=_rows(rng, N, 32))


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K,kind", [(6144, 4096, "norm"), (28672, 4096, "norm"), (4096, 4096, "resid"), (4096, 14336, "resid")])
def test_lnq_like_layers_fast_mode(oracle, bits, N, K, kind):
    """not uniform noise: skewed code histogram, outlier centroids in 1 % of the rows, heavy-tailed activations with
    massive channels (through RMSNorm for the wqkv / w1w3 shapes) -- envelope AND the north-star norm-wise figure"""
    q, lut, x = lnq_like_layer(N, K, bits, seed=bits * 1000 + N + K)
    rng = np.random.default_rng(N + bits)
    _fast()
    if kind == "norm":
        nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
        got = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS)
        xin = rmsnorm_ref(x, nw, EPS)
    else:
        xin = (x.astype(np.float32) / 8).astype(np.float16)
        got = run_fused(xin, q, lut, bits)
    assert np.isfinite(got.astype(np.float32)).all()
    # rows: the outlier-centroid rows must be in the sample
    big = np.argsort(-np.abs(lut.astype(np.float32)).max(axis=1))[:24]
    rows = np.unique(np.concatenate([big, _rows(rng, N, 96)]))
    _check_fast(got, xin, q, lut, bits, oracle, rows=rows)
    qs, ls = np.ascontiguousarray(q[:, rows, :]), lut[rows]
    ref = oracle.ap_gemv_f16(xin, qs, ls, bits)[0].astype(np.float64)
    y64 = oracle.ap_gemv_f64(xin, qs, ls, bits)[0]
    rel = np.linalg.norm(got[rows].astype(np.float64) - ref) / np.linalg.norm(ref)
    d_ref = np.linalg.norm(y64 - ref) / np.linalg.norm(ref)  # the reference-order kernel's own distance from the exact product
    # north star: within 1e-3 (relative, norm-wise) of the reference-order result -- unless that result is itself about that far
    # from the exact product (heavy-tailed inputs: d_ref 0.9e-3 .. 1.2e-3 here); never more than the reference's own distance and
    # the fp16 rounding of the output (asserted <= 4e-4 below) combined
    assert rel <= max(1e-3, d_ref + 4e-4)  # (triangle inequality)
    # against the exact product: the fp16 output rounding only (round 2: up to 7.5e-4 from the alignment loss next to the
    # massive channels, before they were taken out of the MFMA image); the reference-order kernel's own figure is d_ref
    assert np.linalg.norm(got[rows].astype(np.float64) - y64) / np.linalg.norm(y64) <= 4e-4


@pytest.mark.parametrize("local", [1, 0])
@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("nhot", [1, 6, 16])
def test_hot_channels_plain(oracle, bits, nhot, local):
    """1, 6, 16 channels 2^8 .. 2^14 times the rest (the probe of tools/plane_dynrange_probe.py as a test): the error in excess
    of the fp16 output rounding stays <= 2e-5 of sum|w||x| over the OTHER elements, on both plane kernels.  (The threshold is 64 x
    the mean magnitude INCLUDING the hot channels: a channel is extracted once it holds 1 / 64 of the vector's -- in the local-image
    kernel of its wave's 1024-element chunk's -- total magnitude.)"""
    from ap_helpers import check_nonhot_accuracy
    N, K = 256, 4096
    rng, q, lut = _layer(N, K, bits, 77 + bits)
    _fast(local=local)
    for lr in (8, 10, 12, 14):
        x = rng.normal(0, 1, K)
        hot = rng.choice(K, nhot, replace=False)
        if nhot == 6:
            hot[1] = hot[0] ^ 8   # two extracted elements in the same 128-element MFMA group
        x[hot] = 2.0**lr * np.sign(x[hot]) * rng.uniform(0.75, 1.0, nhot)
        x = x.astype(np.float16)
        got = run_fused(x, q, lut, bits)
        check_nonhot_accuracy(got, x, hot, q, lut, bits, oracle)
        _check_fast(got, x, q, lut, bits, oracle)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(6144, 4096), (1024, 4352), (10240, 8192), (512, 14336), (512, 28672)])
def test_hot_channels_rmsnorm_and_tails(oracle, bits, N, K):
    """the same through the RMSNorm prologue (shared-image kernel: the scanner waves take the threshold from the products x w
    before the normalisation), with a 256-weight tail chunk (K = 4352: the virtual-lane geometry of a short chunk), at the 70B
    width (8192), and -- plain prologue -- with the late-wave image helpers / one chunk per wave (14336).  K = 14336 is not a
    model width: behind RMSNorm its staged copy does not fit next to the rings and nothing is extracted (ap_plane.hip `detect`).
    K = 28672 (the 70B down projection, plain prologue: its input is the silu * up vector the w1w3 epilogue wrote, where real
    Llama has its massive channels) runs as two launches over K halves chained through the residual epilogue."""
    from ap_helpers import check_nonhot_accuracy
    rng, q, lut = _layer(N, K, bits, 5 * bits + K)
    rows = _rows(rng, N)
    qs, ls = np.ascontiguousarray(q[:, rows, :]), lut[rows]
    _fast()
    for lr in (9, 13):
        x = rng.normal(0, 1, K)
        hot = np.concatenate([rng.choice(K, 3, replace=False), [K - 1, K - 250]])
        x[hot] = 2.0**lr * np.sign(x[hot])
        x = (x / 64).astype(np.float16)
        if K not in (14336, 28672):
            nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
            got = run_fused(x, q, lut, bits, norm_weight=nw, eps=EPS)[rows]
            xn = rmsnorm_ref(x, nw, EPS)
            check_nonhot_accuracy(got, xn, hot, qs, ls, bits, oracle)
        got = run_fused(x, q, lut, bits)[rows]
        if K > 16384:
            # two fp16 roundings, the first of a partial sum that may be larger than the result: bound it by the hot part's size
            W = np.abs(oracle.ap_dequant(qs, ls, bits).astype(np.float64))
            y64 = oracle.ap_gemv_f64(x, qs, ls, bits)[0]
            part = W @ np.abs(x.astype(np.float64))
            over = np.abs(got.astype(np.float64) - y64) - 2.0**-11 * (np.abs(y64) + part) * 1.001
            xs = np.abs(x.astype(np.float64)); xs[hot] = 0
            assert (over <= 2e-5 * (W @ xs) + 1e-9).all()
            continue
        check_nonhot_accuracy(got, x, hot, qs, ls, bits, oracle)


@pytest.mark.parametrize("mode", ["default", "exact"])
def test_anyprecision_linear_forward_multi_precision(oracle, mode):
    """AnyPrecisionLinear.forward (any_precision/modules/AnyPrecisionLinear.py:63-80): a 4-bit parent tensor served at
    2 / 3 / 4 bits with the per-precision LUTs, decode row (GEMV into the persistent output, clamp) and prefill rows
    (dequant + matmul), bias on and off"""
    from guidedquant_amd import _lib
    from guidedquant_amd.AnyPrecisionLinear import AnyPrecisionLinear
    d = torch.device("cuda:0")
    N, K = 512, 4096
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 16, (N, K), dtype=np.uint8)
    q4 = oracle.ap_pack(codes, 4)
    luts = {b: np.sort(rng.normal(0, 0.03, (N, 1 << b)).astype(np.float16), axis=1) for b in (2, 3, 4)}
    bias = rng.normal(0, 0.1, N).astype(np.float16)
    _lib.check(_lib.lib().gq_set_ap_mode(1 if mode == "exact" else 0), "mode")
    if mode == "default":
        os.environ["GQ_PL_MIN_MWEIGHTS"] = "0"
        _lib.lib().gq_reset_env_cache()
    x = rng.normal(0, 1, (1, 1, K)).astype(np.float16)
    nobias = {}
    for use_bias in (False, True):
        lin = AnyPrecisionLinear(K, N, [2, 3, 4], bias=use_bias, device=d, dtype=torch.float16)
        sd = {"qweight": torch.from_numpy(q4)}
        sd.update({f"lut{b}": torch.from_numpy(luts[b]) for b in (2, 3, 4)})
        if use_bias:
            sd["bias"] = torch.from_numpy(bias)
        lin.load_state_dict(sd)
        assert lin.precision == 4
        for b in (2, 3, 4):
            y = lin(torch.from_numpy(x).to(d), precision=b)
            assert tuple(y.shape) == (1, 1, N) and y.dtype == torch.float16
            qb = oracle.ap_pack(codes >> (4 - b), b)
            got = y.cpu().numpy().reshape(N).copy()
            if use_bias:  # x += bias in fp16, then the clamp (a no-op at these magnitudes)
                assert np.array_equal(got.view(np.uint16), half_add(nobias[b], bias).view(np.uint16))
                continue
            nobias[b] = got
            if mode == "exact":
                want = oracle.ap_gemv_f16(x.reshape(K), qb, luts[b], b)[0]
                assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
            else:
                _check_fast(got, x.reshape(K), qb, luts[b], b, oracle)
        lin.set_precision(3)
        y3 = lin(torch.from_numpy(x).to(d)).cpu().numpy().reshape(N)
        y3b = lin(torch.from_numpy(x).to(d), precision=3).cpu().numpy().reshape(N)
        assert np.array_equal(y3.view(np.uint16), y3b.view(np.uint16))
        with pytest.raises(RuntimeError):
            lin.set_precision(5)
        # prefill rows: dequant + matmul
        xs = torch.from_numpy(rng.normal(0, 1, (1, 5, K)).astype(np.float16)).to(d)
        ys = lin(xs, precision=2)
        W = oracle.ap_dequant(oracle.ap_pack(codes >> 2, 2), luts[2], 2).astype(np.float32)
        ref = xs.float().cpu().numpy()[0] @ W.T + (bias.astype(np.float32) if use_bias else 0.0)
        np.testing.assert_allclose(ys.float().cpu().numpy()[0], ref, rtol=2e-2, atol=2e-2)
        if mode != "exact":
            continue
        # the clamp: finfo.max * (1 - 5e-3)
        lin2 = AnyPrecisionLinear(K, N, [2], bias=False, device=d, dtype=torch.float16)
        lin2.load_state_dict({"qweight": torch.from_numpy(oracle.ap_pack(np.zeros((N, K), dtype=np.uint8), 2)),
                              "lut2": torch.full((N, 4), 60000.0, dtype=torch.float16)})
        big = lin2(torch.full((1, 1, K), 100.0, dtype=torch.float16, device=d))
        assert float(big.float().abs().max()) == float(torch.tensor(torch.finfo(torch.float16).max * (1.0 - 5e-3)).half())


def test_anyprecision_for_causal_lm_on_gpu(tmp_path):
    """the HF path on the GPU (inference_example.py:34-77 harness shape): from_quantized -> HF generate with the rows == 1
    calls on the HIP LUT-GEMV; logits against the dense HF model holding the dequantised weights, at 2 and 3 bits"""
    transformers = pytest.importorskip("transformers")
    from ap_helpers import tiny_hf_anyprec_checkpoint
    from guidedquant_amd import ap_gemv
    from guidedquant_amd.AnyPrecisionForCausalLM import AnyPrecisionForCausalLM
    hf_cfg, sd, names, (D, I, H, KV, Lr, V) = tiny_hf_anyprec_checkpoint(tmp_path, D=512, I=1024, H=8, KV=2, V=256)
    d = torch.device("cuda:0")
    m = AnyPrecisionForCausalLM.from_quantized(str(tmp_path))
    assert m.device.type == "cuda" and m.ap_linears[0].qweight.is_cuda and m.ap_linears[0].output.is_cuda
    ids = torch.tensor([[3, 17, 5, 60, 2]], device=d)
    for b in (2, 3):
        dense = transformers.LlamaForCausalLM(hf_cfg).half()
        dsd = {k: v for k, v in sd.items() if not (k.endswith(".qweight") or ".lut" in k)}
        for i in range(Lr):
            for name in names:
                p = f"model.layers.{i}.{name}"
                dsd[p + ".weight"] = ap_gemv.anyprec_dequant(sd[p + ".qweight"], sd[p + f".lut{b}"], b)
        dense.load_state_dict(dsd, strict=True)
        dense = dense.to(d)
        with torch.no_grad():
            want = dense(ids).logits.float()
            got = m(ids, precision=b).logits.float()
            got1 = m(ids[:, :1], precision=b).logits.float()
        assert float((got - want).abs().max()) <= 2e-2 * float(want.abs().max())
        assert float((got1[0, 0] - want[0, 0]).abs().max()) <= 2e-2 * float(want.abs().max())
    out = m.generate(ids[:, :2], max_new_tokens=8, do_sample=False, precision=2)
    ref = transformers.LlamaForCausalLM(hf_cfg)  # noqa: F841  (shape only)
    assert out.shape == (1, 10) and m.precision == 3
