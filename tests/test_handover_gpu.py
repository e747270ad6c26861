"""GPU: the statistics hand-over between the launches of a decode step (include/gq_hip.h, GQ_SSQ_SLOTS; round 5; the decode step uses it
only with GQ_SSQ_HANDOVER=1 -- it measured slower end to end, profiles/r05_handover.txt -- the entry points are served either way).  The producer's
residual epilogue leaves the partial sums of squares of the fp16 hidden state it stores, the consumer's RMSNorm prologue adds them
instead of exchanging per-wave sums -- the reference's rounding points (inference/model.py:281-292) are unchanged, only the fp32
summation order of the mean of squares differs (as it already did from torch's)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from ap_helpers import _check_fast, half_add, rmsnorm_ref, silu_mul_ref  # noqa: E402

EPS = 1e-5


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(autouse=True)
def _default_dispatch():
    from guidedquant_amd import _lib
    for k in ("GQ_ST", "GQ_PL_MIN_MWEIGHTS", "GQ_PL_MAX_BITS", "GQ_PL_LOCAL", "GQ_SSQ_HANDOVER"):
        os.environ.pop(k, None)
    _lib.lib().gq_reset_env_cache()
    _lib.lib().gq_set_ap_mode(0)
    yield
    _lib.lib().gq_set_ap_mode(-1)


def _layer(N, K, bits, seed):
    from guidedquant_amd import pack
    rng = np.random.default_rng(seed)
    return rng, pack.random_planes(N, K, bits, seed=seed), np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)


def _dev(a, dt):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to("cuda:0")


def run_ho(x, q, lut, bits, norm_weight=None, residual=None, flags=0, ssq_in=None, want_ssq=False, out_elems=None):
    """gq_anyprec_gemv_fused_ho through the C ABI; returns (out, ssq slots or None)"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    N, K = q.shape[1], q.shape[2] * 32
    xt, qt, lt, nw, rs = _dev(x, np.float16), _dev(q, q.dtype), _dev(lut, np.float16), _dev(norm_weight, np.float16), _dev(residual, np.float16)
    out = torch.full((out_elems or N, ), float("nan"), dtype=torch.float16, device="cuda:0")
    so = torch.full((_lib.SSQ_SLOTS, ), float("nan"), dtype=torch.float32, device="cuda:0") if want_ssq else None
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    _lib.check(L.gq_anyprec_gemv_fused_ho(p(xt), p(out), p(qt), p(lt), N, K, bits, p(nw), EPS, p(rs), flags, None, 0, p(ssq_in), p(so),
                                          _lib.current_stream_ptr()), "gq_anyprec_gemv_fused_ho")
    torch.cuda.synchronize()
    return out.cpu().numpy(), (so.cpu().numpy() if so is not None else None)


def ssq_of(x16):
    from guidedquant_amd import _lib
    xt = _dev(x16, np.float16)
    s = torch.full((_lib.SSQ_SLOTS, ), float("nan"), dtype=torch.float32, device="cuda:0")
    _lib.check(_lib.lib().gq_ssq_rows(xt.data_ptr(), xt.numel(), s.data_ptr(), _lib.current_stream_ptr()), "gq_ssq_rows")
    return s


def test_ssq_rows_and_embedding():
    from guidedquant_amd import _lib
    rng = np.random.default_rng(0)
    for n in (4096, 2048, 8192, 1000):
        x = rng.normal(0, 3, n).astype(np.float16)
        s = ssq_of(x).cpu().numpy().astype(np.float64)
        assert np.isfinite(s).all() and abs(s.sum() - (x.astype(np.float64)**2).sum()) <= 1e-5 * (x.astype(np.float64)**2).sum()
    V, D = 300, 4096
    table = torch.from_numpy(rng.normal(0, 1, (V, D)).astype(np.float16)).to("cuda:0")
    tok = torch.tensor([123], dtype=torch.int32, device="cuda:0")
    out = torch.zeros(D, dtype=torch.float16, device="cuda:0")
    s = torch.full((_lib.SSQ_SLOTS, ), float("nan"), dtype=torch.float32, device="cuda:0")
    _lib.check(_lib.lib().gq_embed_lookup_ho(tok.data_ptr(), table.data_ptr(), out.data_ptr(), D, V, s.data_ptr(), _lib.current_stream_ptr()), "embed")
    torch.cuda.synchronize()
    assert torch.equal(out, table[123])
    want = (table[123].double()**2).sum().item()
    assert abs(s.double().sum().item() - want) <= 1e-5 * want


@pytest.mark.parametrize("N,K,flags", [(6144, 4096, 0), (28672, 4096, 4)])
def test_rmsnorm_prologue_with_handed_over_statistics(oracle, N, K, flags):
    """the 8B / 1B wqkv and w1w3 launches of the decode graph: same envelope against the oracle with and without the hand-over;
    between the two forms only elements whose normalised value sits on an fp16 rounding boundary may differ"""
    from guidedquant_amd import _lib
    bits = 2
    rng, q, lut = _layer(N, K, bits, 7 + N + K + flags)
    x = rng.normal(0, 1, K)
    x[rng.choice(K, 4, replace=False)] *= 30.0
    x = x.astype(np.float16)
    nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
    plan = _lib.lib().gq_anyprec_handover_plan(N, K, bits, 1, flags)
    assert plan & 1, "the stream kernel serves these launches: its RMSNorm prologue must take the statistics"
    out_e = N // 2 if flags & 4 else N
    base, _ = run_ho(x, q, lut, bits, norm_weight=nw, flags=flags, out_elems=out_e)
    got, _ = run_ho(x, q, lut, bits, norm_weight=nw, flags=flags, ssq_in=ssq_of(x), out_elems=out_e)
    assert np.isfinite(got).all()
    nd = (got.view(np.uint16) != base.view(np.uint16)).sum()
    assert nd <= max(4, out_e // 200), nd  # (a last-bit difference of the fp32 scale moves a handful of fp16 roundings)
    if not flags:
        xn = rmsnorm_ref(x, nw, EPS)
        rows = np.unique(np.concatenate([np.arange(0, 24), np.arange(N - 24, N), rng.integers(0, N, 48)]))
        _check_fast(got, xn, q, lut, bits, oracle, rows=rows)
    # wrong statistics must change the result: the prologue really reads the slots
    bad = ssq_of(x) * 4.0
    off, _ = run_ho(x, q, lut, bits, norm_weight=nw, flags=flags, ssq_in=bad, out_elems=out_e)
    assert np.abs(off.astype(np.float32) - got.astype(np.float32)).max() > 0.1 * np.abs(got.astype(np.float32)).max()


@pytest.mark.parametrize("bits,N,K", [(2, 4096, 4096), (2, 4096, 14336), (3, 4096, 4096), (3, 4096, 14336), (4, 4096, 14336), (2, 2048, 8192),
                                      (2, 8192, 8192), (4, 4096, 4096)])
def test_residual_epilogue_leaves_the_statistics(bits, N, K):
    """wo / w2 with the residual epilogue: outputs unchanged by the request, slots = sums of squares of the stored fp16 values -- from
    the GEMV kernel's own epilogue where it has the form (plan bit 1), else from the small launch behind it"""
    rng, q, lut = _layer(N, K, bits, 11 * bits + N + K)
    x = rng.normal(0, 1, K).astype(np.float16)
    res = rng.normal(0, 1, N).astype(np.float16)
    plain, _ = run_ho(x, q, lut, bits, residual=res, flags=1)
    got, s = run_ho(x, q, lut, bits, residual=res, flags=1, want_ssq=True)
    assert np.array_equal(got.view(np.uint16), plain.view(np.uint16))
    want = (got.astype(np.float64)**2).sum()
    assert np.isfinite(s).all() and abs(s.astype(np.float64).sum() - want) <= 1e-5 * want


def test_plan_of_the_benchmark_model():
    """8B 2-bit: both RMSNorm edges of a layer are free (consumer reads, producer writes in its epilogue); exact mode: none"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    assert L.gq_anyprec_handover_plan(6144, 4096, 2, 1, 0) & 1 and L.gq_anyprec_handover_plan(28672, 4096, 2, 1, 4) & 1
    assert L.gq_anyprec_handover_plan(4096, 4096, 2, 0, 1) & 2 and L.gq_anyprec_handover_plan(4096, 14336, 2, 0, 1) & 2
    L.gq_set_ap_mode(1)
    assert L.gq_anyprec_handover_plan(6144, 4096, 2, 1, 0) == 0 and L.gq_anyprec_handover_plan(4096, 4096, 2, 0, 1) == 0
    L.gq_set_ap_mode(0)


def test_decode_step_with_and_without_the_handover():
    """a 2-layer model at the 8B widths: logits of a decode step with the hand-over against the same step without it"""
    from guidedquant_amd import _lib
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.generate import random_init_
    from guidedquant_amd.model import ModelArgs, Transformer
    dev = torch.device("cuda:0")
    args = ModelArgs(block_size=256, vocab_size=4096, n_layer=3, n_head=32, dim=4096, intermediate_size=14336, n_local_heads=8, rope_base=500000.0, model_name="llama-test")
    model = Transformer(torch.float16, args, linear_class=APLinear, linear_kwargs=dict(bitwidth=2, device=dev), fuse_linears=True).to(device=dev, dtype=torch.float16).eval()
    random_init_(model, seed=3)
    model.setup_caches(1, 64)
    assert model._handover_plan(model.layers[0]) == dict(qkv_in=False, w13=False, w2_out=False)  # off by default (measured slower)
    os.environ["GQ_SSQ_HANDOVER"] = "1"
    assert model._handover_plan(model.layers[0]) == dict(qkv_in=True, w13=True, w2_out=True)

    def steps():
        outs = []
        for p, t in enumerate([5, 17, 900, 33]):
            lg = model.decode_native(torch.tensor([t], dtype=torch.int32, device=dev), torch.tensor([p], dtype=torch.int32, device=dev))
            outs.append(lg.float().cpu().numpy().reshape(-1).copy())
        return np.stack(outs)
    with torch.inference_mode():
        a = steps()
        os.environ["GQ_SSQ_HANDOVER"] = "0"
        _lib.lib().gq_reset_env_cache()
        b = steps()
    os.environ.pop("GQ_SSQ_HANDOVER")
    _lib.lib().gq_reset_env_cache()
    assert np.isfinite(a).all()
    assert np.abs(a - b).max() <= 2e-2 * np.abs(b).max() and np.linalg.norm(a - b) <= 3e-3 * np.linalg.norm(b)
    assert (a.argmax(1) == b.argmax(1)).all()
