"""GPU parity of the QTIP trellis-decoded matvec, the Hadamard transform and the QuantizedLinear forward against the
oracle (pinned to the reference's decode_compressed / matmul_hadU goldens).  Recipe of the random tests = the
reference's own kernel test (qtip/qtip-kernels/test_decompress_matvec.py:251-305): random int32 words,
codebook = clamp(randn/16, -1, 1) fp16 [1024], x = clamp(randn/16, -1, 1) fp16, seed 42; the reference asserts
allclose(out.half(), ref, atol=1e-5, rtol=0.01) -- here the bound is tighter (fp32 accumulation of exact products)."""
import numpy as np
import pytest

from conftest import golden_files

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run(compressed, tlut, x, M, K, R):
    from guidedquant_amd.qtip import qtip_kernels
    d = torch.device("cuda:0")
    out = torch.full((M, 1), float("nan"), dtype=torch.float32, device=d)
    fn = getattr(qtip_kernels, f"decompress_matvec_16_9_{R}_1_{M}_1_{K}")
    fn(out, torch.from_numpy(np.ascontiguousarray(compressed)).to(d), torch.from_numpy(np.ascontiguousarray(x).reshape(K, 1)).to(d),
       torch.from_numpy(np.ascontiguousarray(tlut).reshape(-1)).to(d))
    torch.cuda.synchronize()
    return out.cpu().numpy()[:, 0]


def _check(got, compressed, tlut, x, M, K, R, oracle):
    W = oracle.qtip_decode(compressed, tlut, M, K, R).astype(np.float64)
    xd = np.asarray(x, dtype=np.float64).reshape(-1)
    ref = W @ xd
    scale = np.abs(W) @ np.abs(xd)
    assert (np.abs(got - ref) <= 2e-6 * scale + 1e-7).all(), (np.abs(got - ref) / (scale + 1e-30)).max()
    # the reference test's own criterion
    assert np.allclose(got.astype(np.float16), ref.astype(np.float16), atol=1e-5, rtol=0.01)


@pytest.mark.parametrize("path", golden_files("qtip_R"))
def test_matvec_goldens(oracle, path):
    g = np.load(path)
    R, m, k = int(g["R"]), int(g["m"]), int(g["k"])
    got = _run(g["compressed"], g["tlut"], g["x"], m, k, R)
    _check(got, g["compressed"], g["tlut"], g["x"], m, k, R, oracle)
    np.testing.assert_allclose(got, g["y64"], rtol=0, atol=2e-6 * np.abs(g["W"].astype(np.float64)).sum(1).max())


@pytest.mark.parametrize("R", [2, 3, 4])
@pytest.mark.parametrize("M,K", [(4096, 4096), (11008, 4096), (4096, 11008), (32, 32), (1024, 8192)])
def test_matvec_reference_recipe(oracle, R, M, K):
    torch.manual_seed(42)
    compressed = torch.randint(torch.iinfo(torch.int32).min, torch.iinfo(torch.int32).max, (R * M * K // 32, ), dtype=torch.int32).numpy()
    tlut = torch.clamp(torch.randn(512, 2) / 16, -1, 1).to(torch.float16).numpy()
    x = torch.clamp(torch.randn(K, 1) / 16, -1, 1).to(torch.float16).numpy()
    got = _run(compressed, tlut, x, M, K, R)
    _check(got, compressed, tlut, x, M, K, R, oracle)


def test_sanity_zero_trellis_like_reference_smoke():
    """qtip/qtip-kernels/src/test.cu:11-66: zero trellis, codebook all 1.0, x all 1.0 -> every output equals K ...
    (state 0 -> idx 0 -> entry 0, no sign flip)"""
    M = K = 1024
    got = _run(np.zeros(2 * M * K // 32, dtype=np.int32), np.ones((512, 2), dtype=np.float16), np.ones(K, dtype=np.float16), M, K, 2)
    assert (got == K).all()


@pytest.mark.parametrize("path", golden_files("had_n"))
def test_hadamard_goldens(path):
    from guidedquant_amd.qtip import matmul_hadU_cuda, matmul_hadUt_cuda
    g = np.load(path)
    d = torch.device("cuda:0")
    X = torch.from_numpy(g["X"]).to(d)
    Kf = int(g["K"])
    hk = torch.from_numpy(g["hadK"].astype(np.float32)).to(d) if Kf > 1 else None
    Y = matmul_hadU_cuda(X, hk, Kf).cpu().numpy()
    Yt = matmul_hadUt_cuda(X, hk, Kf).cpu().numpy()
    np.testing.assert_allclose(Y, g["Y"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(Yt, g["Yt"], rtol=2e-5, atol=2e-5)


def test_hadamard_op_validation():
    d = torch.device("cuda:0")
    with pytest.raises(RuntimeError, match="power of two"):
        torch.ops.hadamard.hadamard(torch.zeros(2, 96, device=d), 1.0)


@pytest.mark.parametrize("R", [2, 4])
def test_quantized_linear_forward(oracle, R):
    """y = (SV*32) * H_m( W_hat @ ( H_n^T(x * SU) / 32 ) )   (bitshift.py:415-472) with power-of-two dims"""
    from guidedquant_amd.qtip import QuantizedLinear
    d = torch.device("cuda:0")
    N, K = 512, 256
    rng = np.random.default_rng(R)
    lin = QuantizedLinear(K, N, 16, 16, 16, R, 2, 9, 'quantlut_sym', device=d)
    assert lin.trellis.shape == ((N // 16) * (K // 16), 16 * R) and lin.trellis.dtype == torch.int16
    trellis = rng.integers(-2**15, 2**15, lin.trellis.shape, dtype=np.int64).astype(np.int16)
    tlut = np.clip(rng.normal(0, 1 / 16, (512, 2)), -1, 1).astype(np.float16)
    SU = np.sign(rng.normal(0, 1, K)).astype(np.float16)
    SV = (np.sign(rng.normal(0, 1, N)) * rng.uniform(0.5, 1.5, N)).astype(np.float32)
    lin.load_state_dict({"trellis": torch.from_numpy(trellis), "tlut": torch.from_numpy(tlut), "SU": torch.from_numpy(SU),
                         "SV": torch.from_numpy(SV), "rcp": torch.tensor(0), "tp_rank": torch.tensor(8)})
    x = rng.normal(0, 1, (1, 1, K)).astype(np.float16)
    y = lin(torch.from_numpy(x).to(d)).float().cpu().numpy().reshape(N)
    W = oracle.qtip_decode(trellis.view(np.int32).reshape(-1), tlut, N, K, R).astype(np.float64)
    xs = x.reshape(K).astype(np.float64) * SU.astype(np.float64)
    xh = oracle.matmul_hadU(xs.astype(np.float32)[None], None, transpose=True)[0].astype(np.float64) / 32
    z = W @ xh.astype(np.float16).astype(np.float64)  # the kernel takes x in fp16 (lib/codebook/__init__.py:108)
    zh = oracle.matmul_hadU(z.astype(np.float32)[None], None)[0].astype(np.float64)
    ref = zh * (SV.astype(np.float64) * 32)
    assert np.abs(y - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-4


def test_qtip_backend_model_decodes():
    """generate.py --backend qtip: an unfused QuantizedLinear model (power-of-two dims: no Hadamard factor tables needed)
    steps through the reference decode loop; every linear runs hadamard -> trellis matvec -> hadamard on the HIP ops."""
    from guidedquant_amd import model as gm
    from guidedquant_amd.generate import load_model, decode_one_token
    gm.transformer_configs["qtip-gpu-test"] = dict(model_name="llama-qtip-gpu-test", block_size=128, vocab_size=512, n_layer=2,
                                                   n_head=8, dim=1024, intermediate_size=2048, n_local_heads=4)
    try:
        m = load_model("qtip-gpu-test", "cuda:0", "qtip", 2, random_init=True)
    finally:
        del gm.transformer_configs["qtip-gpu-test"]
    with torch.device("cuda:0"):
        m.setup_caches(max_batch_size=1, max_seq_length=64)
    tok = torch.tensor([[1]], dtype=torch.int32, device="cuda:0")
    for pos in range(3):
        with torch.no_grad():
            nxt, probs = decode_one_token(m, tok, torch.tensor([pos], dtype=torch.int32, device="cuda:0"), temperature=0.0, top_k=32)
        assert nxt.shape[-1] == 1 and 0 <= int(nxt.reshape(-1)[0]) < 512
        assert bool(torch.isfinite(probs.float()).all())
        tok = nxt.reshape(1, 1).to(torch.int32)


# ----------------------------------------------------------------------------- fused linear (gq_qtip_linear_in / _out)
def _fused_linear(lin, x16, pro=0, x2=None, normw=None, eps=1e-5, resid=None, group=None):
    """runs the two-launch fused linear for the QuantizedLinear modules in `group` (default [lin]) on input x16 (fp16 [K])"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    mods = group or [lin]
    d = x16.device
    K = mods[0].in_features
    su = [m.SU.float().contiguous() for m in mods]
    sv = [(m.SV.float() * 32.0).contiguous() for m in mods]
    y32 = [torch.full((m.out_features,), float("nan"), dtype=torch.float32, device=d) for m in mods]
    outs = [torch.full((m.out_features,), float("nan"), dtype=torch.float16, device=d) for m in mods]
    ain = (_lib.GqQtipIn * len(mods))()
    aout = (_lib.GqQtipOut * len(mods))()
    for i, m in enumerate(mods):
        ain[i] = _lib.GqQtipIn(m.trellis.data_ptr(), su[i].data_ptr(), m.tlut.data_ptr(), y32[i].data_ptr(), m.out_features)
        aout[i] = _lib.GqQtipOut(y32[i].data_ptr(), sv[i].data_ptr(), resid.data_ptr() if resid is not None else None,
                                 outs[i].data_ptr(), m.out_features)
    st = _lib.current_stream_ptr()
    _lib.check(L.gq_qtip_linear_in(x16.data_ptr(), x2.data_ptr() if x2 is not None else None,
                                   normw.data_ptr() if normw is not None else None, eps, pro, K, mods[0].K, len(mods), ain, 0, None, 1, st), "in")
    _lib.check(L.gq_qtip_linear_out(len(mods), aout, st), "out")
    torch.cuda.synchronize()
    return outs


def _rand_qlinear(K, M, R, seed):
    from guidedquant_amd.qtip import QuantizedLinear
    d = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    lin = QuantizedLinear(K, M, 16, 16, 16, R, 2, 9, "quantlut_sym", device=d)
    lin.trellis.copy_(torch.randint(-2**15, 2**15 - 1, lin.trellis.shape, dtype=torch.int16, generator=g))
    lin.tlut.data.copy_(torch.clamp(torch.randn(512, 2, generator=g) / 16, -1, 1).half())
    lin.SU.copy_((torch.randint(0, 2, (K,), generator=g) * 2 - 1).half() * (1 + 0.1 * torch.rand(K, generator=g)).half())
    lin.SV.copy_((torch.randint(0, 2, (M,), generator=g) * 2 - 1).float() * (1 + 0.1 * torch.rand(M, generator=g)))
    return lin


@pytest.mark.parametrize("R", [2, 3, 4])
@pytest.mark.parametrize("M,K", [(4096, 4096), (1024, 8192), (8192, 4096), (256, 128), (2048, 16384)])
def test_fused_linear_is_bit_identical_to_the_op_chain(R, M, K):
    """gq_qtip_linear_in + _out == x*SU -> hadamard -> /32 -> half -> decompress_matvec -> hadamard -> *(SV*32) -> half
    as QuantizedLinear.forward runs it on the separate ops (bitshift.py:415-472): same arithmetic, same butterfly order"""
    lin = _rand_qlinear(K, M, R, seed=R * 1000 + M + K)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(1, 1, K, generator=g).half().cuda()
    with torch.no_grad():
        want = lin(x).reshape(-1)
    got = _fused_linear(lin, x.reshape(-1).contiguous())[0]
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_fused_linear_prologues_and_residual(oracle):
    """RMSNorm / silu*mul prologues and the residual epilogue against the module chain fed with the torch-side ops of the
    decode step (model.py:266,281-292,311-313); q/k/v-style group launch against per-linear launches (fp32 sum order of
    the K-split may differ with the block shape: tolerance 2 fp16 ulp)"""
    from guidedquant_amd.model import RMSNorm
    K, M, R = 4096, 4096, 2
    d = torch.device("cuda:0")
    a, b, c = (_rand_qlinear(K, m, R, seed=s) for m, s in ((M, 1), (1024, 2), (1024, 3)))
    g = torch.Generator(device="cpu").manual_seed(11)
    x = (torch.randn(K, generator=g) * 0.7).half().to(d)
    x2 = torch.randn(K, generator=g).half().to(d)
    norm = RMSNorm(K, 1e-5).to(d).half()
    norm.weight.data.copy_((1 + 0.2 * torch.randn(K, generator=g)).half())
    resid = torch.randn(M, generator=g).half().to(d)

    def close(got, want, ulps=2):
        # the prologue's sum of squares is added in another order than torch's mean(): 1/rms may differ in its last bit, which
        # flips the fp16 rounding of a few normalised activations -> a perturbation of ~2^-11 of the output's rms
        gw, ww = got.float(), want.float()
        tol = ulps * torch.maximum(torch.abs(ww) * 2.0**-10, torch.full_like(ww, 2.0**-24)) + 3e-3 * torch.sqrt(torch.mean(ww * ww))
        assert bool((torch.abs(gw - ww) <= tol).all()), float(torch.abs(gw - ww).max())

    with torch.no_grad():
        xn = norm(x.view(1, 1, K))
        want = [m(xn).reshape(-1).clone() for m in (a, b, c)]
        got = _fused_linear(a, x, pro=1, normw=norm.weight.data, eps=1e-5, group=[a, b, c])
        for gq, w in zip(got, want):
            close(gq, w)
        xs = torch.nn.functional.silu(x.view(1, 1, K)) * x2.view(1, 1, K)
        want = resid + a(xs).reshape(-1)
        got = _fused_linear(a, x, pro=2, x2=x2, resid=resid)[0]
        close(got, want)
        assert torch.equal(_fused_linear(a, xs.reshape(-1).contiguous(), resid=resid)[0].view(torch.int16), want.view(torch.int16))


def test_folded_transform_out_is_bit_identical():
    """gq_qtip_linear_in with n_prev: the producer's transform-out (+ residual) rebuilt in the consumer's prologue ==
    gq_qtip_linear_out followed by gq_qtip_linear_in; the stored vector equals what gq_qtip_linear_out writes"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    K, R = 4096, 2
    prod = [_rand_qlinear(2048, K, R, seed=21), _rand_qlinear(2048, K, R, seed=22)]  # two producers with M == K
    cons = _rand_qlinear(K, 1024, R, seed=23)
    g = torch.Generator(device="cpu").manual_seed(3)
    xin = torch.randn(2048, generator=g).half().to(d)
    resid = torch.randn(K, generator=g).half().to(d)
    normw = (1 + 0.2 * torch.randn(K, generator=g)).half().to(d)
    st = _lib.current_stream_ptr()
    su_p = [m.SU.float().contiguous() for m in prod]
    sv_p = [(m.SV.float() * 32).contiguous() for m in prod]
    y32 = [torch.zeros(K, dtype=torch.float32, device=d) for _ in prod]
    pin = (_lib.GqQtipIn * 2)(*[_lib.GqQtipIn(m.trellis.data_ptr(), su_p[i].data_ptr(), m.tlut.data_ptr(), y32[i].data_ptr(), K)
                                for i, m in enumerate(prod)])
    _lib.check(L.gq_qtip_linear_in(xin.data_ptr(), None, None, 0.0, 0, 2048, R, 2, pin, 0, None, 1, st), "producers")
    su_c, sv_c = cons.SU.float().contiguous(), (cons.SV.float() * 32).contiguous()
    for pro, nprev in ((0, 1), (1, 1), (2, 2)):
        outs = [torch.zeros(K, dtype=torch.float16, device=d) for _ in prod]
        rs = resid if pro != 2 else None
        pout = (_lib.GqQtipOut * 2)(*[_lib.GqQtipOut(y32[i].data_ptr(), sv_p[i].data_ptr(), rs.data_ptr() if rs is not None else None,
                                                     outs[i].data_ptr(), K) for i in range(2)])
        _lib.check(L.gq_qtip_linear_out(nprev, pout, st), "out")
        yc = [torch.zeros(1024, dtype=torch.float32, device=d) for _ in range(2)]
        cin = [(_lib.GqQtipIn * 1)(_lib.GqQtipIn(cons.trellis.data_ptr(), su_c.data_ptr(), cons.tlut.data_ptr(), yc[j].data_ptr(), 1024))
               for j in range(2)]
        _lib.check(L.gq_qtip_linear_in(outs[0].data_ptr(), outs[1].data_ptr(), normw.data_ptr(), 1e-5, pro, K, R, 1, cin[0], 0, None, 1, st), "two-launch")
        stored = [torch.full((K,), float("nan"), dtype=torch.float16, device=d) for _ in prod]
        pfold = (_lib.GqQtipOut * 2)(*[_lib.GqQtipOut(y32[i].data_ptr(), sv_p[i].data_ptr(), rs.data_ptr() if rs is not None else None,
                                                      stored[i].data_ptr(), K) for i in range(2)])
        _lib.check(L.gq_qtip_linear_in(None, None, normw.data_ptr(), 1e-5, pro, K, R, 1, cin[1], nprev, pfold, 1, st), "folded")
        torch.cuda.synchronize()
        assert torch.equal(yc[0], yc[1]) and bool(torch.isfinite(yc[0]).all())
        for i in range(nprev):
            assert torch.equal(stored[i].view(torch.int16), outs[i].view(torch.int16))


@pytest.mark.parametrize("K,parts", [(4096, 1), (4096, 2), (1024, 1), (8192, 1)])
def test_transform_out_with_the_consumers_transform_in_is_bit_identical(K, parts):
    """gq_qtip_linear_out_in (round 5): the transform-out of a producer (+ residual) and, in the same launch, the RMSNorm -> SU -> Hadamard
    prologue of the three linears that read it; their matvec launch on the pre-transformed vectors (GQ_QPRO_PRETRANSFORMED, one vector
    per linear) gives the SAME sums, bit for bit, as gq_qtip_linear_out followed by gq_qtip_linear_in with the RMSNorm prologue, and
    the stored hidden state is the same"""
    import ctypes
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    R = 2
    prod = _rand_qlinear(2048, K, R, seed=31 + K)
    cons = [_rand_qlinear(K, m, R, seed=40 + i) for i, m in enumerate((K, 1024, 1024))]
    g = torch.Generator(device="cpu").manual_seed(5 + parts)
    xin = torch.randn(2048, generator=g).half().to(d)
    resid = torch.randn(K, generator=g).half().to(d)
    normw = (1 + 0.2 * torch.randn(K, generator=g)).half().to(d)
    st = _lib.current_stream_ptr()
    su_p, sv_p = prod.SU.float().contiguous(), (prod.SV.float() * 32).contiguous()
    y32 = torch.zeros(parts * K, dtype=torch.float32, device=d)
    pin = (_lib.GqQtipIn * 1)(_lib.GqQtipIn(prod.trellis.data_ptr(), su_p.data_ptr(), prod.tlut.data_ptr(), y32.data_ptr(), K))
    _lib.check(L.gq_qtip_linear_in(xin.data_ptr(), None, None, 0.0, 0, 2048, R, 1, pin, 0, None, parts, st), "producer")
    su_c = [m.SU.float().contiguous() for m in cons]
    # two launches: transform-out, then the consumers with their RMSNorm prologue
    h_ref = torch.zeros(K, dtype=torch.float16, device=d)
    pout = (_lib.GqQtipOut * 1)(_lib.GqQtipOut(y32.data_ptr(), sv_p.data_ptr(), resid.data_ptr(), h_ref.data_ptr(), K, parts))
    _lib.check(L.gq_qtip_linear_out(1, pout, st), "out")
    y_ref = [torch.full((m.out_features,), float("nan"), dtype=torch.float32, device=d) for m in cons]
    cin = (_lib.GqQtipIn * 3)(*[_lib.GqQtipIn(m.trellis.data_ptr(), su_c[i].data_ptr(), m.tlut.data_ptr(), y_ref[i].data_ptr(), m.out_features)
                                for i, m in enumerate(cons)])
    _lib.check(L.gq_qtip_linear_in(h_ref.data_ptr(), None, normw.data_ptr(), 1e-5, 1, K, R, 3, cin, 0, None, 1, st), "consumers")
    # one launch for transform-out + the three transform-ins, then the bare matvecs
    h = torch.full((K,), float("nan"), dtype=torch.float16, device=d)
    xt = [torch.full((K,), float("nan"), dtype=torch.float16, device=d) for _ in cons]
    pout2 = (_lib.GqQtipOut * 1)(_lib.GqQtipOut(y32.data_ptr(), sv_p.data_ptr(), resid.data_ptr(), h.data_ptr(), K, parts))
    sup = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in su_c])
    xtp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in xt])
    _lib.check(L.gq_qtip_linear_out_in(pout2, normw.data_ptr(), 1e-5, 3, sup, xtp, st), "out_in")
    y_new = [torch.full((m.out_features,), float("nan"), dtype=torch.float32, device=d) for m in cons]
    cpre = (_lib.GqQtipIn * 3)(*[_lib.GqQtipIn(m.trellis.data_ptr(), xt[i].data_ptr(), m.tlut.data_ptr(), y_new[i].data_ptr(), m.out_features)
                                 for i, m in enumerate(cons)])
    _lib.check(L.gq_qtip_linear_in(None, None, None, 0.0, 3, K, R, 3, cpre, 0, None, 1, st), "pre-transformed")
    torch.cuda.synchronize()
    assert torch.equal(h.view(torch.int16), h_ref.view(torch.int16))
    for a, b in zip(y_new, y_ref):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    assert L.gq_qtip_linear_out_in(pout2, normw.data_ptr(), 1e-5, 4, sup, xtp, st) != 0   # at most 3 consumers
    assert L.gq_qtip_linear_in(None, None, None, 0.0, 1, K, R, 3, cpre, 0, None, 1, st) != 0  # x == NULL only with the pre-transformed form


@pytest.mark.parametrize("M,K", [(4096, 4096), (2048, 8192), (1024, 1024)])
def test_split_k_and_pretransformed_input(M, K):
    """ksplit = 2 (two blocks per band, partial sums added by gq_qtip_linear_out with parts = 2) agrees with ksplit = 1 to
    fp32 rounding of one extra addition; GQ_QPRO_PRETRANSFORMED on the fp16 vector the prologue would have produced is the
    bare matvec (== gq_qtip_matvec, same block shape)"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    lin = _rand_qlinear(K, M, 2, seed=M + K)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(K, generator=g).half().to(d)
    su, sv = lin.SU.float().contiguous(), (lin.SV.float() * 32).contiguous()
    res = []
    for ks in (1, 2):
        y32 = torch.full((2 * M,), float("nan"), dtype=torch.float32, device=d)
        out = torch.zeros(M, dtype=torch.float16, device=d)
        ain = (_lib.GqQtipIn * 1)(_lib.GqQtipIn(lin.trellis.data_ptr(), su.data_ptr(), lin.tlut.data_ptr(), y32.data_ptr(), M))
        aout = (_lib.GqQtipOut * 1)(_lib.GqQtipOut(y32.data_ptr(), sv.data_ptr(), None, out.data_ptr(), M, ks))
        _lib.check(L.gq_qtip_linear_in(x.data_ptr(), None, None, 0.0, 0, K, 2, 1, ain, 0, None, ks, None), "in")
        _lib.check(L.gq_qtip_linear_out(1, aout, None), "out")
        torch.cuda.synchronize()
        res.append((y32.clone(), out.clone()))
    y1 = res[0][0][:M]
    y2 = res[1][0][:M] + res[1][0][M:]
    assert bool(torch.isfinite(y2).all())
    assert float((y1 - y2).abs().max()) <= 2e-6 * float(y1.abs().max()) + 1e-7
    assert float((res[0][1].float() - res[1][1].float()).abs().max()) <= 2.0**-9 * float(res[0][1].float().abs().max())
    # pre-transformed input: feed the matvec's own fp16 input
    from guidedquant_amd.qtip import matmul_hadUt_cuda
    xs = (matmul_hadUt_cuda(x.view(1, K).float() * lin.SU, None, 1) / 32).half().reshape(-1).contiguous()
    y3 = torch.full((M,), float("nan"), dtype=torch.float32, device=d)
    ain = (_lib.GqQtipIn * 1)(_lib.GqQtipIn(lin.trellis.data_ptr(), None, lin.tlut.data_ptr(), y3.data_ptr(), M))
    _lib.check(L.gq_qtip_linear_in(xs.data_ptr(), None, None, 0.0, 3, K, 2, 1, ain, 0, None, 1, None), "pre")
    torch.cuda.synchronize()
    assert torch.equal(y3, y1)


def test_fused_linear_validation():
    from guidedquant_amd import _lib
    L = _lib.lib()
    lin = _rand_qlinear(128, 64, 2, seed=5)
    x = torch.zeros(128, dtype=torch.float16, device="cuda:0")
    with pytest.raises(RuntimeError, match="power of two"):
        ain = (_lib.GqQtipIn * 1)(_lib.GqQtipIn(lin.trellis.data_ptr(), lin.SU.float().data_ptr(), lin.tlut.data_ptr(), x.data_ptr(), 64))
        _lib.check(L.gq_qtip_linear_in(x.data_ptr(), None, None, 0.0, 0, 96, 2, 1, ain, 0, None, 1, None), "in")
    with pytest.raises(RuntimeError, match="prologue operand"):
        ain = (_lib.GqQtipIn * 1)(_lib.GqQtipIn(lin.trellis.data_ptr(), lin.SU.float().data_ptr(), lin.tlut.data_ptr(), x.data_ptr(), 64))
        _lib.check(L.gq_qtip_linear_in(x.data_ptr(), None, None, 0.0, 1, 128, 2, 1, ain, 0, None, 1, None), "in")


def test_qtip_native_decode_matches_module_forward():
    """the native QTIP decode step (Transformer.decode_native; 9 launches per layer) against the module-by-module forward
    of the same model: logits agree to fp16 rounding noise, greedy tokens agree"""
    import os
    from guidedquant_amd import model as gm
    from guidedquant_amd.generate import load_model
    gm.transformer_configs["qtip-native-test"] = dict(model_name="llama-qtip-native-test", block_size=128, vocab_size=512, n_layer=3,
                                                      n_head=8, dim=1024, intermediate_size=2048, n_local_heads=4)
    try:
        m = load_model("qtip-native-test", "cuda:0", "qtip", 2, random_init=True)
    finally:
        del gm.transformer_configs["qtip-native-test"]
    with torch.device("cuda:0"):
        m.setup_caches(max_batch_size=1, max_seq_length=64)
    assert m.native_ready() and m._native_kind() == "qtip"
    toks = [1, 17, 200, 5]
    ref = []
    with torch.no_grad():
        for p, t in enumerate(toks):
            ref.append(m(torch.tensor([[t]], dtype=torch.int32, device="cuda:0"), torch.tensor([p], dtype=torch.int32, device="cuda:0"))
                       .float().reshape(-1).clone())
        for b in m.layers:  # fresh caches for the native pass
            b.attention.kv_cache.k_cache.zero_()
            b.attention.kv_cache.v_cache.zero_()
        for p, t in enumerate(toks):
            got = m.decode_native(torch.tensor([t], dtype=torch.int32, device="cuda:0"),
                                  torch.tensor([p], dtype=torch.int32, device="cuda:0")).float().reshape(-1)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(got).all())
            err = float(torch.abs(got - ref[p]).max()) / (float(torch.abs(ref[p]).max()) + 1e-9)
            assert err < 2e-2, err
            assert int(got.argmax()) == int(ref[p].argmax()) or err < 5e-3


@pytest.mark.parametrize("Ms,parts,resid", [([4096], 1, True), ([4096, 1024, 1024], 1, False), ([8192], 2, True), ([128], 1, False), ([2048, 4096], 3, True)])
def test_segmented_transform_out_matches_the_one_block_form(Ms, parts, resid):
    """gq_qtip_linear_out_seg (M / 128 blocks per linear: segments combined with the signs of the block's Sylvester row, one 128-point
    transform) against gq_qtip_linear_out (one block, the full butterflies): the same additions in another order -- outputs agree to
    fp32 rounding, i.e. a one-ulp fp16 flip here and there"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(sum(Ms) + parts)
    outs = {}
    ys = [torch.randn(parts, M, device=d, generator=g) * 3 for M in Ms]
    svs = [(torch.randn(M, device=d, generator=g) * 0.3 + 1) * 32 for M in Ms]
    res = [torch.randn(M, device=d, generator=g).half() if resid else None for M in Ms]
    for name in ("gq_qtip_linear_out", "gq_qtip_linear_out_seg"):
        o = [torch.full((M,), float("nan"), dtype=torch.float16, device=d) for M in Ms]
        arr = (_lib.GqQtipOut * len(Ms))(*[_lib.GqQtipOut(ys[i].data_ptr(), svs[i].data_ptr(), res[i].data_ptr() if resid else None, o[i].data_ptr(), M, parts)
                                           for i, M in enumerate(Ms)])
        _lib.check(getattr(L, name)(len(Ms), arr, _lib.current_stream_ptr()), name)
        torch.cuda.synchronize()
        outs[name] = [t.float() for t in o]
    for a, b in zip(outs["gq_qtip_linear_out"], outs["gq_qtip_linear_out_seg"]):
        assert bool(torch.isfinite(b).all())
        # one fp16 ulp of the larger magnitude
        tol = torch.maximum(a.abs(), b.abs()) * 2.0**-10 + 1e-6
        assert bool(((a - b).abs() <= tol).all()), float(((a - b).abs() / tol).max())
        assert float((a != b).float().mean()) < 0.02  # and rarely at all


@pytest.mark.parametrize("heads", [(8, 4, 1024), (32, 32, 4096), (16, 2, 2048)])
def test_attention_with_the_qkv_transform_out_folded_in(monkeypatch, heads):
    """gq_attn_decode_qtip (the transform-out of q / k / v inside the attention launch: per head, the segments of the sums combined
    with the signs of its row of the Sylvester matrix, then one head_dim-point transform) against gq_qtip_linear_out +
    gq_attn_decode_split: the same additions in another order -- rotated keys and logits of a decode step agree to fp32 / fp16
    rounding noise; MHA (Llama-2-7b head layout), GQA, several positions"""
    from guidedquant_amd import model as gm
    from guidedquant_amd.generate import load_model
    H, Hkv, dim = heads
    gm.transformer_configs["qtip-fold-test"] = dict(model_name="llama-qtip-fold-test", block_size=128, vocab_size=512, n_layer=2,
                                                    n_head=H, dim=dim, intermediate_size=2048, n_local_heads=Hkv)
    outs = {}
    try:
        for fold in ("1", "0"):
            monkeypatch.setenv("GQ_QTIP_ATTN_FOLD", fold)
            torch.manual_seed(7)
            m = load_model("qtip-fold-test", "cuda:0", "qtip", 2, random_init=True)
            with torch.device("cuda:0"):
                m.setup_caches(max_batch_size=1, max_seq_length=64)
            st = m._native_state()
            assert (st["qtip_layers"][0]["attn_qt"] is not None) == (fold == "1")
            res = []
            for p, t in enumerate([1, 17, 200, 5, 9]):
                got = m.decode_native(torch.tensor([t], dtype=torch.int32, device="cuda:0"), torch.tensor([p], dtype=torch.int32, device="cuda:0"))
                torch.cuda.synchronize()
                res.append(got.reshape(-1).clone())
                res.append(m.layers[1].attention.kv_cache.k_cache[0, :, p].reshape(-1).clone())
            outs[fold] = res
            del m
    finally:
        del gm.transformer_configs["qtip-fold-test"]
    for i, (a, b) in enumerate(zip(outs["1"], outs["0"])):
        a, b = a.float(), b.float()
        assert bool(torch.isfinite(a).all())
        # keys: fp16 values from fp32 sums that differ in the last bits -- a one-ulp flip here and there; logits: fp16 noise on top
        tol = 2e-3 if i % 2 else 1e-2
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-6, (i, float((a - b).abs().max()), float(b.abs().max()))
        if i % 2 == 0:
            assert int(a.argmax()) == int(b.argmax())


# ----------------------------------------------------------------------------- widths with a Hadamard factor (gq_qtip_transform)
@pytest.mark.parametrize("name", ["had_n11008", "had_n14336"])
def test_factor_transform_goldens(name):
    """gq_qtip_transform against the reference-generated matmul_hadU / matmul_hadUt vectors (tests/golden/had_n*.npz:
    11008 = 172 * 64, 14336 = 28 * 512); output side with vec = 1 gives half(Y), input side half(Yt / 32)"""
    import os
    from guidedquant_amd import _lib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    L = _lib.lib()
    d = torch.device("cuda:0")
    n, Kf = int(g["n"]), int(g["K"])
    hk = torch.from_numpy(g["hadK"].astype(np.float32)).to(d).contiguous()
    ones = torch.ones(n, dtype=torch.float32, device=d)
    for row in range(2):
        y32 = torch.from_numpy(g["X"][row]).to(d).contiguous()
        out = torch.full((n,), float("nan"), dtype=torch.float16, device=d)
        xf = (_lib.GqQtipXf * 1)(_lib.GqQtipXf(y32.data_ptr(), ones.data_ptr(), hk.data_ptr(), None, out.data_ptr()))
        _lib.check(L.gq_qtip_transform(0, None, None, None, 0.0, 0, 1, xf, n, Kf, 0, None), "out side")
        torch.cuda.synchronize()
        want = g["Y"][row]
        np.testing.assert_allclose(out.float().cpu().numpy(), want, rtol=2e-3, atol=2e-3 * np.abs(want).max())
        # input side: the source is fp16, so compare on an fp16-representable input through linearity of the golden pair
        x16 = torch.from_numpy(g["X"][row]).half().to(d)
        out2 = torch.full((n,), float("nan"), dtype=torch.float16, device=d)
        xf2 = (_lib.GqQtipXf * 1)(_lib.GqQtipXf(None, ones.data_ptr(), hk.data_ptr(), None, out2.data_ptr()))
        _lib.check(L.gq_qtip_transform(1, x16.data_ptr(), None, None, 0.0, 0, 1, xf2, n, Kf, 1, None), "in side")
        torch.cuda.synchronize()
        want_t = g["Yt"][row] / 32.0
        np.testing.assert_allclose(out2.float().cpu().numpy(), want_t, rtol=4e-3, atol=4e-3 * np.abs(want_t).max())


@pytest.mark.parametrize("parts", [1, 2])
def test_mlp_mid_equals_the_two_transform_chain(parts):
    """gq_qtip_mlp_mid + gq_qtip_linear_in_rows against gq_qtip_transform (output side of gate / up) -> gq_qtip_transform (input side,
    silu * up) -> gq_qtip_linear_in (pre-transformed) at Llama-2-7b's MLP width 11008 = 172 * 64: gate / up fp16 bit-identical
    (same additions in the same order), the matvec input equal up to fp32 rounding of the re-ordered input side (a few fp16 ulps
    on a few elements), the sums of down within that noise."""
    import os
    from guidedquant_amd import _lib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "had_n11008.npz"))
    L = _lib.lib()
    d = torch.device("cuda:0")
    n, Kf, M, R = 11008, 172, 256, 2
    hk = torch.from_numpy(g["hadK"].astype(np.float32)).to(d).contiguous()
    hkT16, hk16 = hk.t().contiguous().half(), hk.half().contiguous()
    gen = torch.Generator(device="cpu").manual_seed(11)
    yg = (torch.randn(parts, n, generator=gen) * 3.0).to(d)
    yu = (torch.randn(parts, n, generator=gen) * 3.0).to(d)
    svg = ((torch.rand(n, generator=gen) + 0.5) * torch.sign(torch.randn(n, generator=gen)) * 32.0 * 0.02).to(d)
    svu = ((torch.rand(n, generator=gen) + 0.5) * torch.sign(torch.randn(n, generator=gen)) * 32.0 * 0.02).to(d)
    sud = torch.sign(torch.randn(n, generator=gen)).to(d)
    # two-launch chain (parts: the output side of gq_qtip_transform reads one vector -- add the parts the way the kernels do)
    ysum_g, ysum_u = yg[0].clone(), yu[0].clone()
    for p in range(1, parts):
        ysum_g += yg[p]
        ysum_u += yu[p]
    g16 = torch.empty(n, dtype=torch.float16, device=d)
    u16 = torch.empty(n, dtype=torch.float16, device=d)
    xf = (_lib.GqQtipXf * 2)(_lib.GqQtipXf(ysum_g.data_ptr(), svg.data_ptr(), hk.data_ptr(), None, g16.data_ptr()),
                             _lib.GqQtipXf(ysum_u.data_ptr(), svu.data_ptr(), hk.data_ptr(), None, u16.data_ptr()))
    _lib.check(L.gq_qtip_transform(0, None, None, None, 0.0, 0, 2, xf, n, Kf, 0, None), "out side")
    xs16 = torch.empty(n, dtype=torch.float16, device=d)
    xf2 = (_lib.GqQtipXf * 1)(_lib.GqQtipXf(None, sud.data_ptr(), hk.data_ptr(), None, xs16.data_ptr()))
    _lib.check(L.gq_qtip_transform(1, g16.data_ptr(), u16.data_ptr(), None, 0.0, 2, 1, xf2, n, Kf, 1, None), "in side")
    # one launch
    z32 = torch.full((n,), float("nan"), dtype=torch.float32, device=d)
    g16b = torch.full((n,), float("nan"), dtype=torch.float16, device=d)
    u16b = torch.full((n,), float("nan"), dtype=torch.float16, device=d)
    mid = _lib.GqQtipMid(yg.data_ptr(), yu.data_ptr(), svg.data_ptr(), svu.data_ptr(), hkT16.data_ptr(), sud.data_ptr(), hk16.data_ptr(),
                         z32.data_ptr(), g16b.data_ptr(), u16b.data_ptr())
    import ctypes
    _lib.check(L.gq_qtip_mlp_mid(ctypes.pointer(mid), parts, n, Kf, None), "mid")
    torch.cuda.synchronize()
    assert torch.equal(g16.view(torch.int16), g16b.view(torch.int16))
    assert torch.equal(u16.view(torch.int16), u16b.view(torch.int16))
    assert bool(torch.isfinite(z32).all())
    # the row transforms that are left, in float64, against the chain's fp16 matvec input
    z = z32.double().cpu().numpy().reshape(Kf, 64)
    H = np.array([[1.0]])
    while H.shape[0] < 64:
        H = np.block([[H, H], [H, -H]])
    want = (z @ H).reshape(-1) * n ** -0.5 / 32.0
    got = xs16.float().cpu().numpy().astype(np.float64)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1.5e-3 * scale, np.abs(got - want).max() / scale
    # and through the matvec: sums of a 256-row linear from both inputs
    rng = np.random.default_rng(5)
    comp = torch.from_numpy(rng.integers(0, 2 ** 32, size=R * M * n // 32, dtype=np.uint32).view(np.int32)).to(d)
    tlut = (torch.randn(1024, generator=gen) * 0.5).half().to(d)
    y_a = torch.zeros(M, dtype=torch.float32, device=d)
    y_b = torch.zeros(M, dtype=torch.float32, device=d)
    arr_a = (_lib.GqQtipIn * 1)(_lib.GqQtipIn(comp.data_ptr(), None, tlut.data_ptr(), y_a.data_ptr(), M))
    arr_b = (_lib.GqQtipIn * 1)(_lib.GqQtipIn(comp.data_ptr(), None, tlut.data_ptr(), y_b.data_ptr(), M))
    _lib.check(L.gq_qtip_linear_in(xs16.data_ptr(), None, None, 0.0, 3, n, R, 1, arr_a, 0, None, 1, None), "pre")
    _lib.check(L.gq_qtip_linear_in_rows(z32.data_ptr(), n, 64, R, 1, arr_b, 1, None), "rows")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y_b).all())
    ref = float(torch.abs(y_a).max())
    assert float(torch.abs(y_a - y_b).max()) <= 4e-3 * ref, float(torch.abs(y_a - y_b).max()) / ref
    # the row prologue itself: fp16 input of the matvec bit-identical to rounding the float64 rows?  (not required; the envelope
    # above is the contract)  Validation: wrong widths are refused, not mis-served
    assert L.gq_qtip_mlp_mid(ctypes.pointer(mid), parts, 28672, 28, None) == _lib.GQ_ENOTSUP
    assert L.gq_qtip_linear_in(z32.data_ptr(), None, None, 0.0, 4, n, R, 1, arr_b, 0, None, 1, None) != 0


@pytest.mark.parametrize("inter,key,mid", [(11008, "had172", "1"), (11008, "had172", "0"), (28672, "had28", "1")])
def test_qtip_native_decode_with_factor_width(tmp_path, monkeypatch, inter, key, mid):
    """MLP width 11008 = 172 * 64 (Llama-2-7b's) / 28672 = 28 * 1024 (Llama-2-70b's): the table comes from the caller (GQ_HADAMARD_TABLES; here the golden
    fixture's copy).  Native decode (fused kernels on the power-of-two sides, gq_qtip_transform + gq_qtip_matvec on the
    factor side) against the module-by-module forward."""
    import os
    from guidedquant_amd import model as gm, qtip
    from guidedquant_amd.generate import load_model
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "had_n11008.npz" if key == "had172" else "had_n14336.npz"))
    np.savez(tmp_path / "tables.npz", **{key: g["hadK"]})
    monkeypatch.setenv("GQ_HADAMARD_TABLES", str(tmp_path / "tables.npz"))
    qtip._tables = None
    gm.transformer_configs["qtip-factor-test"] = dict(model_name="llama-qtip-factor-test", block_size=128, vocab_size=512, n_layer=2,
                                                      n_head=8, dim=1024, intermediate_size=inter, n_local_heads=8)
    try:
        m = load_model("qtip-factor-test", "cuda:0", "qtip", 2, random_init=True)
    finally:
        del gm.transformer_configs["qtip-factor-test"]
        qtip._tables = None
    with torch.device("cuda:0"):
        m.setup_caches(max_batch_size=1, max_seq_length=64)
    monkeypatch.setenv("GQ_QTIP_MLP_MID", mid)
    assert m._native_kind() == "qtip"
    names = [nm for nm, _ in m._native_state()["qtip_layers"][0]["d"]]
    assert ("gq_qtip_mlp_mid" in names) == (mid == "1" and inter == 11008), names  # (28672 = 28 * 1024: two transform launches)
    toks = [3, 77, 401]
    ref = []
    with torch.no_grad():
        for p, t in enumerate(toks):
            ref.append(m(torch.tensor([[t]], dtype=torch.int32, device="cuda:0"), torch.tensor([p], dtype=torch.int32, device="cuda:0"))
                       .float().reshape(-1).clone())
        for b in m.layers:
            b.attention.kv_cache.k_cache.zero_()
            b.attention.kv_cache.v_cache.zero_()
        for p, t in enumerate(toks):
            got = m.decode_native(torch.tensor([t], dtype=torch.int32, device="cuda:0"),
                                  torch.tensor([p], dtype=torch.int32, device="cuda:0")).float().reshape(-1)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(got).all())
            err = float(torch.abs(got - ref[p]).max()) / (float(torch.abs(ref[p]).max()) + 1e-9)
            assert err < 2e-2, err


def test_qtip_native_decode_with_factor_hidden_size(tmp_path, monkeypatch):
    """hidden size AND MLP width with Hadamard factors (the Llama-2-13b pattern: 5120 = 20 * 256, 13824 = 108 * 128; here
    2560 = 20 * 128 and 6912 = 108 * 64): every linear runs transform -> bare matvec -> transform, with the factor tables
    shipped in guidedquant_amd/data/hadamard_factors.npz (matmul_had.py:13-67 order: 20 is the first factor of 2560)."""
    from guidedquant_amd import model as gm, qtip
    from guidedquant_amd.generate import load_model
    monkeypatch.delenv("GQ_HADAMARD_TABLES", raising=False)
    qtip._tables = None
    assert qtip.get_hadK(2560)[1] == 20 and qtip.get_hadK(6912)[1] == 108
    gm.transformer_configs["qtip-factor13-test"] = dict(model_name="llama-qtip-factor13-test", block_size=128, vocab_size=512, n_layer=2,
                                                        n_head=20, dim=2560, intermediate_size=6912, n_local_heads=20)
    try:
        m = load_model("qtip-factor13-test", "cuda:0", "qtip", 2, random_init=True)
    finally:
        del gm.transformer_configs["qtip-factor13-test"]
        qtip._tables = None
    with torch.device("cuda:0"):
        m.setup_caches(max_batch_size=1, max_seq_length=64)
    assert m._native_kind() == "qtip"
    toks = [9, 100]
    ref = []
    with torch.no_grad():
        for p, t in enumerate(toks):
            ref.append(m(torch.tensor([[t]], dtype=torch.int32, device="cuda:0"), torch.tensor([p], dtype=torch.int32, device="cuda:0"))
                       .float().reshape(-1).clone())
        for b in m.layers:
            b.attention.kv_cache.k_cache.zero_()
            b.attention.kv_cache.v_cache.zero_()
        for p, t in enumerate(toks):
            got = m.decode_native(torch.tensor([t], dtype=torch.int32, device="cuda:0"),
                                  torch.tensor([p], dtype=torch.int32, device="cuda:0")).float().reshape(-1)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(got).all())
            err = float(torch.abs(got - ref[p]).max()) / (float(torch.abs(ref[p]).max()) + 1e-9)
            assert err < 2e-2, err


# ----------------------------------------------------------------------------- band engine corner cases (round 2 rebuild)
@pytest.mark.parametrize("R", [2, 3, 4])
@pytest.mark.parametrize("M,K", [(64, 160), (96, 2080), (32, 96), (22016, 512), (9024, 1056)])
def test_matvec_ragged_chunks_and_many_items(oracle, R, M, K):
    """K / 32 not a multiple of the engine's load chunk (4 tile blocks at R = 2, 2 else): the last chunk of a band is partly
    valid and runs into the next band's bytes (or past the tensor: buffer bounds); more bands than blocks: every block walks
    several items with one codebook fill and one activation copy"""
    g = torch.Generator(device="cpu").manual_seed(R * 7919 + M + K)
    compressed = torch.randint(torch.iinfo(torch.int32).min, torch.iinfo(torch.int32).max, (R * M * K // 32, ), dtype=torch.int32, generator=g).numpy()
    tlut = torch.clamp(torch.randn(512, 2, generator=g) / 16, -1, 1).to(torch.float16).numpy()
    x = torch.clamp(torch.randn(K, 1, generator=g) / 16, -1, 1).to(torch.float16).numpy()
    got = _run(compressed, tlut, x, M, K, R)
    _check(got, compressed, tlut, x, M, K, R, oracle)


@pytest.mark.parametrize("R", [2, 3, 4])
@pytest.mark.parametrize("Ms,K", [([4096, 1024, 1024], 4096), ([1024], 11008), ([2048, 2048], 1056), ([352], 160)])
def test_pretransformed_linear_in_all_splits(oracle, R, Ms, K):
    """gq_qtip_linear_in (GQ_QPRO_PRETRANSFORMED = the bare matvec, any K % 32 == 0) with 1..4 K ranges per band and up to three
    linears in one launch (each block serves one linear): the parts, added in ascending order, equal the oracle's product"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(R * 31 + K + sum(Ms))
    comp = [torch.randint(torch.iinfo(torch.int32).min, torch.iinfo(torch.int32).max, (R * M * K // 32, ), dtype=torch.int32, generator=g) for M in Ms]
    tlut = torch.clamp(torch.randn(512, 2, generator=g) / 16, -1, 1).to(torch.float16)
    x = torch.clamp(torch.randn(K, generator=g) / 16, -1, 1).to(torch.float16)
    cd = [c.to(d) for c in comp]
    td, xd = tlut.reshape(-1).to(d), x.to(d)
    for ks in (1, 2, 3, 4):
        if ks > K // 32:
            continue
        y32 = [torch.full((ks * M,), float("nan"), dtype=torch.float32, device=d) for M in Ms]
        ain = (_lib.GqQtipIn * len(Ms))(*[_lib.GqQtipIn(cd[i].data_ptr(), None, td.data_ptr(), y32[i].data_ptr(), M) for i, M in enumerate(Ms)])
        rc = L.gq_qtip_linear_in(xd.data_ptr(), None, None, 0.0, 3, K, R, len(Ms), ain, 0, None, ks, None)
        nch = -(-(K // 32) // (4 if R == 2 else 2))
        if -(-nch // ks) * (ks - 1) >= nch:  # an empty K range: rejected, not silently mis-split
            assert rc != 0
            continue
        _lib.check(rc, "linear_in")
        torch.cuda.synchronize()
        for i, M in enumerate(Ms):
            parts = y32[i].reshape(ks, M).cpu().numpy()
            assert np.isfinite(parts).all()
            got = parts[0].copy()
            for p in range(1, ks):
                got += parts[p]
            _check(got, comp[i].numpy(), tlut.numpy(), x.numpy().reshape(K, 1), M, K, R, oracle)


def test_plan_ksplit_is_a_valid_split():
    from guidedquant_amd import _lib
    import ctypes
    L = _lib.lib()
    for Ms, K in (([4096], 4096), ([4096, 4096, 4096], 4096), ([11008, 11008], 4096), ([4096], 11008), ([64], 128), ([32], 32)):
        ks = L.gq_qtip_plan_ksplit(len(Ms), (ctypes.c_uint32 * len(Ms))(*Ms), K, 4)
        assert 1 <= ks <= 4
        assert ks == 1 or (K // 32) // 4 // ks >= 16  # every range keeps 16 waves busy
    # one linear with 128 bands on a 256-unit chip: two ranges
    assert L.gq_qtip_plan_ksplit(1, (ctypes.c_uint32 * 1)(4096), 4096, 4) == 2


@pytest.mark.parametrize("R", [2, 3])
@pytest.mark.parametrize("Ms,K,pro,ks", [([4096, 1024, 1024], 4096, 1, 1), ([4096], 4096, 0, 2), ([2048], 8192, 2, 2), ([256], 128, 0, 1), ([4096], 2048, 0, 4)])
def test_one_launch_linear_equals_two_launches(R, Ms, K, pro, ks):
    """gq_qtip_linear (the block that finishes a linear last runs its transform-out) == gq_qtip_linear_in + gq_qtip_linear_out,
    bit for bit, also with a residual and split K; repeated launches re-use the counters (the finishing block resets them)"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    mods = [_rand_qlinear(K, M, R, seed=100 * R + i + M + K) for i, M in enumerate(Ms)]
    g = torch.Generator(device="cpu").manual_seed(K + sum(Ms))
    x = (torch.randn(K, generator=g) * 0.7).half().to(d)
    x2 = torch.randn(K, generator=g).half().to(d)
    normw = (1 + 0.2 * torch.randn(K, generator=g)).half().to(d)
    resid = torch.randn(Ms[0], generator=g).half().to(d) if len(Ms) == 1 else None
    su = [m.SU.float().contiguous() for m in mods]
    sv = [(m.SV.float() * 32.0).contiguous() for m in mods]

    def run(one):
        y32 = [torch.full((ks * M,), float("nan"), dtype=torch.float32, device=d) for M in Ms]
        outs = [torch.full((M,), float("nan"), dtype=torch.float16, device=d) for M in Ms]
        ain = (_lib.GqQtipIn * len(Ms))(*[_lib.GqQtipIn(m.trellis.data_ptr(), su[i].data_ptr(), m.tlut.data_ptr(), y32[i].data_ptr(), Ms[i])
                                           for i, m in enumerate(mods)])
        aout = (_lib.GqQtipOut * len(Ms))(*[_lib.GqQtipOut(y32[i].data_ptr(), sv[i].data_ptr(), resid.data_ptr() if resid is not None else None,
                                                           outs[i].data_ptr(), Ms[i], ks) for i in range(len(Ms))])
        args = (x.data_ptr(), x2.data_ptr(), normw.data_ptr(), 1e-5, pro, K, R, len(Ms), ain)
        if one:
            ctr = torch.zeros(4, dtype=torch.int32, device=d)
            for _ in range(3):  # the counters come back to zero after every launch
                _lib.check(L.gq_qtip_linear(*args, aout, ks, ctr.data_ptr(), None), "one launch")
            torch.cuda.synchronize()
            assert int(ctr.abs().sum()) == 0
        else:
            _lib.check(L.gq_qtip_linear_in(*args, 0, None, ks, None), "in")
            _lib.check(L.gq_qtip_linear_out(len(Ms), aout, None), "out")
            torch.cuda.synchronize()
        return outs

    two, one = run(False), run(True)
    for a, b in zip(two, one):
        assert bool(torch.isfinite(a.float()).all())
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
