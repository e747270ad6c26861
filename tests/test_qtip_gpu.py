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
