"""CPU: the product's host twins of the Any-Precision ops (`gq_anyprec_gemv_cpu` / `gq_anyprec_dequant_cpu`,
guidedquant_amd/csrc/cpu_twins.cpp) against the oracle and the reference-generated goldens, and BASELINE.json configs[0]
-- "Llama-3.2-1B-Instruct 2-bit, bs=1, CPU reference APLinear path via generate.py (plumbing, no GPU)" -- end to end on a
shrunken layer count.  The twins are product code: checked AGAINST oracle/, never routed through it."""
import numpy as np
import pytest

from conftest import golden_files

torch = pytest.importorskip("torch")


def _gemv_cpu(x, q, lut, bits, M=1):
    from guidedquant_amd import ap_gemv
    K, N = q.shape[2] * 32, q.shape[1]
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float16).reshape(M, 1, K))
    out = torch.full((M, 1, N), float("nan"), dtype=torch.float16)
    ap_gemv.anyprec_gemv(xt, out, torch.from_numpy(np.ascontiguousarray(q)), torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float16)), bits)
    return out.numpy().reshape(M, N)


def _envelope(got, x, q, lut, bits, oracle):
    """the documented tolerance of the twin: |out - exact| <= 2^-11 |exact| + 1e-5 sum|w||x|"""
    y64 = oracle.ap_gemv_f64(x, q, lut, bits)
    W = np.abs(oracle.ap_dequant(q, lut, bits).astype(np.float64))
    scale = np.abs(np.asarray(x, dtype=np.float64).reshape(y64.shape[0], -1)) @ W.T
    assert (np.abs(got.astype(np.float64) - y64) <= 2.0**-11 * 1.001 * np.abs(y64) + 1e-5 * scale + 1e-7).all()


@pytest.mark.parametrize("path", golden_files("ap_b"))
def test_cpu_twin_goldens(oracle, path):
    """every reference-generated fixture (bits 2..8, tail chunks, K = 96 .. 14336): dequant bit-exact against
    `_dequantize_weight`, GEMV inside the envelope of the reference-generated exact product"""
    from guidedquant_amd import ap_gemv
    g = np.load(path)
    bits = int(g["bits"])
    W = ap_gemv.anyprec_dequant(torch.from_numpy(g["qweight"]), torch.from_numpy(g["lut"]), bits)
    assert W.dtype == torch.float16 and np.array_equal(W.numpy().view(np.uint16), g["W"].view(np.uint16))
    got = _gemv_cpu(g["x"], g["qweight"], g["lut"], bits)
    _envelope(got, g["x"], g["qweight"], g["lut"], bits, oracle)
    scale = np.abs(g["W"].astype(np.float64)) @ np.abs(g["x"].astype(np.float64))
    assert (np.abs(got[0].astype(np.float64) - g["y64"]) <= 2.0**-11 * 1.001 * np.abs(g["y64"]) + 1e-5 * scale + 1e-7).all()


@pytest.mark.parametrize("bits", [2, 3, 4, 5, 8])
@pytest.mark.parametrize("N,K,M", [(64, 4096, 1), (33, 1152, 1), (8, 96, 1), (24, 2048, 3), (5, 11008, 1)])
def test_cpu_twin_random(oracle, bits, N, K, M):
    rng = np.random.default_rng(bits * 1009 + N + K)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = (rng.normal(0, 1, (N, 1 << bits)) * 10.0**rng.integers(-4, 1, (N, 1))).astype(np.float16)
    X = (rng.normal(0, 1, (M, K)) * np.where(rng.random((M, K)) < 0.02, 40.0, 1.0)).astype(np.float16)
    got = _gemv_cpu(X, q, lut, bits, M=M)
    _envelope(got, X, q, lut, bits, oracle)
    # fp16 special values survive the scalar conversions: subnormal / zero activations, an all-zero row
    X2 = X.copy()
    X2[:, ::3] = np.float16(6e-8)
    X2[:, 1::3] = 0
    _envelope(_gemv_cpu(X2, q, lut, bits, M=M), X2, q, lut, bits, oracle)
    assert (_gemv_cpu(np.zeros((M, K), np.float16), q, lut, bits, M=M) == 0).all()


def test_cpu_twin_any_precision_parent_tensor(oracle):
    """a 4-bit parent tensor served at 2 / 3 / 4 bits (first b planes; plane stride N*K/32 words, anyprec.cu:446)"""
    N, K = 16, 2048
    rng = np.random.default_rng(4)
    codes = rng.integers(0, 16, (N, K), dtype=np.uint8)
    q4 = oracle.ap_pack(codes, 4)
    x = rng.normal(0, 1, K).astype(np.float16)
    for b in (2, 3, 4):
        lut = rng.normal(0, 0.05, (N, 1 << b)).astype(np.float16)
        _envelope(_gemv_cpu(x, q4, lut, b), x, oracle.ap_pack(codes >> (4 - b), b), lut, b, oracle)


def test_aplinear_module_on_cpu(oracle):
    """APLinear(device='cpu'): decode row -> GEMV twin into the persistent output, prefill rows -> dequant twin + matmul
    (inference/APLinear.py:35-60), through the plugin::anyprec_gemv op"""
    from guidedquant_amd.APLinear import APLinear
    bits, N, K = 3, 96, 1024
    rng = np.random.default_rng(8)
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q, lut = oracle.ap_pack(codes, bits), rng.normal(0, 0.05, (N, 1 << bits)).astype(np.float16)
    lin = APLinear(K, N, bits, device="cpu")
    lin.load_state_dict({"qweight": torch.from_numpy(q), "lut": torch.from_numpy(lut)})
    x = rng.normal(0, 1, (1, 1, K)).astype(np.float16)
    y = lin(torch.from_numpy(x))
    assert y is lin.output
    _envelope(y.numpy().reshape(1, N), x.reshape(K), q, lut, bits, oracle)
    xs = torch.from_numpy(rng.normal(0, 1, (1, 4, K)).astype(np.float16))
    ys = lin(xs)
    ref = xs.float().numpy()[0] @ oracle.ap_dequant(q, lut, bits).astype(np.float32).T
    np.testing.assert_allclose(ys.float().numpy()[0], ref, rtol=2e-2, atol=2e-2)


def test_config0_generate_on_cpu_llama_3_2_1b_shapes():
    """BASELINE.json configs[0]: the generate.py harness with --device cpu on the Llama-3.2-1B-Instruct geometry (dim 2048,
    MLP 8192, GQA 32/8, vocab 128256, llama3 rope scaling), 2-bit, random init -- 2 of the 16 layers to stay in seconds.
    Greedy tokens equal a float64 forward of the dequantised model wherever its top-1 margin is not a near-tie."""
    from guidedquant_amd import generate as G
    from guidedquant_amd.model import transformer_configs
    name = "meta-llama/Llama-3.2-1B-Instruct"
    saved = dict(transformer_configs[name])
    transformer_configs[name] = dict(saved, n_layer=2)
    try:
        torch.manual_seed(0)
        model = G.load_model(name, "cpu", "ap", 2, random_init=True)
    finally:
        transformer_configs[name] = saved
    assert model.config.dim == 2048 and model.config.rope_scaling["rope_type"] == "llama3"
    assert not model.native_ready()
    model.tok_embeddings.weight.data.mul_(25.0)
    model.output.weight.data.mul_(4.0)
    prompt = torch.tensor([128000], dtype=torch.int32)
    seq = G.generate(model, prompt, 6, use_graph=False, temperature=0.0, top_k=32)
    assert seq.shape == (1, 7) and int(seq[0, 0]) == 128000
    # independent float32 statement: dequantised weights in nn.Linear modules, same tokens teacher-forced
    from guidedquant_amd import ap_gemv
    from guidedquant_amd.model import Transformer
    ref = Transformer(torch.float32, model.config, fuse_linears=True)
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith(".qweight"):
            base = k[:-len(".qweight")]
            sd[base + ".weight"] = ap_gemv.anyprec_dequant(v, model.state_dict()[base + ".lut"], 2).float()
        elif not k.endswith(".lut") and "kv_cache" not in k:
            sd[k] = v.float()
    ref.load_state_dict(sd, strict=True)
    ref = ref.float().eval()
    ref.setup_caches(1, 8)
    agree = 0
    with torch.no_grad():
        for p in range(6):
            lg = ref(seq[:, p:p + 1].int(), torch.tensor([p], dtype=torch.int32)).view(-1)
            top2 = torch.topk(lg, 2).values
            if int(lg.argmax()) == int(seq[0, p + 1]):
                agree += 1
            else:
                assert float(top2[0] - top2[1]) < 2e-2 * float(lg.abs().max()), (p, top2)
    assert agree >= 4
