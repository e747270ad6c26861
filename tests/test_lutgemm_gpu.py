"""GPU parity of the LUT-GEMM (BCQ) GEMV against the CPU restatement of lutgemm.cu (parity unpinned: the reference
holds no test or producer for this op).  The HIP kernel walks the tiles in ascending order with the reference's fp16
operation sequence, so it is BIT-IDENTICAL to the oracle's ascending-order restatement."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ws", ["1", "0"])
@pytest.mark.parametrize("bits", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K,g", [(64, 256, 128), (512, 4096, 4096), (300, 1024, 32), (4096, 4096, 128)])
def test_lutgemm_bit_exact(oracle, bits, N, K, g, ws, monkeypatch):
    """ws = 1: tables built once into the scratch buffer + GEMV (gq_lutgemm_gemv_ws); ws = 0: the single kernel that
    rebuilds them per block (gq_lutgemm_gemv).  Both bit-identical to the oracle."""
    from guidedquant_amd.LUTGEMMLinear import LUTGEMMLinear
    monkeypatch.setenv("GQ_LUTGEMM_WS", ws)
    d = torch.device("cuda:0")
    rng = np.random.default_rng(bits * 100 + N + K)
    q = rng.integers(-2**31, 2**31, (K // 32, bits, N), dtype=np.int64).astype(np.int32)
    alpha = (rng.random((K // g, bits, N)) * 0.01).astype(np.float16)
    qb = rng.normal(0, 0.01, (K // g, N)).astype(np.float16)
    x = rng.normal(0, 1, K).astype(np.float16)
    lin = LUTGEMMLinear(K, N, bits, g, device=d)
    assert lin.qweight.shape == (K // 32, bits, N) and lin.alpha.shape == (K // g, bits, N) and lin.q_bias.shape == (K // g, N)
    lin.load_state_dict({"qweight": torch.from_numpy(q), "alpha": torch.from_numpy(alpha), "q_bias": torch.from_numpy(qb)})
    y = lin(torch.from_numpy(x.reshape(1, 1, K)).to(d))
    assert y is lin.output
    got = y.cpu().numpy().reshape(N)
    want = oracle.lutgemm_f16(x, q, alpha, qb, bits, g)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    y64 = oracle.lutgemm_f64(x, q, alpha, qb, bits, g)
    scale = np.abs(y64).max() + 1e-6
    assert np.abs(got.astype(np.float64) - y64).max() <= 2e-2 * scale


def test_lutgemm_validation():
    from guidedquant_amd import ap_gemv
    d = torch.device("cuda:0")
    K, N, b = 256, 64, 2
    x = torch.zeros(1, 1, K, dtype=torch.float16, device=d)
    out = torch.zeros(1, 1, N, dtype=torch.float16, device=d)
    q = torch.zeros(K // 32, b, N, dtype=torch.int32, device=d)
    al = torch.zeros(1, b, N, dtype=torch.float16, device=d)
    qb = torch.zeros(1, N, dtype=torch.float16, device=d)
    ap_gemv.lutgemm_gemv(x, out, q, al, qb, b, K)
    with pytest.raises(RuntimeError, match="Bitwidth must be between 1 and 8"):
        ap_gemv.lutgemm_gemv(x, out, q, al, qb, 9, K)
    with pytest.raises(RuntimeError, match="q_weight tensor must be of shape"):
        ap_gemv.lutgemm_gemv(x, out, q[:, :1].contiguous(), al, qb, b, K)
    with pytest.raises(RuntimeError, match="alpha tensor must be of shape"):
        ap_gemv.lutgemm_gemv(x, out, q, al[:, :1].contiguous(), qb, b, K)
    with pytest.raises(RuntimeError, match="Batch size must be 1"):
        ap_gemv.lutgemm_gemv(torch.zeros(2, 1, K, dtype=torch.float16, device=d), out, q, al, qb, b, K)
