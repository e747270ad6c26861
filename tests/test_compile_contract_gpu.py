"""GPU: torch.library.opcheck of the registered operators on the device and a fullgraph torch.compile of APLinear.forward
(reference: inference/plugin.py:7-26, inference/generate.py --compile); CPU counterpart: test_compile_contract_cpu.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_opcheck_registered_ops_on_the_device():
    from torch.library import opcheck
    from guidedquant_amd import pack, plugin, qtip  # noqa: F401
    d = torch.device("cuda:0")
    N, K, bits = 256, 512, 3
    q, lut = pack.random_quantized_linear(N, K, bits, 3)
    q, lut = torch.from_numpy(q).to(d), torch.from_numpy(lut).to(d)
    x = torch.randn(1, 1, K, device=d).half()
    out = torch.zeros(1, 1, N, dtype=torch.float16, device=d)
    for name, res in (("anyprec_gemv", opcheck(torch.ops.plugin.anyprec_gemv.default, (x, q, lut, out, bits))),):
        assert all(v == "SUCCESS" for v in res.values()), (name, res)
    g = 128
    qw = torch.randint(-2**31, 2**31 - 1, (K // 32, bits, N), dtype=torch.int32, device=d)
    al = (torch.rand(K // g, bits, N, device=d) * 0.01).half()
    qb = (torch.randn(K // g, N, device=d) * 0.01).half()
    res = opcheck(torch.ops.plugin.lutgemm_gemv.default, (x, torch.zeros_like(out), qw, al, qb, bits, g))
    assert all(v == "SUCCESS" for v in res.values()), res
    res = opcheck(torch.ops.hadamard.hadamard.default, (torch.randn(4, 256, device=d).half(), 0.0625),
                  test_utils=("test_schema", "test_faketensor"))
    assert all(v == "SUCCESS" for v in res.values()), res
    op = qtip.quip_lib_op(256, 256, 2)
    comp = torch.randint(-2**15, 2**15 - 1, (256 * 256 * 2 // 16,), dtype=torch.int16, device=d)
    cb = torch.randn(1 << 9, 2, device=d).half()
    res = opcheck(op.default, (comp, torch.randn(1, 256, device=d).half(), cb), test_utils=("test_schema", "test_faketensor"))
    assert all(v == "SUCCESS" for v in res.values()), res


def test_aplinear_forward_compiles_fullgraph_on_the_device():
    from guidedquant_amd import pack
    from guidedquant_amd.APLinear import APLinear
    d = torch.device("cuda:0")
    m = APLinear(512, 256, 2, device=d)
    q, lut = pack.random_quantized_linear(256, 512, 2, 9)
    m.qweight.copy_(torch.from_numpy(q))
    m.lut.copy_(torch.from_numpy(lut))
    x = torch.randn(1, 1, 512, device=d).half()
    want = m(x).clone()
    torch._dynamo.reset()
    f = torch.compile(m.forward, fullgraph=True, backend="eager")
    assert torch.equal(f(x), want)
