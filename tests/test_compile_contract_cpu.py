"""The operator boundary's compile contract (the ops "must be traceable by torch.compile(fullgraph=True)",
reference: inference/plugin.py:7-26 fake registrations, inference/generate.py --compile).  CPU box: `backend="eager"` needs no
code generator; plugin::anyprec_gemv runs on the CPU twin of the kernel (csrc/cpu_twins.cpp), the GPU-only ops are traced with
fake tensors only (schema + fake implementation); tests/test_compile_contract_gpu.py runs the same checks on the device."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _aplinear(N=64, K=256, bits=3):
    from guidedquant_amd import pack
    from guidedquant_amd.APLinear import APLinear
    m = APLinear(K, N, bits, device="cpu")
    q, lut = pack.random_quantized_linear(N, K, bits, 5)
    m.qweight.copy_(torch.from_numpy(q))
    m.lut.copy_(torch.from_numpy(lut))
    return m


def test_opcheck_anyprec_gemv_on_the_cpu_twin():
    from torch.library import opcheck
    from guidedquant_amd import plugin  # noqa: F401  (registers the ops)
    m = _aplinear()
    x = torch.randn(1, 1, m.in_features).half()
    out = torch.zeros(1, 1, m.out_features, dtype=torch.float16)
    res = opcheck(torch.ops.plugin.anyprec_gemv.default, (x, m.qweight, m.lut, out, m.bitwidth))
    assert all(v == "SUCCESS" for v in res.values()), res


def test_aplinear_forward_compiles_fullgraph():
    """one decode row through torch.compile(fullgraph=True): the mutating custom op is traced through its fake, the compiled
    function returns what eager returns (the persistent output buffer, APLinear.py:52-60)"""
    m = _aplinear()
    x = torch.randn(1, 1, m.in_features).half()
    want = m(x).clone()
    torch._dynamo.reset()
    f = torch.compile(m.forward, fullgraph=True, backend="eager")
    got = f(x)
    assert torch.equal(got, want)
    x2 = torch.randn(1, 1, m.in_features).half()
    assert torch.equal(f(x2), m(x2))  # (second call: no recompilation error, buffer semantics kept)


def _fake_trace(fn, *shapes_dtypes):
    """run fn on fake tensors (no kernel executes): the schema and the registered fake implementation are exercised"""
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode(allow_non_fake_inputs=False) as mode:
        args = [torch.empty(s, dtype=d, device="cuda") if isinstance(s, tuple) else s for s, d in shapes_dtypes]
        return fn(*args)


def test_gpu_only_ops_trace_with_fake_tensors():
    from guidedquant_amd import plugin, qtip  # noqa: F401
    K, N, bits, g = 256, 64, 3, 128
    r = _fake_trace(lambda x, out, qw, al, qb: torch.ops.plugin.lutgemm_gemv(x, out, qw, al, qb, bits, g),
                    ((1, 1, K), torch.float16), ((1, 1, N), torch.float16), ((K // 32, bits, N), torch.int32),
                    ((K // g, bits, N), torch.float16), ((K // g, N), torch.float16))
    assert r is None
    op = qtip.quip_lib_op(256, 256, 2)
    y = _fake_trace(lambda c, x, cb: op(c, x, cb), ((256 * 256 * 2 // 16,), torch.int16), ((1, 256), torch.float16), ((1 << 9, 2), torch.float16))
    assert tuple(y.shape) == (1, 256) and y.dtype == torch.float32
    h = _fake_trace(lambda x: torch.ops.hadamard.hadamard(x, 0.5), ((4, 256), torch.float16))
    assert tuple(h.shape) == (4, 256) and h.dtype == torch.float16
