"""GPU: the HIP prompt pass (`Transformer.prefill_native`, csrc/prefill.hip) against the module forward it replaces
(`Transformer.forward` with seq_len > 1 = inference/model.py:206-266 semantics).

  * the three row kernels against the tensor expressions of model.py on the same fp16 inputs: RoPE + cache write bit for bit,
    RMSNorm / silu * up bit for bit up to the fp32 summation order / one ulp of exp (<= 0.1 % of the elements, one or two fp16 ulps);
  * the whole pass on a small model and on the 2-layer model of the real 8B widths: logits of every prompt position within
    1e-2 of max|logit| (the linears run the same kernels in both; the attention sums differ in order), K / V caches equal up
    to that noise, the same greedy continuation from generate() either way.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _lib():
    from guidedquant_amd import _lib as L
    return L


def _ulp_close(a, b, frac=1e-3, ulps=1):
    """equal except for at most `frac` of the elements, which differ by `ulps` fp16 ulps at most"""
    a16, b16 = a.view(torch.int16).int(), b.view(torch.int16).int()
    diff = (a16 - b16).abs()
    assert int(diff.max()) <= ulps, int(diff.max())
    assert float((diff != 0).float().mean()) <= frac, float((diff != 0).float().mean())


@pytest.mark.parametrize("S,D", [(1, 4096), (7, 512), (130, 4096), (33, 14336), (5, 16384)])
def test_rmsnorm_rows(S, D):
    from guidedquant_amd.model import RMSNorm
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(S + D)
    x = (torch.randn(S, D, device=d, generator=g) * torch.rand(S, 1, device=d, generator=g) * 8).half()
    norm = RMSNorm(D, eps=1e-5).to(d).half()
    norm.weight.data.copy_((1 + 0.2 * torch.randn(D, device=d, generator=g)).half())
    want = norm(x)
    out = torch.empty_like(x)
    L = _lib()
    L.check(L.lib().gq_rmsnorm_rows(x.data_ptr(), None, norm.weight.data_ptr(), out.data_ptr(), S, D, norm.eps, None), "gq_rmsnorm_rows")
    torch.cuda.synchronize()
    # the fp32 sum of squares is added in another order: where the normalised value flips by one fp16 ulp, its product with
    # a weight of up to 1.6 moves by up to two
    _ulp_close(out, want, ulps=2)
    # with the residual add in front: x + delta (fp16) written back, then the same norm
    delta = torch.randn(S, D, device=d, generator=g).half()
    xs = x + delta
    want2 = norm(xs)
    x2 = x.clone()
    L.check(L.lib().gq_rmsnorm_rows(x2.data_ptr(), delta.data_ptr(), norm.weight.data_ptr(), out.data_ptr(), S, D, norm.eps, None), "gq_rmsnorm_rows")
    torch.cuda.synchronize()
    assert torch.equal(x2.view(torch.int16), xs.view(torch.int16))
    _ulp_close(out, want2, ulps=2)


@pytest.mark.parametrize("S,H,Hkv,hd,start", [(5, 8, 2, 64, 0), (130, 32, 8, 128, 0), (17, 4, 4, 128, 9)])
def test_rope_cache_rows(S, H, Hkv, hd, start):
    from guidedquant_amd.model import apply_rotary_pos_emb, rope_tables
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(S + H)
    max_seq = start + S + 3
    cos, sin = rope_tables(hd, max_seq, 500000.0, d)
    qkv = torch.randn(S, (H + 2 * Hkv) * hd, device=d, generator=g).half()
    pos = torch.arange(start, start + S, dtype=torch.int32, device=d)
    qr, kr, vr = qkv.view(1, S, -1).split([H * hd, Hkv * hd, Hkv * hd], dim=-1)
    qr = qr.view(1, S, H, hd).transpose(1, 2)
    kr = kr.view(1, S, Hkv, hd).transpose(1, 2)
    vr = vr.view(1, S, Hkv, hd).transpose(1, 2)
    qw, kw = apply_rotary_pos_emb(qr, kr, cos[pos.long()].unsqueeze(0), sin[pos.long()].unsqueeze(0))
    kc = torch.full((1, Hkv, max_seq, hd), 7.0, dtype=torch.float16, device=d)
    vc = torch.full((1, Hkv, max_seq, hd), 7.0, dtype=torch.float16, device=d)
    q = torch.empty(H, S, hd, dtype=torch.float16, device=d)
    L = _lib()
    L.check(L.lib().gq_rope_cache_rows(qkv.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), q.data_ptr(), kc.data_ptr(), vc.data_ptr(), S, H, Hkv,
                                       hd, max_seq, None), "gq_rope_cache_rows")
    torch.cuda.synchronize()
    assert torch.equal(q.view(torch.int16), qw[0].contiguous().view(torch.int16))
    assert torch.equal(kc[0, :, start:start + S].contiguous().view(torch.int16), kw[0].contiguous().view(torch.int16))
    assert torch.equal(vc[0, :, start:start + S].contiguous().view(torch.int16), vr[0].contiguous().view(torch.int16))
    # nothing outside the written positions was touched
    assert bool((kc[0, :, :start] == 7.0).all()) and bool((kc[0, :, start + S:] == 7.0).all()) and bool((vc[0, :, start + S:] == 7.0).all())


@pytest.mark.parametrize("paired", [0, 1])
@pytest.mark.parametrize("S,inter", [(3, 1024), (130, 14336), (9, 11008)])
def test_silu_mul_rows(S, inter, paired):
    import torch.nn.functional as F
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(S + inter + paired)
    y = (torch.randn(S, 2 * inter, device=d, generator=g) * 3).half()
    if paired:
        gate, up = y[:, 0::2], y[:, 1::2]
    else:
        gate, up = y[:, :inter], y[:, inter:]
    want = (F.silu(gate) * up).contiguous()
    out = torch.empty(S, inter, dtype=torch.float16, device=d)
    L = _lib()
    L.check(L.lib().gq_silu_mul_rows(y.data_ptr(), out.data_ptr(), S, inter, paired, None), "gq_silu_mul_rows")
    torch.cuda.synchronize()
    _ulp_close(out, want)


def _tiny(bits):
    from test_decode_gpu import _tiny_model
    return _tiny_model(bits, hd=64)


def _wide(bits):
    from test_decode_default_gpu import _model
    return _model(bits, n_layer=2)


@pytest.mark.parametrize("which,bits,S", [("tiny", 2, 9), ("tiny", 4, 70), ("wide", 2, 130), ("wide", 3, 40)])
def test_prefill_native_matches_module_forward(which, bits, S):
    d = torch.device("cuda:0")
    m = _tiny(bits) if which == "tiny" else _wide(bits)
    m.setup_caches(1, S + 8)
    assert m.native_ready()  # (pairs the gate / up rows in place: the paired read path of silu * up)
    g = torch.Generator(device=d).manual_seed(S)
    idx = torch.randint(0, m.config.vocab_size, (1, S), dtype=torch.int32, device=d, generator=g)
    pos = torch.arange(S, dtype=torch.int32, device=d)
    assert m.prefill_ready(idx)
    with torch.no_grad():
        want = m(idx, pos).float()
        kw = [b.attention.kv_cache.k_cache.clone() for b in m.layers]
        vw = [b.attention.kv_cache.v_cache.clone() for b in m.layers]
        for b in m.layers:
            b.attention.kv_cache.k_cache.zero_()
            b.attention.kv_cache.v_cache.zero_()
        got = m.prefill_native(idx, pos, start=0, last_only=False).float()
        last = m.prefill_native(idx, pos, start=0, last_only=True).float()
    torch.cuda.synchronize()
    assert got.shape == want.shape and last.shape == (1, 1, m.config.vocab_size)
    assert torch.isfinite(got).all()
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-2 * scale, ((got - want).abs().max().item(), scale)
    assert ((got - want).norm() / want.norm()).item() <= 3e-3
    assert (last[0, 0] - got[0, -1]).abs().max().item() <= 2e-3 * scale
    for i, b in enumerate(m.layers):
        ks = kw[i].float().abs().max().item()
        assert (b.attention.kv_cache.k_cache.float() - kw[i].float()).abs().max().item() <= 1e-2 * ks
        assert (b.attention.kv_cache.v_cache.float() - vw[i].float()).abs().max().item() <= 1e-2 * vw[i].float().abs().max().item()
        assert bool((b.attention.kv_cache.k_cache[:, :, S:] == 0).all())  # only the prompt's positions were written


def test_prefill_native_at_an_offset_matches_module_forward():
    """a second chunk of a prompt (positions start .. start + S) attends to the cached first chunk: explicit mask path"""
    d = torch.device("cuda:0")
    m = _tiny(2)
    S0, S1 = 12, 20
    m.setup_caches(1, S0 + S1 + 4)
    assert m.native_ready()
    g = torch.Generator(device=d).manual_seed(3)
    idx = torch.randint(0, m.config.vocab_size, (1, S0 + S1), dtype=torch.int32, device=d, generator=g)
    with torch.no_grad():
        want = m(idx, torch.arange(S0 + S1, dtype=torch.int32, device=d)).float()[:, S0:]
        for b in m.layers:
            b.attention.kv_cache.k_cache.zero_()
            b.attention.kv_cache.v_cache.zero_()
        m.prefill_native(idx[:, :S0], torch.arange(S0, dtype=torch.int32, device=d), start=0)
        got = m.prefill_native(idx[:, S0:], torch.arange(S0, S0 + S1, dtype=torch.int32, device=d), start=S0, last_only=False).float()
    assert (got - want).abs().max().item() <= 1e-2 * want.abs().max().item()


def test_generate_takes_the_native_prompt_pass_and_continues_the_same():
    from guidedquant_amd.generate import generate
    d = torch.device("cuda:0")
    m = _tiny(2)
    g = torch.Generator(device=d).manual_seed(11)
    prompt = torch.randint(0, m.config.vocab_size, (24,), dtype=torch.int32, device=d, generator=g)
    outs = {}
    try:
        for mode in ("1", "0"):
            os.environ["GQ_PREFILL_NATIVE"] = mode
            torch.manual_seed(0)
            outs[mode] = generate(m, prompt, 16, use_graph=False, temperature=0.0, top_k=32)
    finally:
        os.environ.pop("GQ_PREFILL_NATIVE", None)
    assert outs["1"].shape == (1, 24 + 16)
    assert torch.equal(outs["1"][:, :24], outs["0"][:, :24])
    # greedy continuation: identical unless a logit tie sits inside the summation-order noise (not with this seed)
    assert torch.equal(outs["1"], outs["0"])
