"""GPU: the fused HIP decode step (embedding, RMSNorm->wqkv, RoPE+KV+attention, wo+residual, RMSNorm->w1w3,
SiLU*up->w2+residual, final norm + lm_head) against the plain-PyTorch statement of the reference model
(guidedquant_amd.model.Transformer.forward = inference/model.py semantics) on the same synthetic weights.
The quantized GEMVs are bit-identical in both paths (exact mode); the tolerance covers fp32-vs-fp16 attention
arithmetic, reduction order of the norms and the dense lm_head accumulation order."""
import math

import numpy as np

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL = 2e-2  # relative to max|logit|: fp16 pipeline, different (fp32) internal arithmetic in attention / norms


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _exact_mode():
    from guidedquant_amd import _lib
    _lib.check(_lib.lib().gq_set_ap_mode(1), "gq_set_ap_mode")
    yield
    _lib.lib().gq_set_ap_mode(-1)


def _tiny_model(bits, hd=64):
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.generate import random_init_
    from guidedquant_amd.model import ModelArgs, Transformer
    d = _dev()
    n_head = 8
    cfg = ModelArgs(block_size=256, vocab_size=1024, n_layer=2, n_head=n_head, dim=n_head * hd, intermediate_size=1024,
                    n_local_heads=2, rope_base=500000, model_name="llama-test")
    m = Transformer(torch.float16, cfg, linear_class=APLinear, linear_kwargs=dict(bitwidth=bits, device=d))
    m = m.to(device=d, dtype=torch.float16)
    random_init_(m, seed=bits, lut_std=0.05)
    g = torch.Generator(device=d)
    g.manual_seed(1)
    for b in m.layers:
        b.input_layernorm.weight.data.copy_((1 + 0.1 * torch.randn(cfg.dim, device=d, generator=g)).half())
        b.post_attention_layernorm.weight.data.copy_((1 + 0.1 * torch.randn(cfg.dim, device=d, generator=g)).half())
    m.norm.weight.data.copy_((1 + 0.1 * torch.randn(cfg.dim, device=d, generator=g)).half())
    return m.eval()


@pytest.mark.parametrize("bits,hd", [(2, 64), (3, 64), (4, 128)])
def test_decode_native_matches_torch_forward(bits, hd):
    d = _dev()
    m = _tiny_model(bits, hd)
    m.setup_caches(1, 32)
    assert m.native_ready()
    toks = [5, 17, 900, 3, 3, 512, 44, 1023]
    ref_logits = []
    with torch.no_grad():
        for p, t in enumerate(toks):
            lg = m(torch.tensor([[t]], dtype=torch.int32, device=d), torch.tensor([p], dtype=torch.int32, device=d))
            ref_logits.append(lg.float().clone())
    ref_k = [b.attention.kv_cache.k_cache.clone() for b in m.layers]
    ref_v = [b.attention.kv_cache.v_cache.clone() for b in m.layers]
    for b in m.layers:
        b.attention.kv_cache.k_cache.zero_()
        b.attention.kv_cache.v_cache.zero_()
    with torch.no_grad():
        for p, t in enumerate(toks):
            lg = m.decode_native(torch.tensor([t], dtype=torch.int32, device=d), torch.tensor([p], dtype=torch.int32, device=d))
            torch.cuda.synchronize()
            a, r = lg.float().view(-1), ref_logits[p].view(-1)
            scale = r.abs().max().item()
            assert torch.isfinite(a).all()
            assert (a - r).abs().max().item() <= TOL * scale, (p, (a - r).abs().max().item(), scale)
    for i, b in enumerate(m.layers):
        n = len(toks)
        dk = (b.attention.kv_cache.k_cache[:, :, :n].float() - ref_k[i][:, :, :n].float()).abs().max().item()
        dv = (b.attention.kv_cache.v_cache[:, :, :n].float() - ref_v[i][:, :, :n].float()).abs().max().item()
        sk = ref_k[i][:, :, :n].float().abs().max().item()
        assert dk <= TOL * sk and dv <= TOL * sk, (i, dk, dv, sk)


@pytest.mark.parametrize("hd,H,Hkv", [(64, 8, 2), (128, 4, 4), (128, 32, 8)])
def test_attention_kernel(hd, H, Hkv):
    from guidedquant_amd import _lib
    from guidedquant_amd.model import apply_rotary_pos_emb, rope_tables
    d = _dev()
    L = _lib.lib()
    max_seq = 64
    g = torch.Generator(device=d)
    g.manual_seed(hd + H)
    cos, sin = rope_tables(hd, max_seq, 500000.0, d)
    kc = torch.zeros(1, Hkv, max_seq, hd, dtype=torch.float16, device=d)
    vc = torch.zeros_like(kc)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    out = torch.zeros(H * hd, dtype=torch.float16, device=d)
    for p in range(20):
        qkv = torch.randn((H + 2 * Hkv) * hd, device=d, generator=g).half()
        pos = torch.tensor([p], dtype=torch.int32, device=d)
        _lib.check(L.gq_attn_decode(qkv.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(),
                                    vc.data_ptr(), out.data_ptr(), H, Hkv, hd, max_seq, 1.0 / math.sqrt(hd),
                                    _lib.current_stream_ptr()), "attn")
        q, k, v = qkv.split([H * hd, Hkv * hd, Hkv * hd])
        q = q.view(1, 1, H, hd).transpose(1, 2)
        k = k.view(1, 1, Hkv, hd).transpose(1, 2)
        v = v.view(1, 1, Hkv, hd).transpose(1, 2)
        q, k = apply_rotary_pos_emb(q, k, cos[p:p + 1].unsqueeze(0), sin[p:p + 1].unsqueeze(0))
        kc_ref[:, :, p] = k[:, :, 0]
        vc_ref[:, :, p] = v[:, :, 0]
        kk = kc_ref[:, :, :p + 1].float().repeat_interleave(H // Hkv, dim=1)
        vv = vc_ref[:, :, :p + 1].float().repeat_interleave(H // Hkv, dim=1)
        att = torch.softmax((q.float() @ kk.transpose(-1, -2)) / math.sqrt(hd), dim=-1) @ vv
        ref = att.transpose(1, 2).reshape(-1)
        torch.cuda.synchronize()
        assert torch.equal(kc[:, :, :p + 1], kc_ref[:, :, :p + 1]), "rotated keys must be bit-identical (fp16 RoPE)"
        assert torch.equal(vc[:, :, :p + 1], vc_ref[:, :, :p + 1])
        assert (out.float() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N,K,norm", [(1000, 512, False), (128256, 4096, True), (3000, 2048, True)])
def test_dense_gemv(N, K, norm):
    from guidedquant_amd import _lib
    d = _dev()
    L = _lib.lib()
    g = torch.Generator(device=d)
    g.manual_seed(N)
    W = (torch.randn(N, K, device=d, generator=g) * 0.02).half()
    x = torch.randn(K, device=d, generator=g).half()
    nw = (1 + 0.1 * torch.randn(K, device=d, generator=g)).half()
    out = torch.zeros(N, dtype=torch.float16, device=d)
    _lib.check(L.gq_dense_gemv_f16(x.data_ptr(), W.data_ptr(), out.data_ptr(), N, K, nw.data_ptr() if norm else None, 1e-5,
                                   _lib.current_stream_ptr()), "dense")
    xr = x
    if norm:
        xf = x.float()
        xr = (xf * torch.rsqrt((xf * xf).mean() + 1e-5)).half() * nw
    ref = W.float() @ xr.float()
    torch.cuda.synchronize()
    assert (out.float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-3


def test_generate_with_graph_matches_eager():
    from guidedquant_amd.generate import generate
    d = _dev()
    m = _tiny_model(2)
    prompt = torch.tensor([1], dtype=torch.int32, device=d)
    torch.manual_seed(0)
    a = generate(m, prompt, 12, use_graph=True, temperature=0.0, top_k=32)
    torch.manual_seed(0)
    b = generate(m, prompt, 12, use_graph=False, temperature=0.0, top_k=32)
    assert a.shape == (1, 13)
    assert torch.equal(a, b)


def test_fused_sampler_matches_reference_distribution():
    """temperature -> 0: the exponential race picks the arg-max logit (as the reference's sample() does); at T = 1 the
    empirical distribution over many draws matches softmax(top-k logits)."""
    from guidedquant_amd import _lib
    d = _dev()
    L = _lib.lib()
    V = 128256
    g = torch.Generator(device=d)
    g.manual_seed(3)
    logits = (torch.randn(V, device=d, generator=g) * 2).half()
    counter = torch.zeros(1, dtype=torch.int32, device=d)
    wv = torch.zeros(128 * 32, dtype=torch.float32, device=d)
    wi = torch.zeros(128 * 32, dtype=torch.int32, device=d)
    nt = torch.zeros(1, dtype=torch.int32, device=d)
    tok = torch.zeros(1, dtype=torch.int32, device=d)
    pos = torch.zeros(1, dtype=torch.int32, device=d)

    def draw(T, k):
        _lib.check(L.gq_sample_topk(logits.data_ptr(), V, k, T, 77, counter.data_ptr(), wv.data_ptr(), wi.data_ptr(),
                                    tok.data_ptr(), pos.data_ptr(), nt.data_ptr(), _lib.current_stream_ptr()), "sample")
        return int(nt.item())

    assert draw(0.0, 32) == int(logits.float().argmax().item())
    assert int(tok.item()) == int(nt.item()) and int(pos.item()) == 1 and int(counter.item()) == 1
    # stage-1/2 top-k equals torch.topk
    top = torch.topk(logits.float(), 32)
    draws = [draw(1.0, 32) for _ in range(3000)]
    assert set(draws) <= set(top.indices.tolist())
    p = torch.softmax(top.values, dim=0).cpu().numpy()
    import numpy as np
    cnt = np.array([draws.count(int(i)) for i in top.indices.tolist()], dtype=np.float64) / len(draws)
    assert np.abs(cnt - p).max() < 0.04, (cnt, p)
    assert int(pos.item()) == 3001


def test_native_sampling_graph_advances_by_itself():
    from guidedquant_amd.generate import DecodeGraph
    d = _dev()
    m = _tiny_model(2)
    m.setup_caches(1, 32)
    g = DecodeGraph(m, d, native_sampling=True, temperature=0.0, top_k=32)
    assert g.native_sampling
    g.tok.fill_(1)
    g.pos.zero_()
    seq = []
    for _ in range(10):
        g.step()
        seq.append(int(g.next_tok.item()))
    assert int(g.pos.item()) == 10
    # greedy (T=0) tokens equal the torch-sampling path
    g2 = DecodeGraph(m, d, native_sampling=False, temperature=0.0, top_k=32)
    g2.tok.fill_(1)
    g2.pos.zero_()
    seq2 = []
    for _ in range(10):
        g2.step()
        seq2.append(int(g2.next_tok.item()))
    assert seq == seq2


@pytest.mark.parametrize("fold", [False, True])
def test_several_token_steps_per_graph_replay_decode_the_same_tokens(fold):
    """DecodeGraph(steps_per_replay=k): token, position and RNG counter feed back on the device, so k captured steps per replay
    give the tokens of k single-step replays (greedy and sampled), all of them in the sequence store"""
    from guidedquant_amd.generate import DecodeGraph
    d = _dev()
    m = _tiny_model(2)
    m.setup_caches(1, 32)
    for temp in (0.0, 0.8):
        seqs = []
        for k in (1, 4):
            g = DecodeGraph(m, d, native_sampling=True, temperature=temp, top_k=32, seed=7, fold_embed=fold, seq_capacity=32, steps_per_replay=k)
            assert g.steps_per_replay == k
            g.set_token(1, 0)
            for _ in range(12 // k):
                g.step()
            g.step_one()  # (the tail of a sequence: one step)
            torch.cuda.synchronize()
            assert int(g.pos.item()) == 13
            seqs.append(g.seq[1:14].tolist())
            assert seqs[-1][-1] == int(g.next_tok.item())
        assert seqs[0] == seqs[1], (temp, seqs)


def test_pipelined_decoder_native_single_rank():
    """the layer-pipeline driver on one rank (native fused kernels, 2 sequences with their own KV slots) reproduces the
    plain greedy decode of the same model"""
    from guidedquant_amd.generate import generate
    from guidedquant_amd.pipeline import PipelinedDecoder
    d = _dev()
    m = _tiny_model(2)
    dec = PipelinedDecoder(m, 0, 1, range(0, m.config.n_layer), n_seq=2, max_new_tokens=10, temperature=0.0, top_k=32, bos_id=1)
    assert dec.native
    with torch.no_grad():
        out = dec.run(10)
    torch.cuda.synchronize()
    assert out[0].tolist() == out[1].tolist()
    m2 = _tiny_model(2)
    ref = generate(m2, torch.tensor([1], dtype=torch.int32, device=d), 10, use_graph=False, temperature=0.0, top_k=32)
    assert out[0].tolist() == ref[0, 1:].tolist()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("N,K", [(28672, 4096), (2048, 1024)])
def test_silu_pairs_epilogue_matches_separate_ops(mode, N, K):
    """GQ_EPI_SILU_PAIRS on a row-interleaved (gate_i, up_i) tensor == the plain GEMV of the [gate; up] tensor followed
    by F.silu(gate) * up on fp16 tensors (inference/model.py:266), in both arithmetic modes."""
    import ctypes
    from guidedquant_amd import _lib, pack
    L = _lib.lib()
    d = torch.device("cuda:0")
    rng = np.random.default_rng(N + K)
    q = pack.random_planes(N, K, 2, seed=N + 1)
    lut = np.sort(rng.normal(0, 0.02, (N, 4)).astype(np.float16), axis=1)
    x = torch.from_numpy(rng.normal(0, 1, K).astype(np.float16)).to(d)
    qt, lt = torch.from_numpy(q).to(d), torch.from_numpy(lut).to(d)
    half = N // 2
    perm = torch.stack((torch.arange(half, device=d), torch.arange(half, N, device=d)), dim=1).reshape(-1)
    qp, lp = qt[:, perm, :].contiguous(), lt[perm].contiguous()
    _lib.check(L.gq_set_ap_mode(mode), "mode")
    try:
        y = torch.empty(N, dtype=torch.float16, device=d)
        _lib.check(L.gq_anyprec_gemv_fused(x.data_ptr(), y.data_ptr(), qt.data_ptr(), lt.data_ptr(), N, K, 2, None, 0.0, None, 0, None), "plain")
        o = torch.full((half,), float("nan"), dtype=torch.float16, device=d)
        _lib.check(L.gq_anyprec_gemv_fused(x.data_ptr(), o.data_ptr(), qp.data_ptr(), lp.data_ptr(), N, K, 2, None, 0.0, None, 4, None), "pairs")
        torch.cuda.synchronize()
    finally:
        L.gq_set_ap_mode(-1)
    want = torch.nn.functional.silu(y[:half]) * y[half:]
    # same fp16 rounding points; the only freedom is the last bit of exp()
    diff = (o.float() - want.float()).abs()
    tol = 2.0 ** -10 * want.float().abs() + 1e-7
    assert bool((diff <= tol).all()), float((diff / (want.float().abs() + 1e-6)).max())
    assert float((o != want).float().mean()) < 0.02
    # validation
    assert L.gq_anyprec_gemv_fused(x.data_ptr(), o.data_ptr(), qp.data_ptr(), lp.data_ptr(), N, K, 2, None, 0.0, y.data_ptr(), 5, None) != 0


@pytest.mark.parametrize("nsplit", [1, 2, 8, 5])
@pytest.mark.parametrize("hd,H,Hkv", [(128, 32, 8), (64, 8, 2)])
def test_attention_kernel_long_context(hd, H, Hkv, nsplit):
    """several passes of the position loop (128 positions per pass at head_dim 128) and a context far beyond what a
    score buffer in LDS would hold: pre-filled caches, decode at positions up to 5000; nsplit > 1: the split-KV form
    (gq_attn_decode_split: blocks over position ranges + combine), including splits that get no position at all"""
    from guidedquant_amd import _lib
    from guidedquant_amd.model import apply_rotary_pos_emb, rope_tables
    d = _dev()
    L = _lib.lib()
    max_seq = 5120
    g = torch.Generator(device=d)
    g.manual_seed(7)
    cos, sin = rope_tables(hd, max_seq, 500000.0, d)
    kc = (torch.randn(1, Hkv, max_seq, hd, device=d, generator=g) * 0.5).half()
    vc = (torch.randn(1, Hkv, max_seq, hd, device=d, generator=g)).half()
    out = torch.zeros(H * hd, dtype=torch.float16, device=d)
    for p in (0, 5, 127, 128, 129, 1000, 4999):
        kc_ref, vc_ref = kc.clone(), vc.clone()
        qkv = torch.randn((H + 2 * Hkv) * hd, device=d, generator=g).half()
        pos = torch.tensor([p], dtype=torch.int32, device=d)
        ws = torch.full((H * nsplit * (hd + 2),), float("nan"), dtype=torch.float32, device=d)
        _lib.check(L.gq_attn_decode_split(qkv.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                          out.data_ptr(), H, Hkv, hd, max_seq, 1.0 / math.sqrt(hd), nsplit, ws.data_ptr() if nsplit > 1 else None,
                                          _lib.current_stream_ptr()), "attn")
        q, k, v = qkv.split([H * hd, Hkv * hd, Hkv * hd])
        q = q.view(1, 1, H, hd).transpose(1, 2)
        k = k.view(1, 1, Hkv, hd).transpose(1, 2)
        v = v.view(1, 1, Hkv, hd).transpose(1, 2)
        q, k = apply_rotary_pos_emb(q, k, cos[p:p + 1].unsqueeze(0), sin[p:p + 1].unsqueeze(0))
        kc_ref[:, :, p] = k[:, :, 0]
        vc_ref[:, :, p] = v[:, :, 0]
        kk = kc_ref[:, :, :p + 1].float().repeat_interleave(H // Hkv, dim=1)
        vv = vc_ref[:, :, :p + 1].float().repeat_interleave(H // Hkv, dim=1)
        ref = (torch.softmax((q.float() @ kk.transpose(-1, -2)) / math.sqrt(hd), dim=-1) @ vv).transpose(1, 2).reshape(-1)
        torch.cuda.synchronize()
        assert torch.equal(kc[:, :, p], kc_ref[:, :, p]) and torch.equal(vc[:, :, p], vc_ref[:, :, p])
        assert (out.float() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


def test_decode_native_split_kv_attention_long_cache():
    """a cache long enough for the split-KV attention (max_seq_length 2048 -> 4 blocks per head + combine): the native
    step against the module forward while the context grows past the 256 positions one block per head still handles
    (head_dim 64: 2 x 256 positions per pass -> the switch is at 512)"""
    d = _dev()
    m = _tiny_model(2, 64)
    m.setup_caches(1, 2048)
    assert m.native_ready() and m._native_state()["attn_split"] == 4
    n = 540
    g = torch.Generator(device="cpu").manual_seed(3)
    toks = torch.randint(0, 1024, (n,), generator=g).tolist()
    check = {0, 1, 255, 256, 511, 512, 513, n - 1}
    ref = {}
    with torch.no_grad():
        for p, t in enumerate(toks):
            lg = m(torch.tensor([[t]], dtype=torch.int32, device=d), torch.tensor([p], dtype=torch.int32, device=d))
            if p in check:
                ref[p] = lg.float().view(-1).clone()
        for b in m.layers:
            b.attention.kv_cache.k_cache.zero_()
            b.attention.kv_cache.v_cache.zero_()
        for p, t in enumerate(toks):
            lg = m.decode_native(torch.tensor([t], dtype=torch.int32, device=d), torch.tensor([p], dtype=torch.int32, device=d))
            if p in check:
                torch.cuda.synchronize()
                a, r = lg.float().view(-1), ref[p]
                assert torch.isfinite(a).all()
                assert (a - r).abs().max().item() <= TOL * r.abs().max().item(), (p, (a - r).abs().max().item())


@pytest.mark.parametrize("nsplit", [1, 4])
def test_attention_past_the_cache_poisons_the_output(nsplit):
    """*pos >= max_seq: nothing is written to the caches and the heads' output is NaN (a loud failure downstream) instead of
    silently overwriting the last slot (include/gq_hip.h)"""
    from guidedquant_amd import _lib
    from guidedquant_amd.model import rope_tables
    d = _dev()
    L = _lib.lib()
    H, Hkv, hd, max_seq = 8, 2, 128, 64
    cos, sin = rope_tables(hd, max_seq, 500000.0, d)
    kc = torch.ones(1, Hkv, max_seq, hd, dtype=torch.float16, device=d)
    vc = torch.ones_like(kc)
    out = torch.zeros(H * hd, dtype=torch.float16, device=d)
    qkv = torch.randn((H + 2 * Hkv) * hd, device=d).half()
    ws = torch.zeros(H * nsplit * (hd + 2), dtype=torch.float32, device=d)
    for p in (max_seq, max_seq + 7):
        pos = torch.tensor([p], dtype=torch.int32, device=d)
        _lib.check(L.gq_attn_decode_split(qkv.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                          out.data_ptr(), H, Hkv, hd, max_seq, 0.1, nsplit, ws.data_ptr() if nsplit > 1 else None,
                                          _lib.current_stream_ptr()), "attn")
        torch.cuda.synchronize()
        assert bool(torch.isnan(out.float()).all())
        assert bool((kc == 1).all()) and bool((vc == 1).all())
    pos = torch.tensor([max_seq - 1], dtype=torch.int32, device=d)
    _lib.check(L.gq_attn_decode_split(qkv.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                      out.data_ptr(), H, Hkv, hd, max_seq, 0.1, nsplit, ws.data_ptr() if nsplit > 1 else None,
                                      _lib.current_stream_ptr()), "attn")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())


def test_converted_checkpoint_route_decodes_natively(tmp_path):
    """SURVEY section 8 f-1 / generate.py:222-245: a `converted_pytorch_model.bin` -- the converter's output on the seeded 32-layer
    checkpoint, the file tests/golden/convert_ap_fuse_b3.npz pins to the reference script's own output -- loaded through
    `load_model(random_init=False, checkpoint_path=...)` (torch.load mmap -> load_state_dict(assign, strict) -> .to(device)); eight
    tokens through the fused HIP decode step == the torch forward of the same weights, and generate() with the captured graph returns
    the same greedy tokens as the eager loop."""
    from ap_helpers import CONVERT_DIMS, convert_input_state_dict
    from guidedquant_amd.convert import convert_anyprec_fuse
    from guidedquant_amd.generate import generate, load_model
    from guidedquant_amd.model import transformer_configs
    d, c = _dev(), CONVERT_DIMS
    out = convert_anyprec_fuse(convert_input_state_dict(), 3)
    torch.save(out, tmp_path / "converted_pytorch_model.bin")
    name = "test/convert-golden-32l"
    transformer_configs[name] = dict(model_name="llama-convert-golden-32l", block_size=64, n_layer=c["Lr"], n_head=c["H"], n_local_heads=c["KV"], dim=c["D"],
                                     intermediate_size=c["I"], vocab_size=c["V"], rope_base=10000)
    try:
        m = load_model(name, d, "ap", 3, random_init=False, checkpoint_path=str(tmp_path))
    finally:
        del transformer_configs[name]
    assert m.layers[31].feed_forward.w2.qweight.is_cuda and torch.equal(m.layers[31].feed_forward.w2.qweight.cpu(), out["layers.31.feed_forward.w2.qweight"])
    m.setup_caches(1, 32)
    assert m.native_ready()
    toks = [5, 17, 200, 3, 3, 128, 44, 255]
    ref = []
    with torch.no_grad():
        for p, t in enumerate(toks):
            ref.append(m(torch.tensor([[t]], dtype=torch.int32, device=d), torch.tensor([p], dtype=torch.int32, device=d)).float().clone())
    for b in m.layers:
        b.attention.kv_cache.k_cache.zero_()
        b.attention.kv_cache.v_cache.zero_()
    with torch.no_grad():
        st = m._native_state()
        for p, t in enumerate(toks):
            # (the golden checkpoint is 128 wide: the HIP lm_head kernel takes widths in multiples of 512, so the fused layers are
            # checked through the torch head -- embedding lookup and all 32 layers run on the HIP path)
            pos = torch.tensor([p], dtype=torch.int32, device=d)
            m.native_embed(torch.tensor([t], dtype=torch.int32, device=d), st["x"], None)
            m.native_layers(st["x"], pos, 0, len(m.layers))
            lg = m.output(m.norm(st["x"].view(1, 1, -1))).float().view(-1)
            r = ref[p].view(-1)
            assert torch.isfinite(lg).all() and (lg - r).abs().max().item() <= TOL * r.abs().max().item(), p
    for i, b in enumerate(m.layers):  # (and the caches the HIP layers filled are the torch forward's)
        kc = b.attention.kv_cache.k_cache[:, :, :len(toks)].float()
        assert torch.isfinite(kc).all() and kc.abs().max().item() > 0, i
