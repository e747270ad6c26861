"""GPU: the three routes behind AnyPrecisionForCausalLM.generate (the reference's HF surface, inference_example.py:34-77) and the
extended fused sampler they share (gq_sample_topk_ex: 64 candidates, EOS suppression until min_new_tokens, sequence store, the next
step's embedding): route 1 = the fused decode model, taken automatically; route 2 (capture=True) = the module tree with its decode step
captured as one hipGraph over a transformers StaticCache; route 3 = transformers' own generate."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")
pytestmark = pytest.mark.gpu

from ap_helpers import tiny_hf_anyprec_checkpoint  # noqa: E402


def _single_precision_model(seed=5, D=512, I=1024, H=8, KV=2, V=512, Lr=3):
    from guidedquant_amd.AnyPrecisionForCausalLM import AnyPrecisionForCausalLM
    hf = transformers.LlamaConfig(hidden_size=D, intermediate_size=I, num_hidden_layers=Lr, num_attention_heads=H, num_key_value_heads=KV,
                                  vocab_size=V, max_position_embeddings=256, rms_norm_eps=1e-5, tie_word_embeddings=False)
    names = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]
    hf.anyprec = dict(seed_precision=2, parent_precision=2, group_count=1, arch_config=dict(module_names=names, model_name="model", layers_name="layers"))
    m = AnyPrecisionForCausalLM.from_config_random(hf, device=torch.device("cuda:0"), seed=seed)
    # (from_config_random draws N(0, 0.02) embeddings: scale them up so that the logits have margins)
    with torch.no_grad():
        m.model.model.embed_tokens.weight.mul_(25.0)
        m.model.lm_head.weight.mul_(10.0)
    return m


class _Collect:
    def __init__(self):
        self.toks, self.ended = [], False

    def put(self, value):
        self.toks += value.reshape(-1).tolist()

    def end(self):
        self.ended = True


def test_sampler_with_64_candidates_ban_and_sequence_store():
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    V, D = 128256, 256
    g = torch.Generator(device=d)
    g.manual_seed(4)
    logits = (torch.randn(V, device=d, generator=g) * 2).half()
    table = torch.randn(V, D, device=d, generator=g).half()
    ctr = torch.zeros(1, dtype=torch.int32, device=d)
    wv, wi = torch.zeros(128 * 64, dtype=torch.float32, device=d), torch.zeros(128 * 64, dtype=torch.int32, device=d)
    nt, tok, pos = (torch.zeros(1, dtype=torch.int32, device=d) for _ in range(3))
    seq = torch.full((16, ), -1, dtype=torch.int32, device=d)
    x = torch.zeros(D, dtype=torch.float16, device=d)
    ssq = torch.zeros(_lib.SSQ_SLOTS, dtype=torch.float32, device=d)
    ban = torch.zeros(6, dtype=torch.int32, device=d)
    top = torch.topk(logits.float(), 64)
    best, second = int(top.indices[0]), int(top.indices[1])

    def draw(T, k, with_ban=True):
        _lib.check(L.gq_sample_topk_ex(logits.data_ptr(), V, k, T, 77, ctr.data_ptr(), wv.data_ptr(), wi.data_ptr(), tok.data_ptr(), pos.data_ptr(),
                                       nt.data_ptr(), ban.data_ptr() if with_ban else None, seq.data_ptr(), seq.numel(), table.data_ptr(), x.data_ptr(),
                                       D, ssq.data_ptr(), _lib.current_stream_ptr()), "gq_sample_topk_ex")
        return int(nt.item())

    assert draw(0.0, 1) == best and int(seq[1]) == best and int(pos.item()) == 1
    assert torch.equal(x, table[best]) and abs(float(ssq.double().sum()) - float((table[best].double()**2).sum())) < 1e-3
    # the arg-max is banned while pos < until_pos: the runner-up is drawn, then the ban expires
    ban.copy_(torch.tensor([1, 3, best, 0, 0, 0], dtype=torch.int32))
    assert draw(0.0, 1) == second and draw(0.0, 1) == second and draw(0.0, 1) == best
    assert seq[:5].tolist() == [-1, best, second, second, best]
    ban.zero_()
    # 64 candidates: every draw is among torch.topk's 64, and the empirical distribution follows their softmax
    draws = [draw(1.0, 64) for _ in range(4000)]
    assert set(draws) <= set(top.indices.tolist())
    p = torch.softmax(top.values, dim=0).cpu().numpy()
    cnt = np.array([draws.count(int(i)) for i in top.indices.tolist()], dtype=np.float64) / len(draws)
    assert np.abs(cnt - p).max() < 0.04
    assert L.gq_sample_topk_ex(logits.data_ptr(), V, 65, 1.0, 7, ctr.data_ptr(), wv.data_ptr(), wi.data_ptr(), None, None, nt.data_ptr(), None, None, 0,
                               None, None, 0, None, None) != 0


def test_decode_graph_with_the_embedding_folded_into_the_sampler():
    from test_decode_gpu import _tiny_model
    from guidedquant_amd.generate import DecodeGraph
    d = torch.device("cuda:0")
    m = _tiny_model(2)
    m.setup_caches(1, 32)
    runs = []
    for fold in (False, True):
        g = DecodeGraph(m, d, native_sampling=True, temperature=0.0, top_k=32, fold_embed=fold, seq_capacity=33)
        assert g.fold_embed == fold
        g.set_token(1, 0)
        toks = []
        for _ in range(12):
            g.step()
            toks.append(int(g.next_tok.item()))
        assert g.seq[1:13].tolist() == toks and int(g.pos.item()) == 12
        runs.append(toks)
    assert runs[0] == runs[1]


def test_routes_agree_and_the_default_call_is_the_fused_model():
    m = _single_precision_model()
    d = m.device
    ids = torch.tensor([[3, 17, 5, 60, 2, 9]], device=d)
    eager = m.generate(ids, max_new_tokens=24, do_sample=False, native=False, pad_token_id=0)
    captured = m.generate(ids, max_new_tokens=24, do_sample=False, native=False, capture=True, pad_token_id=0)
    assert captured.shape == eager.shape == (1, 30) and torch.equal(captured, eager)   # same module tree, same arithmetic, arg-max
    captured2 = m.generate(ids, max_new_tokens=24, do_sample=False, native=False, capture=True, pad_token_id=0)  # the graph replayed on a fresh cache
    assert torch.equal(captured2, eager)
    # the reference's own call (inference_example.py:44-53) goes to the fused decode model without any extra keyword
    fused = m.generate(ids, max_new_tokens=24, do_sample=False, pad_token_id=0, attention_mask=torch.ones_like(ids), cache_implementation="static")
    assert ("decoder", 2) in m._native_cache and fused.shape == (1, 30) and fused.dtype == ids.dtype
    agree = float((fused[0, 6:] == eager[0, 6:]).float().mean())
    assert torch.equal(fused[0, :7], eager[0, :7]) and agree >= 0.8, (agree, fused, eager)  # other summation orders: late near-ties may flip
    # weights by reference: o_proj / down_proj / embeddings share storage.  The automatic route keeps the module tree whole: the
    # inner model's state_dict / direct buffer access after a plain generate() see every plane
    dec = m._native_cache[("decoder", 2)]
    l0 = m.get_model_layers()[0]
    assert dec.layers[0].attention.wo.qweight.data_ptr() == l0.self_attn.o_proj.qweight.data_ptr()
    assert dec.tok_embeddings.weight.data_ptr() == m.model.model.embed_tokens.weight.data_ptr()
    assert l0.self_attn.q_proj.qweight.numel() > 0 and l0.mlp.up_proj.qweight.numel() > 0 and m._released is None
    assert m.model.state_dict()["model.layers.0.mlp.gate_proj.qweight"].shape == (2, 1024, 16)
    # native=True releases the module tree's q/k/v/gate/up planes (one copy of every weight) -- a decoder built without the release is
    # rebuilt for it
    fused2 = m.generate(ids, max_new_tokens=24, do_sample=False, pad_token_id=0, native=True)
    assert torch.equal(fused2, fused)
    assert l0.self_attn.q_proj.qweight.numel() == 0 and l0.mlp.up_proj.qweight.numel() == 0 and m._released is not None
    # the module tree takes its planes back by itself (forward, state_dict -- also of the inner model) and gives the same logits as before
    again = m.generate(ids, max_new_tokens=24, do_sample=False, native=False, pad_token_id=0)
    assert torch.equal(again, eager) and l0.self_attn.q_proj.qweight.numel() > 0 and ("decoder", 2) not in m._native_cache
    sd = m.state_dict()
    assert sd["model.model.layers.0.mlp.gate_proj.qweight"].shape == (2, 1024, 16)
    m.generate(ids, max_new_tokens=4, do_sample=False, pad_token_id=0, native=True)
    assert l0.self_attn.q_proj.qweight.numel() == 0
    assert m.model.state_dict()["model.layers.0.self_attn.q_proj.qweight"].shape == (2, 512, 16) and m._released is None


def test_a_failed_build_of_the_fused_model_gives_the_planes_back():
    m = _single_precision_model(seed=3)
    before = {k: v.clone() for k, v in m.model.state_dict().items() if k.endswith("qweight")}
    real, calls = m._layer_linears, [0]

    def failing(layer):
        calls[0] += 1
        if calls[0] == 3:
            raise RuntimeError("out of memory (injected)")
        return real(layer)
    m._layer_linears = failing
    with pytest.raises(RuntimeError, match="injected"):
        m.native_decoder(2, release_planes=True)
    m._layer_linears = real
    assert m._released is None and ("decoder", 2) not in m._native_cache
    after = m.model.state_dict()
    assert all(torch.equal(after[k], v) for k, v in before.items())
    # and generate() does not swallow such an error into a silent fallback
    calls[0] = 0
    m._layer_linears = failing
    with pytest.raises(RuntimeError, match="injected"):
        m.generate(torch.tensor([[3, 4]], device=m.device), max_new_tokens=4, do_sample=False, native=True)
    m._layer_linears = real
    after = m.model.state_dict()
    assert all(torch.equal(after[k], v) for k, v in before.items())


def test_evicted_graphs_are_destroyed_before_the_next_capture():
    """round 5's driver bench died here: a captured entry evicted from the cache, its hipGraph left to the cyclic collector, the
    collector firing inside the NEXT capture (hipGraphExecDestroy while a stream is capturing -> std::terminate).  With the collector
    at its most eager (threshold 1: a collection at almost every allocation) both routes re-capture at new lengths."""
    import gc
    m = _single_precision_model(seed=11)
    ids = torch.tensor([[3, 17, 5]], device=m.device)
    old = gc.get_threshold()
    gc.set_threshold(1, 1, 1)
    try:
        outs = []
        for n in (5, 9, 6):
            outs.append(m.generate(ids, max_new_tokens=n, do_sample=False, native=False, capture=True, pad_token_id=0))
            assert sum(1 for k in m._native_cache if k[0] == "cap") == 1
        graphs = []
        for n in (5, 9, 6):
            outs.append(m.generate(ids, max_new_tokens=n, do_sample=False, pad_token_id=0))
            assert sum(1 for k in m._native_cache if k[0] == "graph") == 1
            graphs.append(next(v for k, v in m._native_cache.items() if k[0] == "graph"))
        # (round 6: the fused route's caches grow in powers of two from 256 positions -- the three lengths share ONE capture ..)
        assert graphs[0] is graphs[1] is graphs[2]
        # (.. and other sampling parameters evict it and capture again, here under the eager collector)
        for temp in (0.7, 0.9):
            m.generate(ids, max_new_tokens=5, do_sample=True, temperature=temp, top_k=8, pad_token_id=0)
            assert sum(1 for k in m._native_cache if k[0] == "graph") == 1
            assert next(v for k, v in m._native_cache.items() if k[0] == "graph") is not graphs[0]
        assert graphs[0].graph is None  # destroyed at eviction
        assert gc.isenabled()
    finally:
        gc.set_threshold(*old)
    assert torch.equal(outs[0][0, :8], outs[1][0, :8]) and torch.equal(outs[3][0, :8], outs[4][0, :8])
    # an explicit close leaves nothing to a finaliser
    g = next(v for k, v in m._native_cache.items() if k[0] == "graph")
    m._drop_native()
    assert g.graph is None and m._native_cache == {}


def test_sampled_calls_follow_torch_manual_seed():
    """transformers' generate is reproducible under torch.manual_seed; so are the fused routes (the sampler's counter word is drawn
    from torch's generator at every sampled call -- nothing of the seed is baked into a cached graph)"""
    m = _single_precision_model(seed=13)
    with torch.no_grad():
        m.model.lm_head.weight.mul_(0.05)   # flat logits: draws differ
    ids = torch.tensor([[9, 1, 4]], device=m.device)
    kw = dict(max_new_tokens=24, do_sample=True, temperature=1.0, top_k=50, pad_token_id=0)
    for extra in (dict(), dict(native=False, capture=True)):
        torch.manual_seed(77)
        a = m.generate(ids, **kw, **extra)
        b = m.generate(ids, **kw, **extra)
        torch.manual_seed(77)
        a2 = m.generate(ids, **kw, **extra)
        b2 = m.generate(ids, **kw, **extra)
        assert torch.equal(a, a2) and torch.equal(b, b2) and not torch.equal(a, b), extra


def test_eos_min_new_tokens_streamer_and_sampling():
    m = _single_precision_model(seed=9)
    d = m.device
    ids = torch.tensor([[7, 1, 200]], device=d)
    free = m.generate(ids, max_new_tokens=40, do_sample=False)[0].tolist()
    assert len(free) == 43
    eos = free[3 + 10]                     # the 11th new token becomes "EOS"
    first = free.index(eos, 3)
    st = _Collect()
    cut = m.generate(ids, max_new_tokens=40, do_sample=False, eos_token_id=eos, streamer=st)[0].tolist()
    assert cut == free[:first + 1] and st.ended and st.toks == cut
    # min_new_tokens: EOS cannot be drawn before (HF suppresses its logit); the sequence runs on past the old cut
    longer = m.generate(ids, max_new_tokens=40, min_new_tokens=25, do_sample=False, eos_token_id=eos)[0].tolist()
    assert len(longer) >= 3 + 25 and eos not in longer[3:3 + 25] and longer[:first] == free[:first]
    # the module tree with the captured step: same semantics
    cut2 = m.generate(ids, max_new_tokens=40, do_sample=False, eos_token_id=eos, native=False, capture=True)[0].tolist()
    ref = m.generate(ids, max_new_tokens=40, do_sample=False, eos_token_id=eos, native=False, pad_token_id=0)[0].tolist()
    assert cut2 == ref
    # sampling: the reference's arguments (temperature 1, top_p 1, no top_k -> generation_config's 50) are served by route 1
    s = m.generate(ids, max_new_tokens=16, do_sample=True, temperature=1.0, top_p=1.0, pad_token_id=0, attention_mask=torch.ones_like(ids),
                   cache_implementation="static")
    assert s.shape == (1, 19) and ("decoder", 2) in m._native_cache
    # round 6: top_p < 1 on top of the default top_k is served by route 1 too (fused nucleus filter)
    t = m.generate(ids, max_new_tokens=8, do_sample=True, top_p=0.9, pad_token_id=0, native=True)
    assert t.shape == (1, 11)
    # what the fused routes do not serve falls through to transformers (a top_k beyond the sampler's 64 candidates), and native=True then refuses
    t = m.generate(ids, max_new_tokens=8, do_sample=True, top_k=100, pad_token_id=0)
    assert t.shape[1] <= 11
    with pytest.raises(ValueError):
        m.generate(ids, max_new_tokens=8, do_sample=True, top_k=100, native=True)


def test_nucleus_filter_of_the_fused_sampler_matches_transformers_warper_chain():
    """gq_sample_topk_p against TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper of the installed transformers (the chain
    `generate(do_sample=True, temperature=T, top_k=k, top_p=p)` builds): no token outside the warped support is ever drawn, every token
    inside is, and the empirical distribution follows the warped probabilities."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    V = 4096
    g = torch.Generator(device=d)
    g.manual_seed(3)
    for k, p, T in ((50, 0.8, 1.0), (64, 0.5, 0.7), (20, 0.95, 1.3), (50, 0.05, 1.0)):
        logits = (torch.randn(V, device=d, generator=g) * 2.0).half()
        sc = logits.float().view(1, -1)
        ids0 = torch.zeros(1, 1, dtype=torch.long, device=d)
        for wp in (TemperatureLogitsWarper(T), TopKLogitsWarper(k), TopPLogitsWarper(p)):
            sc = wp(ids0, sc)
        want = torch.softmax(sc, dim=-1).view(-1).cpu().numpy()
        support = set(np.nonzero(want > 0)[0].tolist())
        ctr = torch.zeros(1, dtype=torch.int32, device=d)
        wv, wi = torch.zeros(128 * 64, dtype=torch.float32, device=d), torch.zeros(128 * 64, dtype=torch.int32, device=d)
        nt = torch.zeros(1, dtype=torch.int32, device=d)
        seq = torch.zeros(6001, dtype=torch.int32, device=d)
        pos = torch.zeros(1, dtype=torch.int32, device=d)
        tok = torch.zeros(1, dtype=torch.int32, device=d)
        for _ in range(6000):
            _lib.check(L.gq_sample_topk_p(logits.data_ptr(), V, k, p, T, 99, ctr.data_ptr(), wv.data_ptr(), wi.data_ptr(), tok.data_ptr(), pos.data_ptr(),
                                          nt.data_ptr(), None, seq.data_ptr(), seq.numel(), None, None, 0, None, _lib.current_stream_ptr()), "gq_sample_topk_p")
        torch.cuda.synchronize()
        draws = seq[1:6001].cpu().numpy()
        cnt = np.bincount(draws, minlength=V).astype(np.float64) / len(draws)
        assert set(np.nonzero(cnt)[0].tolist()) <= support, (k, p, T, sorted(set(np.nonzero(cnt)[0].tolist()) - support))
        assert np.abs(cnt - want).max() < 0.03, (k, p, T, np.abs(cnt - want).max())
        if len(support) <= 12:  # (a small nucleus: every member shows up)
            assert set(np.nonzero(cnt)[0].tolist()) == support
    # top_p = 1 is the old entry point, draw for draw
    logits = (torch.randn(V, device=d, generator=g) * 2.0).half()
    outs = []
    for fn in ("ex", "p"):
        ctr = torch.zeros(1, dtype=torch.int32, device=d)
        seq = torch.zeros(65, dtype=torch.int32, device=d)
        pos = torch.zeros(1, dtype=torch.int32, device=d)
        for _ in range(64):
            if fn == "ex":
                _lib.check(L.gq_sample_topk_ex(logits.data_ptr(), V, 50, 1.0, 7, ctr.data_ptr(), wv.data_ptr(), wi.data_ptr(), tok.data_ptr(), pos.data_ptr(),
                                               nt.data_ptr(), None, seq.data_ptr(), seq.numel(), None, None, 0, None, _lib.current_stream_ptr()), "ex")
            else:
                _lib.check(L.gq_sample_topk_p(logits.data_ptr(), V, 50, 1.0, 1.0, 7, ctr.data_ptr(), wv.data_ptr(), wi.data_ptr(), tok.data_ptr(), pos.data_ptr(),
                                              nt.data_ptr(), None, seq.data_ptr(), seq.numel(), None, None, 0, None, _lib.current_stream_ptr()), "p")
        torch.cuda.synchronize()
        outs.append(seq.clone())
    assert torch.equal(outs[0], outs[1])
