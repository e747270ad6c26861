"""GPU: RoPE + KV-cache write in the epilogue of the fused q / k / v GEMV (gq_anyprec_gemv_qkv_rope) and the attention launch that
starts at q k^T (gq_attn_decode_roped) against the two launches they replace (gq_anyprec_gemv_fused + gq_attn_decode_split:
inference/model.py:206-241 semantics, tested against torch in test_decode_gpu.py).  Same GEMV arithmetic (the stream kernel,
csrc/ap_stream.hip), same fp16 rounding points of apply_rotary_pos_emb, same score / softmax order: caches and outputs bit for bit."""
import math
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _setup(H, Hkv, hd, K, bits, max_seq, seed):
    from guidedquant_amd.model import rope_tables
    d = torch.device("cuda:0")
    g = torch.Generator(device=d)
    g.manual_seed(seed)
    N = (H + 2 * Hkv) * hd
    q = torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g)
    lut = (torch.randn(N, 1 << bits, device=d, generator=g) * 0.03).half()
    nw = (1 + 0.1 * torch.randn(K, device=d, generator=g)).half()
    cos, sin = rope_tables(hd, max_seq, 500000.0, d)
    return d, g, N, q, lut, nw, cos, sin


@pytest.mark.parametrize("H,Hkv,hd,K,max_seq,nsplit", [(32, 8, 128, 4096, 64, 1), (32, 8, 128, 4096, 1024, 4), (32, 8, 64, 2048, 48, 1),
                                                       (8, 2, 128, 4096, 40, 1)])
def test_qkv_rope_chain_is_bit_identical_to_the_two_launches(H, Hkv, hd, K, max_seq, nsplit):
    from guidedquant_amd import _lib
    L = _lib.lib()
    L.gq_set_ap_mode(0)
    os.environ["GQ_PL_MIN_MWEIGHTS"] = "0"
    os.environ["GQ_ST"] = "2"  # the unfused reference launch on the stream kernel too (same fp32 summation order)
    L.gq_reset_env_cache()
    try:
        bits = 2
        d, g, N, q, lut, nw, cos, sin = _setup(H, Hkv, hd, K, bits, max_seq, H + hd + K)
        if not L.gq_anyprec_qkv_rope_supported(N, K, bits, hd):
            pytest.skip("shape not served by the fused launch in this build")
        st = _lib.current_stream_ptr()
        kc = [torch.zeros(1, Hkv, max_seq, hd, dtype=torch.float16, device=d) for _ in range(2)]
        vc = [torch.zeros_like(kc[0]) for _ in range(2)]
        # stale rows past the position must not matter (the fused attention requests them before it knows the position)
        positions = list(range(12)) + ([200, 201, 300, 555] if max_seq >= 1024 else [max_seq - 9])
        junk = torch.randn(kc[0].shape, device=d, generator=g).half() * 50   # rows in the gaps: read by both chains, finite
        junk[:, :, positions[-1] + 1:, ::7] = float("nan")                    # rows no step may use: anything, NaN included
        for t in kc + vc:
            t.copy_(junk)
        ws = [torch.zeros(H * nsplit * (hd + 2), dtype=torch.float32, device=d) for _ in range(2)]
        qkv = [torch.zeros(N, dtype=torch.float16, device=d) for _ in range(2)]
        out = [torch.zeros(H * hd, dtype=torch.float16, device=d) for _ in range(2)]
        scale = 1.0 / math.sqrt(hd)
        # (positions are visited in order: both chains fill their caches the same way; gaps keep the junk rows of both identical)
        for p in positions:
            x = torch.randn(K, device=d, generator=g).half()
            if p % 5 == 0:
                x[torch.randint(0, K, (3,), device=d, generator=g)] *= 40.0  # massive channels through the extraction path
            pos = torch.tensor([p], dtype=torch.int32, device=d)
            _lib.check(L.gq_anyprec_gemv_fused(x.data_ptr(), qkv[0].data_ptr(), q.data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5,
                                               None, 0, st), "wqkv")
            _lib.check(L.gq_attn_decode_split(qkv[0].data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc[0].data_ptr(),
                                              vc[0].data_ptr(), out[0].data_ptr(), H, Hkv, hd, max_seq, scale, nsplit,
                                              ws[0].data_ptr() if nsplit > 1 else None, st), "attn")
            _lib.check(L.gq_anyprec_gemv_qkv_rope(x.data_ptr(), qkv[1].data_ptr(), q.data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5,
                                                  pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc[1].data_ptr(), vc[1].data_ptr(), H, Hkv, hd,
                                                  max_seq, st), "wqkv+rope")
            _lib.check(L.gq_attn_decode_roped(qkv[1].data_ptr(), pos.data_ptr(), kc[1].data_ptr(), vc[1].data_ptr(), out[1].data_ptr(), H, Hkv,
                                              hd, max_seq, scale, nsplit, ws[1].data_ptr() if nsplit > 1 else None, st), "attn roped")
            torch.cuda.synchronize()
            assert torch.equal(kc[0][:, :, p].view(torch.int16), kc[1][:, :, p].view(torch.int16)), ("rotated keys", p)
            assert torch.equal(vc[0][:, :, p].view(torch.int16), vc[1][:, :, p].view(torch.int16)), ("values", p)
            assert torch.isfinite(out[1].float()).all(), (p, torch.isnan(out[1].float()).view(H, hd).sum(1).tolist())
            assert torch.equal(out[0].view(torch.int16), out[1].view(torch.int16)), ("attention output", p, (out[0].float() - out[1].float()).abs().max().item())
        # rotated queries against apply_rotary_pos_emb on the unfused projection's output
        from guidedquant_amd.model import apply_rotary_pos_emb
        qq = qkv[0][:H * hd].view(1, 1, H, hd).transpose(1, 2)
        kk = qkv[0][H * hd:(H + Hkv) * hd].view(1, 1, Hkv, hd).transpose(1, 2)
        p = positions[-1]
        qr, _ = apply_rotary_pos_emb(qq, kk, cos[p:p + 1].unsqueeze(0), sin[p:p + 1].unsqueeze(0))
        assert torch.equal(qr.transpose(1, 2).reshape(-1).view(torch.int16), qkv[1][:H * hd].view(torch.int16))
        # past the cache: nothing written, output poisoned
        before = (kc[1].clone(), vc[1].clone())
        pos = torch.tensor([max_seq], dtype=torch.int32, device=d)
        _lib.check(L.gq_anyprec_gemv_qkv_rope(x.data_ptr(), qkv[1].data_ptr(), q.data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5,
                                              pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc[1].data_ptr(), vc[1].data_ptr(), H, Hkv, hd, max_seq,
                                              st), "wqkv+rope")
        _lib.check(L.gq_attn_decode_roped(qkv[1].data_ptr(), pos.data_ptr(), kc[1].data_ptr(), vc[1].data_ptr(), out[1].data_ptr(), H, Hkv, hd,
                                          max_seq, scale, nsplit, ws[1].data_ptr() if nsplit > 1 else None, st), "attn roped")
        torch.cuda.synchronize()
        assert torch.equal(before[0].view(torch.int16), kc[1].view(torch.int16)) and torch.equal(before[1].view(torch.int16), vc[1].view(torch.int16))
        assert torch.isnan(out[1].float()).all()
    finally:
        os.environ.pop("GQ_ST", None)
        L.gq_reset_env_cache()
        L.gq_set_ap_mode(-1)


def test_decode_step_with_the_fused_launch_matches_the_two_launch_step():
    """the whole native decode step at the 8B attention geometry: GQ_QKV_ROPE=0/1 give the same logits bit for bit"""
    from guidedquant_amd import _lib
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.generate import random_init_
    from guidedquant_amd.model import ModelArgs, Transformer
    L = _lib.lib()
    L.gq_set_ap_mode(0)
    os.environ["GQ_PL_MIN_MWEIGHTS"] = "0"
    d = torch.device("cuda:0")
    cfg = ModelArgs(block_size=128, vocab_size=2048, n_layer=2, n_head=32, dim=4096, intermediate_size=2048, n_local_heads=8,
                    rope_base=500000, model_name="llama-test")
    m = Transformer(torch.float16, cfg, linear_class=APLinear, linear_kwargs=dict(bitwidth=2, device=d)).to(device=d, dtype=torch.float16)
    random_init_(m, seed=3, lut_std=0.02)
    m.eval()
    m.setup_caches(1, 64)
    toks = [5, 17, 900, 3, 3, 512, 44, 1023, 7, 7]
    res = {}
    try:
        for flag in ("0", "1"):
            os.environ["GQ_QKV_ROPE"] = flag
            L.gq_reset_env_cache()
            for b in m.layers:
                b.attention.kv_cache.k_cache.zero_()
                b.attention.kv_cache.v_cache.zero_()
            outs = []
            with torch.no_grad():
                for p, t in enumerate(toks):
                    lg = m.decode_native(torch.tensor([t], dtype=torch.int32, device=d), torch.tensor([p], dtype=torch.int32, device=d))
                    torch.cuda.synchronize()
                    outs.append(lg.clone())
            res[flag] = (outs, [b.attention.kv_cache.k_cache.clone() for b in m.layers])
        assert L.gq_anyprec_qkv_rope_supported(cfg.dim + 2 * 8 * 128, cfg.dim, 2, 128)
        for a, b in zip(res["0"][0], res["1"][0]):
            assert torch.isfinite(b.float()).all() and torch.equal(a.view(torch.int16), b.view(torch.int16))
        for a, b in zip(res["0"][1], res["1"][1]):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    finally:
        os.environ.pop("GQ_QKV_ROPE", None)
        L.gq_reset_env_cache()
        L.gq_set_ap_mode(-1)


@pytest.mark.parametrize("H,Hkv,hd,max_seq,nsplit", [(32, 8, 128, 4224, 32), (32, 8, 128, 2048, 16), (16, 4, 64, 1500, 8), (8, 2, 128, 700, 4), (16, 2, 128, 1200, 8)])
def test_attention_with_the_heads_of_a_kv_group_in_one_block(H, Hkv, hd, max_seq, nsplit):
    """gq_attn_decode_roped on a grouped-query model with a split cache: the four query heads of a KV group share a block (every cached
    row loaded once, GQ_ATTN_GQA default on) -- bit-identical to one block per head (GQ_ATTN_GQA=0), at short positions (one block per
    head finishes the context), at split boundaries and at the end of the cache; and both against a float64 softmax(q k^T) v."""
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    g = torch.Generator(device=d)
    g.manual_seed(H + hd + max_seq)
    q = torch.randn(H * hd, device=d, generator=g).half()
    kc = torch.randn(Hkv, max_seq, hd, device=d, generator=g).half()
    vc = torch.randn(Hkv, max_seq, hd, device=d, generator=g).half()
    ws = torch.zeros(H * nsplit * (hd + 2), dtype=torch.float32, device=d)
    scale = 1.0 / math.sqrt(hd)
    per_pass = 8 * (64 // (hd // 8)) * 4
    positions = sorted({0, 5, 2 * per_pass - 1, 2 * per_pass, 2 * per_pass + 1, max_seq // 3, max_seq // 2 + 7, max_seq - 2, max_seq - 1})
    try:
        for p in positions:
            pos = torch.tensor([p], dtype=torch.int32, device=d)
            outs = []
            for flag in ("1", "0"):
                os.environ["GQ_ATTN_GQA"] = flag
                L.gq_reset_env_cache()
                out = torch.full((H * hd,), float("nan"), dtype=torch.float16, device=d)
                _lib.check(L.gq_attn_decode_roped(q.data_ptr(), pos.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), H, Hkv, hd, max_seq,
                                                  scale, nsplit, ws.data_ptr(), None), "attn")
                torch.cuda.synchronize()
                outs.append(out.clone())
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), p
            qf = q.double().view(Hkv, H // Hkv, hd)
            sc = torch.einsum("gqd,gtd->gqt", qf, kc[:, :p + 1].double()) * scale
            want = torch.einsum("gqt,gtd->gqd", torch.softmax(sc, dim=-1), vc[:, :p + 1].double()).reshape(-1)
            err = (outs[0].double() - want).abs().max().item()
            assert err <= 2e-3 * max(1.0, want.abs().max().item()), (p, err)
    finally:
        os.environ.pop("GQ_ATTN_GQA", None)
        L.gq_reset_env_cache()


@pytest.mark.parametrize("H,Hkv,hd,K,max_seq", [(32, 8, 128, 4096, 101), (32, 8, 128, 4096, 1024), (32, 8, 64, 2048, 200), (8, 2, 128, 4096, 40)])
def test_attention_inside_the_wqkv_launch_is_bit_identical_to_the_two_launches(H, Hkv, hd, K, max_seq):
    """round 6: gq_anyprec_gemv_qkv_rope_attn (the attention heads as extra blocks of the wqkv launch, waiting on device flags) against
    gq_anyprec_gemv_qkv_rope + gq_attn_decode_roped: rotated q, the caches and the attention output bit for bit, at positions on both
    sides of the first 128-position pass, with junk (NaN included) in the rows past the position, the same position twice in a row (the
    flags are re-armed by the head blocks), captured in a graph and replayed, and past the cache (poisoned output, flags still zero)."""
    from guidedquant_amd import _lib
    from guidedquant_amd._graphs import capture, release
    L = _lib.lib()
    L.gq_set_ap_mode(0)
    os.environ["GQ_PL_MIN_MWEIGHTS"] = "0"
    os.environ["GQ_QKV_ATTN"] = "1"  # (the one-launch form is off by default: measured no faster than the two launches)
    L.gq_reset_env_cache()
    try:
        bits = 2
        d, g, N, q, lut, nw, cos, sin = _setup(H, Hkv, hd, K, bits, max_seq, 7 + H + hd + K)
        if not L.gq_anyprec_qkv_rope_attn_supported(N, K, bits, hd, H, Hkv):
            pytest.skip("geometry not served by the one-launch form in this build / on this device")
        st = _lib.current_stream_ptr()
        kc = [torch.zeros(1, Hkv, max_seq, hd, dtype=torch.float16, device=d) for _ in range(2)]
        vc = [torch.zeros_like(kc[0]) for _ in range(2)]
        positions = [0, 1, 2, 3, 3, 15, 16, 17, 31, 32] + ([99, 100, 126, 127, 128, 129, 255, 256, 300, 300, 777, 1023] if max_seq >= 1024 else [max_seq - 2, max_seq - 1])
        positions = [p for p in positions if p < max_seq]
        junk = torch.randn(kc[0].shape, device=d, generator=g).half() * 50
        junk[:, :, positions[-1] + 1:, ::7] = float("nan")
        for t in kc + vc:
            t.copy_(junk)
        qkv = [torch.zeros(N, dtype=torch.float16, device=d) for _ in range(2)]
        out = [torch.zeros(H * hd, dtype=torch.float16, device=d) for _ in range(2)]
        flags = torch.zeros(H * _lib.ATTN_FLAG_STRIDE, dtype=torch.int32, device=d)
        scale = 1.0 / math.sqrt(hd)
        x = torch.randn(K, device=d, generator=g).half()
        pos = torch.zeros(1, dtype=torch.int32, device=d)

        def two():
            _lib.check(L.gq_anyprec_gemv_qkv_rope(x.data_ptr(), qkv[0].data_ptr(), q.data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5,
                                                  pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc[0].data_ptr(), vc[0].data_ptr(), H, Hkv, hd,
                                                  max_seq, st), "wqkv+rope")
            _lib.check(L.gq_attn_decode_roped(qkv[0].data_ptr(), pos.data_ptr(), kc[0].data_ptr(), vc[0].data_ptr(), out[0].data_ptr(), H, Hkv,
                                              hd, max_seq, scale, 1, None, st), "attn roped")

        def one(stream_ptr=None):
            _lib.check(L.gq_anyprec_gemv_qkv_rope_attn(x.data_ptr(), qkv[1].data_ptr(), q.data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5,
                                                       pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc[1].data_ptr(), vc[1].data_ptr(), H, Hkv, hd,
                                                       max_seq, out[1].data_ptr(), scale, flags.data_ptr(), stream_ptr or _lib.current_stream_ptr()),
                       "wqkv+rope+attention")

        def same(p, what):
            torch.cuda.synchronize()
            assert int(flags.abs().sum()) == 0, (what, p, "flags not re-armed")
            assert torch.equal(qkv[0][:H * hd].view(torch.int16), qkv[1][:H * hd].view(torch.int16)), (what, "rotated queries", p)
            assert torch.equal(kc[0][:, :, p].view(torch.int16), kc[1][:, :, p].view(torch.int16)), (what, "rotated keys", p)
            assert torch.equal(vc[0][:, :, p].view(torch.int16), vc[1][:, :, p].view(torch.int16)), (what, "values", p)
            assert torch.isfinite(out[1].float()).all(), (what, p, torch.isnan(out[1].float()).view(H, hd).sum(1).tolist())
            assert torch.equal(out[0].view(torch.int16), out[1].view(torch.int16)), (what, "attention output", p, (out[0].float() - out[1].float()).abs().max().item())

        for p in positions:
            x.copy_(torch.randn(K, device=d, generator=g).half())
            pos.fill_(p)
            two()
            one()
            same(p, "eager")
        # the same launch inside a captured graph, replayed over new inputs (what the decode step does)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with capture(gr, stream=s):
                for _ in range(3):  # (three launches back to back: every one finds its flags at zero)
                    one()
        torch.cuda.current_stream().wait_stream(s)
        for p in positions[-4:]:
            x.copy_(torch.randn(K, device=d, generator=g).half())
            pos.fill_(p)
            two()
            out[1].zero_()
            gr.replay()
            same(p, "graph")
        release(gr)
        # past the cache: nothing written, output poisoned, flags zero
        before = (kc[1].clone(), vc[1].clone())
        pos.fill_(max_seq)
        one()
        torch.cuda.synchronize()
        assert torch.equal(before[0].view(torch.int16), kc[1].view(torch.int16)) and torch.equal(before[1].view(torch.int16), vc[1].view(torch.int16))
        assert torch.isnan(out[1].float()).all() and int(flags.abs().sum()) == 0
        # a launch that finds a flag line NOT at zero (a caller that did not zero the buffer) must not hang: it may compute garbage, it ends
        flags[0] = 5
        pos.fill_(positions[0])
        one()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("GQ_QKV_ATTN", None)
        os.environ.pop("GQ_PL_MIN_MWEIGHTS", None)
        L.gq_reset_env_cache()
        L.gq_set_ap_mode(-1)


def test_decode_step_with_attention_inside_the_wqkv_launch_matches_the_two_launch_step():
    """the whole native decode step at the 8B attention geometry: GQ_QKV_ATTN=0/1 give the same logits and caches bit for bit, eager and
    as the captured multi-step graph of the benchmark (DecodeGraph, 10 token steps per replay)"""
    from guidedquant_amd import _lib
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.generate import DecodeGraph, random_init_
    from guidedquant_amd.model import ModelArgs, Transformer
    L = _lib.lib()
    L.gq_set_ap_mode(0)
    os.environ["GQ_PL_MIN_MWEIGHTS"] = "0"
    d = torch.device("cuda:0")
    cfg = ModelArgs(block_size=256, vocab_size=2048, n_layer=3, n_head=32, dim=4096, intermediate_size=2048, n_local_heads=8,
                    rope_base=500000, model_name="llama-test")
    m = Transformer(torch.float16, cfg, linear_class=APLinear, linear_kwargs=dict(bitwidth=2, device=d)).to(device=d, dtype=torch.float16)
    random_init_(m, seed=4, lut_std=0.02)
    m.eval()
    m.setup_caches(1, 160)
    toks = [5, 17, 900, 3, 3, 512, 44, 1023, 7, 7]
    res, seqs = {}, {}
    try:
        for flag in ("0", "1"):
            os.environ["GQ_QKV_ATTN"] = flag
            L.gq_reset_env_cache()
            m._reset_native()
            assert bool(L.gq_anyprec_qkv_rope_attn_supported(cfg.dim + 2 * 8 * 128, cfg.dim, 2, 128, 32, 8)) == (flag == "1")
            for b in m.layers:
                b.attention.kv_cache.k_cache.zero_()
                b.attention.kv_cache.v_cache.zero_()
            outs = []
            with torch.no_grad():
                for p, t in enumerate(toks):
                    lg = m.decode_native(torch.tensor([t], dtype=torch.int32, device=d), torch.tensor([p], dtype=torch.int32, device=d))
                    torch.cuda.synchronize()
                    outs.append(lg.clone())
            res[flag] = (outs, [b.attention.kv_cache.k_cache.clone() for b in m.layers])
            gr = DecodeGraph(m, d, native_sampling=True, temperature=0.0, top_k=32, fold_embed=True, seq_capacity=161, steps_per_replay=10)
            gr.set_token(1, 0)
            for _ in range(15):
                gr.step()
            torch.cuda.synchronize()
            seqs[flag] = gr.seq[1:151].clone()
            gr.close()
        for a, b in zip(res["0"][0], res["1"][0]):
            assert torch.isfinite(b.float()).all() and torch.equal(a.view(torch.int16), b.view(torch.int16))
        for a, b in zip(res["0"][1], res["1"][1]):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        assert torch.equal(seqs["0"], seqs["1"]) and int(m._native_state()["attn_flags"].abs().sum()) == 0
    finally:
        os.environ.pop("GQ_QKV_ATTN", None)
        L.gq_reset_env_cache()
        m._reset_native()
        L.gq_set_ap_mode(-1)
