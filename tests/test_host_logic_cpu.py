"""CPU: host-side logic -- the product's packer against reference goldens, module surfaces, model tree / state-dict
contract, argument validation that must raise (not crash) without a GPU."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", golden_files("ap_b"))
def test_product_packer_matches_reference(path):
    from guidedquant_amd import pack
    g = np.load(path)
    b = int(g["bits"])
    assert np.array_equal(pack.pack_codes(g["codes"], b), g["qweight"])
    assert np.array_equal(pack.unpack_codes(g["qweight"], b), g["codes"])


def test_random_planes_are_a_valid_packing():
    from guidedquant_amd import pack
    q = pack.random_planes(8, 1056, 3, seed=1)
    codes = pack.unpack_codes(q, 3)
    assert codes.max() <= 7 and np.array_equal(pack.pack_codes(codes, 3), q)


def test_module_surfaces_on_cpu_device():
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.AnyPrecisionLinear import AnyPrecisionLinear
    from guidedquant_amd.LUTGEMMLinear import LUTGEMMLinear
    a = APLinear(4096, 6144, 2, device="cpu")
    assert a.qweight.shape == (2, 6144, 128) and a.qweight.dtype == torch.int32
    assert a.lut.shape == (6144, 4) and a.lut.dtype == torch.float16 and a.output.shape == (1, 1, 6144)
    assert set(a.state_dict()) == {"qweight", "lut"}
    l = LUTGEMMLinear(4096, 4096, 3, -1, device="cpu")
    assert l.group_size == 4096 and l.qweight.shape == (128, 3, 4096) and l.alpha.shape == (1, 3, 4096) and l.q_bias.shape == (1, 4096)
    h = AnyPrecisionLinear(4096, 1024, [2, 3, 4], bias=False, device="cpu", dtype=torch.float16)
    assert h.qweight.shape == (4, 1024, 128) and {"lut2", "lut3", "lut4"} <= set(h.state_dict())
    h.precisions = [2, 3]
    h.prune_precisions()
    assert h.qweight.shape == (3, 1024, 128) and "lut4" not in h.state_dict()
    with pytest.raises(RuntimeError, match="precisions are supported"):
        h.set_precision(4)


def test_validation_on_cpu_tensors():
    """all tensors in host memory: served by the library's CPU twins (configs[0]); the reference's validation still applies;
    LUT-GEMM has no CPU twin and keeps the reference's placement error"""
    from guidedquant_amd import ap_gemv
    x = torch.zeros(1, 1, 128, dtype=torch.float16)
    out = torch.ones(1, 1, 8, dtype=torch.float16)
    q = torch.zeros(2, 8, 4, dtype=torch.int32)
    lut = torch.zeros(8, 4, dtype=torch.float16)
    ap_gemv.anyprec_gemv(x, out, q, lut, 2)
    assert float(out.abs().max()) == 0.0
    assert ap_gemv.anyprec_dequant(q, lut, 2).shape == (8, 128)
    with pytest.raises(RuntimeError, match="Bitwidth must be between 2 and 8"):
        ap_gemv.anyprec_gemv(x, out, q, lut, 1)
    with pytest.raises(RuntimeError, match="lut tensor must be of shape"):
        ap_gemv.anyprec_gemv(x, out, q, lut, 3)
    with pytest.raises(RuntimeError, match="Only sequence length of 1"):
        ap_gemv.anyprec_gemv(torch.zeros(1, 2, 128, dtype=torch.float16), out, q, lut, 2)
    with pytest.raises(RuntimeError, match="must be on GPU"):
        ap_gemv.lutgemm_gemv(x, out, torch.zeros(4, 2, 8, dtype=torch.int32), torch.zeros(1, 2, 8, dtype=torch.float16),
                             torch.zeros(1, 8, dtype=torch.float16), 2, 128)


def test_model_tree_and_state_dict_contract():
    """keys/shapes a `converted_pytorch_model.bin` of the reference carries (sqllm_llama_convert_fuse.py:71-116)"""
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.model import Transformer, transformer_configs
    assert "meta-llama/Llama-3.2-1B-Instruct" in transformer_configs and "meta-llama/Llama-3.3-70B-Instruct" in transformer_configs
    with torch.device("meta"):
        m = Transformer.from_name(torch.float16, "meta-llama/Meta-Llama-3.1-8B-Instruct", linear_class=APLinear,
                                  linear_kwargs=dict(bitwidth=2, device="meta"))
    sd = m.state_dict()
    assert sd["layers.0.attention.wqkv.qweight"].shape == (2, 6144, 128)
    assert sd["layers.0.attention.wqkv.lut"].shape == (6144, 4)
    assert sd["layers.31.attention.wo.qweight"].shape == (2, 4096, 128)
    assert sd["layers.5.feed_forward.w1w3.qweight"].shape == (2, 28672, 128)
    assert sd["layers.5.feed_forward.w2.qweight"].shape == (2, 4096, 448)
    assert sd["output.weight"].shape == (128256, 4096) and sd["tok_embeddings.weight"].shape == (128256, 4096)
    assert "layers.0.input_layernorm.weight" in sd and "layers.0.post_attention_layernorm.weight" in sd and "norm.weight" in sd
    assert len(m.layers) == 32
    # quantized bytes per token of the 2-bit 8B model: 1.745 GB
    qbytes = sum(v.numel() * 4 for k, v in sd.items() if k.endswith("qweight"))
    assert qbytes == 1744830464


def test_rope_tables_and_generic_forward_on_cpu():
    """the torch statement of the model runs on CPU with nn.Linear (config 1 plumbing, no GPU)"""
    from guidedquant_amd.model import ModelArgs, Transformer, rope_tables
    cos, sin = rope_tables(64, 16, 500000.0, "cpu", torch.float32)
    assert cos.shape == (16, 64) and torch.allclose(cos[0], torch.ones(64)) and torch.allclose(sin[0], torch.zeros(64))
    cfg = ModelArgs(block_size=64, vocab_size=128, n_layer=2, n_head=4, dim=256, intermediate_size=512, n_local_heads=2,
                    model_name="llama-test")
    m = Transformer(torch.float32, cfg).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.setup_caches(1, 16)
    with torch.no_grad():
        a = m(torch.tensor([[3, 5, 7]]), torch.arange(3))
        m2 = Transformer(torch.float32, cfg).eval()
        m2.load_state_dict(sd)
        m2.setup_caches(1, 16)
        outs = [m2(torch.tensor([[t]]), torch.tensor([p])) for p, t in enumerate([3, 5, 7])]
    assert torch.allclose(a[0, 2], outs[2][0, 0], atol=1e-4)  # incremental decode == prefill


def test_qtip_host_side():
    from guidedquant_amd import qtip
    assert qtip.get_hadK(4096) == (None, 1)
    # the reference's factor tables ship as a packed data file (tools/make_hadamard_tables.py); factor order of
    # matmul_had.py:13-67; every shipped table is a Hadamard matrix and equals the reference-generated golden copies
    import glob
    import os
    import numpy as np
    for n, K in ((11008, 172), (14336, 28), (5120, 20), (13824, 108), (28672, 28), (6656, 52), (7680, 60), (20480, 20), (15872, 124)):
        h, k = qtip.get_hadK(n)
        assert k == K and tuple(h.shape) == (K, K)
        assert torch.equal(h @ h.T, K * torch.eye(K))
        ht, _ = qtip.get_hadK(n, transpose=True)
        assert torch.equal(ht, h.T)
    for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "had_n*.npz")):
        g = np.load(f)
        if int(g["K"]) > 1:
            assert np.array_equal(g["hadK"], qtip.get_hadK(int(g["n"]))[0].numpy()), f
    with pytest.raises(AssertionError):
        qtip.get_hadK(11 * 1024)
    assert qtip.has_kernel('quantlut_sym', 16, 2, 2, 9, 16, 16) and not qtip.has_kernel('lut', 16, 2, 2, 9, 16, 16)
    lin = qtip.QuantizedLinear(256, 512, 16, 16, 16, 3, 2, 9, 'quantlut_sym')
    assert lin.trellis.shape == (32 * 16, 48) and lin.tlut.shape == (512, 2) and lin.SU.shape == (256, ) and lin.SV.dtype == torch.float32
    assert {"trellis", "tlut", "SU", "SV", "rcp", "tp_rank"} == set(lin.state_dict())
    f = qtip.qtip_kernels.decompress_matvec_16_9_2_1_4096_1_4096
    with pytest.raises(RuntimeError, match="float32"):
        f(torch.zeros(4096, 1, dtype=torch.float16), torch.zeros(1, dtype=torch.int32), torch.zeros(4096, 1, dtype=torch.float16),
          torch.zeros(1024, dtype=torch.float16))


def test_anyprec_converter_fuses_like_the_reference():
    """HF-layout multi-precision checkpoint -> gpt-fast fused layout (sqllm_llama_convert_fuse.py:35-118): key renames,
    lut{b} -> lut, plane slicing, q/k/v and gate/up concatenation; the result loads strictly into the model."""
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.convert import convert_anyprec_fuse
    from guidedquant_amd.model import ModelArgs, Transformer
    cfg = ModelArgs(block_size=64, vocab_size=128, n_layer=2, n_head=4, dim=256, intermediate_size=512, n_local_heads=2,
                    model_name="llama-test")
    hd, kv, D, I = 64, 128, 256, 512
    g = torch.Generator().manual_seed(0)
    hf = {"model.embed_tokens.weight": torch.randn(128, D, generator=g).bfloat16(), "model.norm.weight": torch.ones(D).bfloat16(),
          "lm_head.weight": torch.randn(128, D, generator=g).bfloat16()}
    shapes = {"self_attn.q_proj": (D, D), "self_attn.k_proj": (kv, D), "self_attn.v_proj": (kv, D), "self_attn.o_proj": (D, D),
              "mlp.gate_proj": (I, D), "mlp.up_proj": (I, D), "mlp.down_proj": (D, I)}
    for i in range(2):
        hf[f"model.layers.{i}.input_layernorm.weight"] = torch.ones(D).bfloat16()
        hf[f"model.layers.{i}.post_attention_layernorm.weight"] = torch.ones(D).bfloat16()
        for name, (n, k) in shapes.items():
            hf[f"model.layers.{i}.{name}.qweight"] = torch.randint(-2**31, 2**31 - 1, (4, n, k // 32), dtype=torch.int32, generator=g)
            for b in (2, 3, 4):
                hf[f"model.layers.{i}.{name}.lut{b}"] = torch.randn(n, 2**b, generator=g).half()
    out = convert_anyprec_fuse(hf, 3)
    assert out["layers.1.attention.wqkv.qweight"].shape == (3, D + 2 * kv, D // 32)
    assert torch.equal(out["layers.1.attention.wqkv.qweight"][:, D:D + kv], hf["model.layers.1.self_attn.k_proj.qweight"][:3])
    assert torch.equal(out["layers.0.feed_forward.w1w3.lut"][I:], hf["model.layers.0.mlp.up_proj.lut3"])
    assert out["layers.0.attention.wo.lut"].shape == (D, 8) and out["tok_embeddings.weight"].dtype == torch.float16
    assert not any("lut2" in k or "lut4" in k or "q_proj" in k or "gate_proj" in k for k in out)
    m = Transformer(torch.float16, cfg, linear_class=APLinear, linear_kwargs=dict(bitwidth=3, device="cpu"))
    m.load_state_dict(out, strict=True)


def test_qtip_converter_renames():
    from guidedquant_amd.convert import convert_qtip_no_fuse
    out = convert_qtip_no_fuse({"model.layers.0.self_attn.q_proj.trellis": torch.zeros(1), "model.layers.0.mlp.down_proj.SU": torch.zeros(1),
                                "lm_head.weight": torch.zeros(1), "model.embed_tokens.weight": torch.zeros(1)})
    assert set(out) == {"layers.0.attention.wq.trellis", "layers.0.feed_forward.w2.SU", "output.weight", "tok_embeddings.weight"}


def test_load_model_qtip_backend_builds_unfused_tree():
    """generate.py:210-238 -- backend 'qtip': QuantizedLinear with the quip_params, separate wq/wk/wv and w1/w3 modules"""
    from guidedquant_amd import model as gm
    from guidedquant_amd.generate import load_model
    from guidedquant_amd.qtip import QuantizedLinear
    gm.transformer_configs["qtip-test"] = dict(model_name="llama-qtip-test", block_size=64, vocab_size=128, n_layer=1, n_head=4, dim=256, intermediate_size=512,
                                               n_local_heads=2)
    try:
        m = load_model("qtip-test", "cpu", "qtip", 2, random_init=True)
    finally:
        del gm.transformer_configs["qtip-test"]
    att, ff = m.layers[0].attention, m.layers[0].feed_forward
    assert not hasattr(att, "wqkv") and isinstance(att.wq, QuantizedLinear) and isinstance(ff.w3, QuantizedLinear)
    assert att.wk.out_features == 128 and ff.w2.in_features == 512 and att.wq.K == 2 and att.wq.decode_mode == "quantlut_sym"
    assert att.wq.trellis.dtype == torch.int16 and att.wq.trellis.shape == (256 // 16 * (256 // 16), 16 * 16 * 2 // 16)
    assert bool(att.wq.trellis.any()) and set(att.wq.SU.unique().tolist()) <= {-1.0, 1.0}


def test_hf_anyprec_loader(tmp_path):
    """HF-layout checkpoint directory (config.json with the anyprec section + safetensors with HF keys, tied embeddings)
    -> fused Transformer: geometry from the config, strict state-dict load, plane slicing at the requested precision."""
    import json
    from safetensors.torch import save_file
    from guidedquant_amd.hf_loader import load_anyprec_hf, model_args_from_hf_config
    D, I, H, KV, Lr, V = 256, 512, 4, 2, 2, 128
    cfg = dict(model_type="llama", _name_or_path="meta-llama/Llama-test", hidden_size=D, intermediate_size=I, num_hidden_layers=Lr,
               num_attention_heads=H, num_key_value_heads=KV, vocab_size=V, rope_theta=500000.0, rms_norm_eps=1e-5,
               max_position_embeddings=64, tie_word_embeddings=True, anyprec=dict(seed_precision=2, parent_precision=3))
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    g = torch.Generator().manual_seed(1)
    hd = D // H
    sd = {"model.embed_tokens.weight": torch.randn(V, D, generator=g).half(), "model.norm.weight": torch.ones(D).half()}
    shapes = {"self_attn.q_proj": (D, D), "self_attn.k_proj": (KV * hd, D), "self_attn.v_proj": (KV * hd, D), "self_attn.o_proj": (D, D),
              "mlp.gate_proj": (I, D), "mlp.up_proj": (I, D), "mlp.down_proj": (D, I)}
    for i in range(Lr):
        sd[f"model.layers.{i}.input_layernorm.weight"] = torch.ones(D).half()
        sd[f"model.layers.{i}.post_attention_layernorm.weight"] = torch.ones(D).half()
        for name, (n, k) in shapes.items():
            sd[f"model.layers.{i}.{name}.qweight"] = torch.randint(-2**31, 2**31 - 1, (3, n, k // 32), dtype=torch.int32, generator=g)
            for b in (2, 3):
                sd[f"model.layers.{i}.{name}.lut{b}"] = torch.randn(n, 2**b, generator=g).half()
    save_file(sd, str(tmp_path / "model.safetensors"))
    a = model_args_from_hf_config(cfg)
    assert (a.dim, a.n_head, a.n_local_heads, a.head_dim, a.intermediate_size, a.n_layer) == (D, H, KV, hd, I, Lr)
    m = load_anyprec_hf(str(tmp_path), bitwidth=2, device="cpu")
    assert m.layers[1].attention.wqkv.qweight.shape == (2, D + 2 * KV * hd, D // 32)
    assert torch.equal(m.layers[0].feed_forward.w1w3.lut[:I], sd["model.layers.0.mlp.gate_proj.lut2"])
    assert torch.equal(m.output.weight, sd["model.embed_tokens.weight"])  # tied
    with pytest.raises(ValueError):
        load_anyprec_hf(str(tmp_path), bitwidth=4, device="cpu")


def test_llama3_rope_scaling_formula():
    """rope_inv_freq against the scalar statement of apply_rope_scaling (inference/model.py:288-305) for the 3.1/3.3
    (factor 8) and 3.2 (factor 32) configs; `linear`; unsupported types raise instead of decoding with wrong frequencies"""
    import math
    torch = pytest.importorskip("torch")
    from guidedquant_amd.model import ModelArgs, Transformer, rope_inv_freq, rope_tables
    for hd, factor in ((128, 8.0), (64, 32.0)):
        rs = dict(rope_type="llama3", factor=factor, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
        base = 1.0 / (500000.0**(torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
        want = []
        for f in base:
            wl = 2 * math.pi / f
            if wl < 8192 / 4.0:
                want.append(f)
            elif wl > 8192 / 1.0:
                want.append(f / factor)
            else:
                sm = (8192 / wl - 1.0) / (4.0 - 1.0)
                want.append((1 - sm) * f / factor + sm * f)
        want = torch.tensor(want, dtype=torch.float32)
        got = rope_inv_freq(hd, 500000.0, rs, "cpu")
        assert torch.equal(got, want)
        assert int((got != base).sum()) > hd // 8 and torch.equal(got[:4], base[:4])  # high frequencies untouched
        cos, sin = rope_tables(hd, 64, 500000.0, "cpu", torch.float16, rope_scaling=rs)
        cos0, _ = rope_tables(hd, 64, 500000.0, "cpu", torch.float16)
        assert cos.shape == (64, hd) and not torch.equal(cos, cos0)
    assert torch.equal(rope_inv_freq(64, 1e4, dict(type="linear", factor=2.0), "cpu"), rope_inv_freq(64, 1e4, None, "cpu") / 2.0)
    with pytest.raises(NotImplementedError):
        rope_inv_freq(64, 1e4, dict(rope_type="yarn", factor=2.0), "cpu")
    # the model applies it: the tables of a Transformer built from a config with rope_scaling differ from the default ones
    cfg = ModelArgs(block_size=64, vocab_size=32, n_layer=1, n_head=2, dim=128, intermediate_size=256, n_local_heads=2,
                    rope_base=500000, model_name="llama-t", rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0,
                                                                              high_freq_factor=4.0, original_max_position_embeddings=32))
    m = Transformer(torch.float32, cfg)
    m.setup_caches(1, 64)
    c0, _ = rope_tables(64, 64, 500000.0, "cpu", torch.float32)
    assert not torch.equal(m.rope_cos, c0)


def test_gate_up_rows_are_paired_in_place_and_exported_in_reference_layout():
    """pair_gate_up_rows_: the fused decode step re-orders w1w3 rows to (gate_i, up_i) pairs in place (no second copy);
    state_dict() still exports [w1; w3] (inference/sqllm_llama_convert_fuse.py:97-103 layout), load_state_dict takes it,
    and the module forward (prefill path) de-interleaves the paired output"""
    torch = pytest.importorskip("torch")
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.model import FeedForward, ModelArgs, Transformer, pair_gate_up_rows_
    cfg = ModelArgs(block_size=64, vocab_size=32, n_layer=1, n_head=2, dim=128, intermediate_size=256, n_local_heads=2, model_name="llama-t")
    m = Transformer(torch.float16, cfg, linear_class=APLinear, linear_kwargs=dict(bitwidth=2, device="cpu"))
    w = m.layers[0].feed_forward.w1w3
    g = torch.Generator().manual_seed(0)
    w.qweight.copy_(torch.randint(-2**31, 2**31 - 1, w.qweight.shape, dtype=torch.int32, generator=g))
    w.lut.copy_(torch.randn(w.lut.shape, generator=g).half())
    kq, kl = "layers.0.feed_forward.w1w3.qweight", "layers.0.feed_forward.w1w3.lut"
    q0, l0 = w.qweight.clone(), w.lut.clone()
    pair_gate_up_rows_(w)
    pair_gate_up_rows_(w)  # idempotent
    assert w.gq_row_pairs and not torch.equal(w.qweight, q0)
    assert torch.equal(w.qweight[:, 0::2], q0[:, :256]) and torch.equal(w.qweight[:, 1::2], q0[:, 256:])
    assert torch.equal(w.lut[0::2], l0[:256]) and torch.equal(w.lut[1::2], l0[256:])
    sd = m.state_dict()
    assert torch.equal(sd[kq], q0) and torch.equal(sd[kl], l0)
    m._native = {"stale": True}
    m.load_state_dict({k: v for k, v in sd.items()}, strict=True)
    assert w.gq_row_pairs is False and torch.equal(w.qweight, q0) and m._native is None

    # module forward on a paired tensor: a stand-in linear whose output is its row index shows the de-interleave
    class Rows(torch.nn.Module):
        def __init__(self, i, o, bias=False):
            super().__init__()
            self.o = o
            self.lut = None

        def forward(self, x):
            return torch.arange(self.o, dtype=torch.float32).expand(*x.shape[:-1], self.o)
    ff = FeedForward(cfg, linear_class=Rows)
    ff.act_fn = lambda t: t
    ff.w2 = torch.nn.Identity()
    x = torch.zeros(1, 1, 128)
    plain = ff(x)  # gate_i * up_i = i * (256 + i)
    assert torch.equal(plain.view(-1), torch.arange(256.0) * (256 + torch.arange(256.0)))
    ff.w1w3.gq_row_pairs = True  # rows now (gate_0, up_0, ..): output index 2i is gate_i, 2i+1 is up_i
    paired = ff(x)
    assert torch.equal(paired.view(-1), (2 * torch.arange(256.0)) * (2 * torch.arange(256.0) + 1))


def test_anyprecision_for_causal_lm_from_quantized_on_cpu(tmp_path):
    """the HF-path entry (any_precision/modules/AnyPrecisionForCausalLM.py:123-168): a checkpoint directory with the `anyprec`
    config section -> HF Llama with AnyPrecisionLinear modules; forward(precision=b) equals the same HF model with the
    dequantised weights in nn.Linear; set_precision / prune_precisions / generate; host tensors run the CPU twins"""
    transformers = pytest.importorskip("transformers")
    from ap_helpers import tiny_hf_anyprec_checkpoint
    from guidedquant_amd import ap_gemv
    from guidedquant_amd.AnyPrecisionForCausalLM import AnyPrecisionForCausalLM
    from guidedquant_amd.AnyPrecisionLinear import AnyPrecisionLinear
    hf_cfg, sd, names, (D, I, H, KV, Lr, V) = tiny_hf_anyprec_checkpoint(tmp_path)
    hd = D // H
    m = AnyPrecisionForCausalLM.from_quantized(str(tmp_path), device="cpu")
    assert m.supported_bits == [2, 3] and m.precision == 3 and len(m.ap_linears) == 7 * Lr
    assert all(isinstance(l, AnyPrecisionLinear) for l in m.ap_linears) and m.layer_type == "LlamaDecoderLayer"
    ids = torch.tensor([[3, 17, 5, 60, 2]])
    for b in (2, 3):
        dense = transformers.LlamaForCausalLM(hf_cfg).half()
        dsd = {k: v for k, v in sd.items() if not (k.endswith(".qweight") or ".lut" in k)}
        for i in range(Lr):
            for name in names:
                p = f"model.layers.{i}.{name}"
                dsd[p + ".weight"] = ap_gemv.anyprec_dequant(sd[p + ".qweight"], sd[p + f".lut{b}"], b)
        dense.load_state_dict(dsd, strict=True)
        with torch.no_grad():
            want = dense.float()(ids).logits
            got = m(ids, precision=b).logits.float()           # prefill rows: dequant + matmul
            got1 = m(ids[:, :1], precision=b).logits.float()   # one row: the GEMV twin
        assert m.precision == 3  # restored
        assert float((got - want).abs().max()) <= 2e-2 * float(want.abs().max())
        assert float((got1[0, 0] - want[0, 0]).abs().max()) <= 2e-2 * float(want.abs().max())
    out = m.generate(ids[:, :2], max_new_tokens=4, do_sample=False, precision=2)
    assert out.shape == (1, 6) and m.precision == 3
    with pytest.raises(RuntimeError):
        m.set_precision(4)
    with pytest.raises(NotImplementedError):
        m.fuse_layers()
    m2 = AnyPrecisionForCausalLM.from_quantized(str(tmp_path), precisions=[2], device="cpu")
    assert m2.ap_linears[0].qweight.shape[0] == 2 and not hasattr(m2.ap_linears[0], "lut3")
    nat = m2.native_decoder()
    assert nat.layers[0].attention.wqkv.bitwidth == 2 and nat.layers[0].attention.wqkv.qweight.shape == (2, D + 2 * KV * hd, D // 32)


def test_generate_routes_the_reference_call_and_declines_the_rest(tmp_path):
    """AnyPrecisionForCausalLM._route_request (host logic of the HF surface, round 5): the reference's own generate call
    (inference_example.py:44-67: do_sample, temperature 1, top_p 1, pad_token_id, attention_mask of ones, static cache, streamer; top_k
    left to the generation config = 50) is one the fused decode route serves; beams, top_p < 1, batches, masked positions, processors
    and unknown keywords go to transformers' generate"""
    pytest.importorskip("transformers")
    import torch
    from ap_helpers import tiny_hf_anyprec_checkpoint
    from guidedquant_amd.AnyPrecisionForCausalLM import AnyPrecisionForCausalLM
    tiny_hf_anyprec_checkpoint(tmp_path)
    m = AnyPrecisionForCausalLM.from_quantized(str(tmp_path), device="cpu")
    ids = torch.tensor([[3, 17, 5]])
    req, why = m._route_request((ids, ), dict(max_new_tokens=20, min_new_tokens=20, do_sample=True, temperature=1.0, top_p=1.0, pad_token_id=0,
                                              attention_mask=torch.ones_like(ids), cache_implementation="static", streamer=object()))
    assert why is None and req["T"] == 3 and req["max_new"] == 20 and req["min_new"] == 20 and req["top_k"] == 50 and req["temperature"] == 1.0
    req, _ = m._route_request((), dict(input_ids=ids, max_new_tokens=4, do_sample=False, eos_token_id=[7, 9]))
    assert req["top_k"] == 1 and req["temperature"] == 0.0 and req["eos"] == [7, 9] and req["min_new"] == 0
    req, _ = m._route_request((ids, ), dict(max_length=10))
    assert req["max_new"] == 7
    # round 6: a nucleus on top of top-k <= 64 is served (transformers' warper chain); on top of no top-k it is not
    req, why = m._route_request((ids, ), dict(max_new_tokens=4, do_sample=True, top_p=0.9))
    assert why is None and req["top_p"] == 0.9 and req["top_k"] == 50
    for kw in (dict(max_new_tokens=4, do_sample=True, top_p=0.9, top_k=100), dict(max_new_tokens=4, do_sample=True, top_p=0.0),
               dict(max_new_tokens=4, num_beams=2), dict(max_new_tokens=4, do_sample=True, top_k=100),
               dict(max_new_tokens=4, repetition_penalty=1.2), dict(max_new_tokens=4, logits_processor=[]), dict(max_new_tokens=4, eos_token_id=[1, 2, 3, 4, 5]),
               dict(max_new_tokens=4, attention_mask=torch.tensor([[0, 1, 1]])), dict(), dict(max_new_tokens=0)):
        req, why = m._route_request((ids, ), kw)
        assert req is None and why, kw
    assert m._route_request((torch.tensor([[1, 2], [3, 4]]), ), dict(max_new_tokens=4))[0] is None   # a batch
    # generation_config.max_new_tokens is honoured (keyword > generation_config.max_new_tokens > max_length - T)
    keep = m.model.generation_config
    try:
        import types
        m.model.generation_config = types.SimpleNamespace(max_new_tokens=9, top_k=None, do_sample=True, temperature=None)
        # a transformers-4 style config (fields carry their defaults: a None was put there by the user): top_k=None = no top-k filtering,
        # which the 64-candidate sampler does not draw from -> declined; the keyword still wins
        assert m._route_request((ids, ), dict())[0] is None
        req, why = m._route_request((ids, ), dict(top_k=8))
        assert why is None and req["max_new"] == 9 and req["top_k"] == 8 and req["temperature"] == 1.0 and req["do_sample"]
        m.model.generation_config = types.SimpleNamespace(max_new_tokens=9, top_k=0, do_sample=True)
        assert m._route_request((ids, ), dict())[0] is None
    finally:
        m.model.generation_config = keep
    # on a host model the request is declined at the device check and transformers' generate runs
    out = m.generate(ids, max_new_tokens=3, do_sample=False, pad_token_id=0)
    assert out.shape == (1, 6)
    with pytest.raises(ValueError):
        m.generate(ids, max_new_tokens=3, do_sample=False, native=True)


@pytest.mark.parametrize("bitwidth", [2, 3])
def test_anyprec_converter_equals_the_reference_script_output(bitwidth):
    """SURVEY section 8 f-1, pinned: tests/golden/convert_ap_fuse_b{2,3}.npz hold what the reference's OWN script
    (inference/sqllm_llama_convert_fuse.py, run as a subprocess by tests/make_golden.py::gen_convert) wrote for the seeded 32-layer
    multi-precision checkpoint of ap_helpers.convert_input_state_dict: `convert_anyprec_fuse` must produce the same key set and, key
    for key, the same dtype, shape and bytes."""
    import json
    import torch
    from ap_helpers import convert_input_state_dict, tensor_digest
    from guidedquant_amd.convert import convert_anyprec_fuse
    g = np.load(os.path.join(GOLDEN, f"convert_ap_fuse_b{bitwidth}.npz"))
    meta = json.loads(str(g["meta"]))
    sd = convert_input_state_dict()
    norm = lambda d: (d[0], tuple(d[1]), d[2])  # noqa: E731
    assert {k: norm(tensor_digest(v)) for k, v in sd.items()} == {k: norm(d) for k, d in meta["input"].items()}, "the regenerated input is not the golden's"
    for n_layer in (None, 32):
        out = convert_anyprec_fuse(dict(sd), bitwidth, n_layer)
        assert set(out) == set(meta["output"]), (sorted(set(out) ^ set(meta["output"]))[:8])
        for k, v in out.items():
            dt, shape, sha = tensor_digest(v)
            assert [dt, list(shape), sha] == [meta["output"][k][0], list(meta["output"][k][1]), meta["output"][k][2]], k
    for name in g.files:  # (the tensors stored in full: layers 0 and 31, embeddings, norm, head)
        if name.startswith("full::"):
            k, v = name[6:], out[name[6:]]
            got = v.view(torch.int16).numpy() if v.dtype == torch.bfloat16 else v.numpy()
            assert got.dtype == g[name].dtype and np.array_equal(got, g[name]), k
    assert out["layers.0.attention.wqkv.qweight"].shape == (bitwidth, 128 + 64 + 64, 4) and out["layers.31.feed_forward.w1w3.lut"].shape == (512, 1 << bitwidth)
    assert out["tok_embeddings.weight"].dtype == torch.float16 and "layers.0.attention.q_proj.qweight" not in out


def test_qtip_converter_equals_the_reference_script_output():
    """the same for inference/qtip_convert_no_fuse.py:9-46 (renames only) on an hfized-QTIP key set"""
    import json
    from ap_helpers import qtip_convert_input_state_dict, tensor_digest
    from guidedquant_amd.convert import convert_qtip_no_fuse
    meta = json.loads(str(np.load(os.path.join(GOLDEN, "convert_qtip_no_fuse.npz"))["meta"]))
    sd = qtip_convert_input_state_dict()
    norm = lambda d: (d[0], tuple(d[1]), d[2])  # noqa: E731
    assert {k: norm(tensor_digest(v)) for k, v in sd.items()} == {k: norm(d) for k, d in meta["input"].items()}
    out = convert_qtip_no_fuse(sd)
    assert set(out) == set(meta["output"])
    for k, v in out.items():
        dt, shape, sha = tensor_digest(v)
        assert [dt, list(shape), sha] == [meta["output"][k][0], list(meta["output"][k][1]), meta["output"][k][2]], k


def test_converted_checkpoint_route_on_the_host(tmp_path):
    """generate.py:222-245 of the reference: torch.load(converted_pytorch_model.bin, mmap, weights_only) -> load_state_dict(assign,
    strict) -> model.to(device, dtype).  `load_model(random_init=False, checkpoint_path=...)` on the converter's output (the file the
    golden test above pins): strict load, and the model's forward equals a dense Llama forward over the dequantised weights."""
    import torch
    from ap_helpers import CONVERT_DIMS, convert_input_state_dict
    from guidedquant_amd import ap_gemv
    from guidedquant_amd.convert import convert_anyprec_fuse
    from guidedquant_amd.generate import load_model
    from guidedquant_amd.model import transformer_configs
    c = CONVERT_DIMS
    out = convert_anyprec_fuse(convert_input_state_dict(), 3)
    torch.save(out, tmp_path / "converted_pytorch_model.bin")
    name = "test/convert-golden-32l"
    transformer_configs[name] = dict(model_name="llama-convert-golden-32l", block_size=64, n_layer=c["Lr"], n_head=c["H"], n_local_heads=c["KV"], dim=c["D"],
                                     intermediate_size=c["I"], vocab_size=c["V"], rope_base=10000)
    try:
        m = load_model(name, "cpu", "ap", 3, random_init=False, checkpoint_path=str(tmp_path))
    finally:
        del transformer_configs[name]
    sd = m.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in out.items()) and set(sd) == set(out)
    w = m.layers[5].attention.wqkv
    W = ap_gemv.anyprec_dequant(w.qweight, w.lut, 3)
    assert W.shape == (256, 128) and torch.equal(w.qweight, out["layers.5.attention.wqkv.qweight"])
    m.setup_caches(1, 16)
    with torch.no_grad():
        lg = m(torch.tensor([[3, 9, 250]], dtype=torch.int32), torch.arange(3, dtype=torch.int32))
    assert lg.shape == (1, 3, c["V"]) and torch.isfinite(lg.float()).all()


def test_bench_side_legs_cannot_take_the_line_with_them():
    """bench.py (round 6): every side leg of the line runs in a child process; a leg that dies, prints nothing or hangs becomes an
    error record in its place (round 5's driver run lost the whole line to an abort in its ninth leg)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rec = bench.run_leg_child("no_such_leg", [], 120)          # the child ends with an error (no GPU here / unknown leg): rc != 0
    assert set(rec) >= {"error"} and rec["error"].startswith("rc ")
    rec = bench.run_leg_child("no_such_leg", [], 0.05)          # ... or does not finish in time
    assert rec == {"error": "timed out after 0 s"}
    # the legs of the N = 1 line: the three records of the contract outside other_configs, then the other BASELINE configs
    keys = [k for k, _, _, _ in bench.LEGS]
    assert keys[:3] == ["cpu_baseline", "roofline_by_shape", "exact_mode"] and len(keys) == 11
    assert {g for _, g, _, _ in bench.LEGS} == {None, "other_configs"}


def test_exact_gemv_plans_fit_the_occupancy():
    """round 6: the exact-order kernel's planner never asks for more resident blocks per CU than the instance's occupancy holds (8B w2 had
    been launched as 512 blocks of a kernel that fits ONE 512-thread block per CU: two rounds of blocks, 10.5 instead of 8.9 us).
    gq_debug_exact_plan = pick_quad_cfg without a launch (256 CUs assumed without a device)."""
    import ctypes
    from guidedquant_amd import _lib
    L = _lib.lib()
    shapes = {"8b": [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)], "70b": [(10240, 8192), (8192, 8192), (57344, 8192), (8192, 28672)],
              "1b": [(3072, 2048), (2048, 2048), (16384, 2048), (2048, 8192)], "7b": [(12288, 4096), (22016, 4096), (4096, 11008)]}
    seen = 0
    for model, shp in shapes.items():
        for N, K in shp:
            for bits in (2, 3, 4):
                for pro in (0, 1):
                    plan = (ctypes.c_uint32 * 6)()
                    rc = L.gq_debug_exact_plan(N, K, bits, pro, plan)
                    if rc != 0:
                        continue
                    T, RS, SPB, D, grid, bpc = list(plan)
                    seen += 1
                    assert T % 64 == 0 and 64 <= T <= 512 and RS >= 1 and SPB >= 1 and 1 <= D <= 4, (model, N, K, bits, pro, list(plan))
                    assert grid * SPB * RS >= N, (model, N, K, bits, pro, list(plan))          # every row has a slot
                    assert grid <= 256 * bpc, (model, N, K, bits, pro, list(plan))             # one round of blocks
    assert seen >= 60
