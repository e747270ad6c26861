"""locate the NaN of the decode step with GQ_SSQ_HANDOVER=1 (tests/test_handover_gpu.py::test_decode_step_with_and_without_the_handover)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GQ_SSQ_HANDOVER"] = "1"
import torch
from guidedquant_amd import _lib
from guidedquant_amd.APLinear import APLinear
from guidedquant_amd.generate import random_init_
from guidedquant_amd.model import ModelArgs, Transformer
dev = torch.device("cuda:0")
args = ModelArgs(block_size=256, vocab_size=4096, n_layer=2, n_head=32, dim=4096, intermediate_size=14336, n_local_heads=8, rope_base=500000.0, model_name="llama-test")
m = Transformer(torch.float16, args, linear_class=APLinear, linear_kwargs=dict(bitwidth=2, device=dev), fuse_linears=True).to(device=dev, dtype=torch.float16).eval()
random_init_(m, seed=3)
m.setup_caches(1, 64)
L = _lib.lib(); st = None; c = m.config
b = m._native_state()
x, h, y, qkv, gu, ssq = b["x"], b["h"], b["y"], b["qkv"], b["gu"], b["ssq"]
tok = torch.tensor([5], dtype=torch.int32, device=dev); pos = torch.tensor([0], dtype=torch.int32, device=dev)
def chk(name, t):
    torch.cuda.synchronize(); f = t.float()
    print(f"{name:10s} finite={bool(torch.isfinite(f).all())} absmax={float(f.abs().max()):.4g} sum={float(f.double().sum()):.6g}")
m.native_embed(tok, x, ssq); chk("x", x); chk("ssq", ssq); print("  want ssq", float((x.double()**2).sum()))
blk = m.layers[0]; at, ff = blk.attention, blk.feed_forward
kc, vc = at.kv_cache.k_cache.data_ptr(), at.kv_cache.v_cache.data_ptr()
for use in (None, ssq.data_ptr()):
    rc = L.gq_anyprec_gemv_qkv_rope_ho(x.data_ptr(), qkv.data_ptr(), at.wqkv.qweight.data_ptr(), at.wqkv.lut.data_ptr(), at.wqkv.out_features, c.dim, 2,
                                       blk.input_layernorm.weight.data_ptr(), c.norm_eps, pos.data_ptr(), m.rope_cos.data_ptr(), m.rope_sin.data_ptr(), kc, vc,
                                       c.n_head, c.n_local_heads, c.head_dim, m.max_seq_length, use, st)
    assert rc == 0, L.gq_last_error(); chk("qkv ho=%s" % (use is not None), qkv)
L.gq_attn_decode_roped(qkv.data_ptr(), pos.data_ptr(), kc, vc, y.data_ptr(), c.n_head, c.n_local_heads, c.head_dim, m.max_seq_length, 1 / math.sqrt(c.head_dim), 1, None, st)
chk("y", y)
rc = L.gq_anyprec_gemv_fused_ho(y.data_ptr(), h.data_ptr(), at.wo.qweight.data_ptr(), at.wo.lut.data_ptr(), c.dim, c.dim, 2, None, 0.0, x.data_ptr(), 1, None, 0, None, ssq.data_ptr(), st)
assert rc == 0; chk("h", h); chk("ssq(h)", ssq); print("  want", float((h.double()**2).sum()))
for use in (None, ssq.data_ptr()):
    rc = L.gq_anyprec_gemv_fused_ho(h.data_ptr(), gu.data_ptr(), ff.w1w3.qweight.data_ptr(), ff.w1w3.lut.data_ptr(), 2 * c.intermediate_size, c.dim, 2,
                                    blk.post_attention_layernorm.weight.data_ptr(), c.norm_eps, None, 4, None, 0, use, None, st)
    assert rc == 0; chk("gu ho=%s" % (use is not None), gu[:c.intermediate_size])
rc = L.gq_anyprec_gemv_fused_ho(gu.data_ptr(), x.data_ptr(), ff.w2.qweight.data_ptr(), ff.w2.lut.data_ptr(), c.dim, c.intermediate_size, 2, None, 0.0, h.data_ptr(), 1, None, 0, None, ssq.data_ptr(), st)
assert rc == 0; chk("x2", x); chk("ssq(x2)", ssq); print("  want", float((x.double()**2).sum()))
lg = m.decode_native(tok, pos); chk("logits", lg)
