"""gpt-fast style Transformer -- the caller of the quantized linears, mirroring inference/model.py of the reference.

Same public surface: `ModelArgs`, `transformer_configs`, `KVCache`, `Transformer.from_name(dtype, name, linear_class,
linear_kwargs, halve_layers, fuse_linears)`, `setup_caches`, `forward(idx, input_pos)`; same module tree and state-dict
keys (`layers.{i}.attention.{wqkv,wo}`, `layers.{i}.feed_forward.{w1w3,w2}`, `..._layernorm.weight`, `norm.weight`,
`output.weight`, `tok_embeddings.weight`; inference/sqllm_llama_convert_fuse.py:73-116), so a
`converted_pytorch_model.bin` written for the reference loads unchanged.

Two execution paths:
  * `forward(idx, input_pos)`: plain PyTorch ops around `linear_class` modules -- the reference's semantics
    (model.py:121-130, 151-166, 206-266), used for prefill, for any linear class, and as the on-device statement the
    fused path is tested against.  RoPE is re-stated here (`rope_tables`): the reference imports
    `ROPE_INIT_FUNCTIONS["default"]`, which the installed transformers no longer has (model.py:15,353).
  * `decode_native(tok, pos)`: one bs=1 decode step as 5 HIP launches per layer through the C ABI
    (RMSNorm -> wqkv | RoPE + KV update + attention | wo + residual | RMSNorm -> w1w3 | SiLU*up -> w2 + residual),
    an embedding lookup and a fused final-norm + lm_head GEMV: no allocation, no host sync, token / position read
    from device memory, so the whole step is hipGraph-capturable (generate.py).

Model table: the reference's entries (model.py:53-61) plus Llama-3.2-1B-Instruct and Llama-3.3-70B-Instruct, which
BASELINE.json's configs name and the reference table lacks.
"""
import ctypes
import math
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import functional as F

from . import _lib


def find_multiple(n: int, k: int) -> int:
    if n % k == 0:
        return n
    return n + k - (n % k)


@dataclass
class ModelArgs:
    block_size: int = 2048
    vocab_size: int = 32000
    n_layer: int = 32
    n_head: int = 32
    dim: int = 4096
    intermediate_size: int = None
    n_local_heads: int = -1
    head_dim: int = 64
    rope_base: float = 10000
    norm_eps: float = 1e-5
    rope_scaling: Optional[dict] = None
    model_name: Optional[str] = None

    def __post_init__(self):
        if self.n_local_heads == -1:
            self.n_local_heads = self.n_head
        if self.intermediate_size is None:
            hidden_dim = 4 * self.dim
            n_hidden = int(2 * hidden_dim / 3)
            self.intermediate_size = find_multiple(n_hidden, 256)
        self.head_dim = self.dim // self.n_head

    @classmethod
    def from_name(cls, name: str):
        assert name in transformer_configs, f"Unknown model name: {name}, available: {transformer_configs.keys()}"
        return cls(**transformer_configs[name])


transformer_configs = {
    "meta-llama/Meta-Llama-3-8B": dict(model_name="Meta-Llama-3-8B", block_size=8192, n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336, vocab_size=128256, rope_base=500000),
    "meta-llama/Meta-Llama-3-8B-Instruct": dict(model_name="Meta-Llama-3-8B-Instruct", block_size=8192, n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336, vocab_size=128256, rope_base=500000),
    "meta-llama/Meta-Llama-3.1-8B": dict(model_name="Meta-Llama-3.1-8B", block_size=8192, n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336, vocab_size=128256, rope_base=500000),
    "meta-llama/Meta-Llama-3.1-8B-Instruct": dict(model_name="Meta-Llama-3.1-8B-Instruct", block_size=8192, n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336, vocab_size=128256, rope_base=500000),
    "meta-llama/Llama-2-7b": dict(model_name="Llama-2-7b", block_size=4096, n_layer=32, n_head=32, n_local_heads=32, dim=4096, intermediate_size=11008, vocab_size=32000, rope_base=10000),
    "meta-llama/Llama-2-13b": dict(model_name="Llama-2-13b", block_size=4096, n_layer=40, n_head=40, n_local_heads=40, dim=5120, intermediate_size=13824, vocab_size=32000, rope_base=10000),
    "meta-llama/Llama-2-70b": dict(model_name="Llama-2-70b", block_size=4096, n_layer=80, n_head=64, n_local_heads=8, dim=8192, intermediate_size=28672, vocab_size=32000, rope_base=10000),
    # not in the reference table: public HF configs, with the llama3 rope_scaling their config.json
    # carries (the reference's own 3.1 entries above have none and are kept as the reference has them)
    "meta-llama/Llama-3.2-1B-Instruct": dict(model_name="Llama-3.2-1B-Instruct", block_size=8192, n_layer=16, n_head=32, n_local_heads=8, dim=2048, intermediate_size=8192, vocab_size=128256, rope_base=500000,
                                             rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)),
    "meta-llama/Llama-3.3-70B-Instruct": dict(model_name="Llama-3.3-70B-Instruct", block_size=8192, n_layer=80, n_head=64, n_local_heads=8, dim=8192, intermediate_size=28672, vocab_size=128256, rope_base=500000,
                                              rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)),
}


def rope_inv_freq(head_dim: int, base: float, rope_scaling: Optional[dict], device):
    """fp32 inverse frequencies of the rotary embedding for the `rope_scaling` section of a Llama config.json:
    None / "default": base^-(2i/d);  "linear": the same divided by `factor`;  "llama3" (Llama-3.1 / 3.2 / 3.3 checkpoints,
    inference/model.py:288-305 `apply_rope_scaling`): components whose wavelength exceeds original_max_position_embeddings /
    low_freq_factor are divided by `factor`, those shorter than original / high_freq_factor are kept, the band between is
    blended linearly in original / wavelength.  Any other type raises: decoding with the wrong frequencies would differ
    from the HF path that produced and evaluated the checkpoint at every position, silently."""
    inv_freq = 1.0 / (base**(torch.arange(0, head_dim, 2, dtype=torch.int64).to(dtype=torch.float32, device=device) / head_dim))
    if not rope_scaling:
        return inv_freq
    kind = rope_scaling.get("rope_type", rope_scaling.get("type", "default"))
    if kind == "default":
        return inv_freq
    if kind == "linear":
        return inv_freq / float(rope_scaling["factor"])
    if kind != "llama3":
        raise NotImplementedError(f"rope_scaling type {kind!r} is not implemented (default, linear, llama3 are)")
    factor = float(rope_scaling["factor"])
    lo, hi = float(rope_scaling["low_freq_factor"]), float(rope_scaling["high_freq_factor"])
    old_len = float(rope_scaling["original_max_position_embeddings"])
    wavelen = 2.0 * math.pi / inv_freq
    blend = ((old_len / wavelen - lo) / (hi - lo)).clamp(0.0, 1.0)  # 0: long wavelengths (scaled), 1: short ones (kept)
    scaled = inv_freq / factor
    out = torch.where(wavelen > old_len / lo, scaled, torch.where(wavelen < old_len / hi, inv_freq, (1.0 - blend) * scaled + blend * inv_freq))
    return out.to(torch.float32)


def rope_tables(head_dim: int, max_seq: int, base: float, device, dtype=torch.float16, rope_scaling: Optional[dict] = None):
    """cos/sin [max_seq, head_dim] as LlamaRotaryEmbedding.forward produces them (model.py:343-405): inv_freq in fp32
    (`rope_inv_freq`), freqs = pos * inv_freq, emb = cat(freqs, freqs), cos/sin in fp32, then cast to the activation dtype
    (attention_scaling == 1 for default / linear / llama3)."""
    inv_freq = rope_inv_freq(head_dim, base, rope_scaling, device)
    pos = torch.arange(max_seq, device=device, dtype=torch.float32)
    freqs = torch.outer(pos, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype).contiguous(), emb.sin().to(dtype).contiguous()


def rotate_half(x):
    x1 = x[..., :x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=1):
    cos = cos.unsqueeze(unsqueeze_dim)
    sin = sin.unsqueeze(unsqueeze_dim)
    q_embed = (q * cos) + (rotate_half(q) * sin)
    k_embed = (k * cos) + (rotate_half(k) * sin)
    return q_embed, k_embed


_SDPA_GQA = [True]  # F.scaled_dot_product_attention(enable_gqa=True): grouped K / V heads without the repeat_interleave copies


def _sdpa_gqa(q, k, v, mask, rep):
    """q [1, H, S, hd], k / v [1, Hkv, T, hd] (views of the cache), causal when mask is None.  Bit-identical to the form with the
    K / V heads repeated (measured on this stack: 21.6 vs 34.3 us at S = 128); older torch versions take the copies."""
    if _SDPA_GQA[0] and rep > 1:
        try:
            return F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=mask is None, enable_gqa=True)
        except (TypeError, RuntimeError):
            _SDPA_GQA[0] = False
    if rep > 1:
        Hkv, T, hd = k.shape[1], k.shape[2], k.shape[3]
        k = k[0].unsqueeze(1).expand(Hkv, rep, T, hd).reshape(1, Hkv * rep, T, hd)
        v = v[0].unsqueeze(1).expand(Hkv, rep, T, hd).reshape(1, Hkv * rep, T, hd)
    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=mask is None)


class KVCache(nn.Module):

    def __init__(self, max_batch_size, max_seq_length, n_heads, head_dim, dtype=torch.half, device=None):
        super().__init__()
        cache_shape = (max_batch_size, n_heads, max_seq_length, head_dim)
        self.register_buffer('k_cache', torch.zeros(cache_shape, dtype=dtype, device=device))
        self.register_buffer('v_cache', torch.zeros(cache_shape, dtype=dtype, device=device))

    def update(self, input_pos, k_val, v_val):
        assert input_pos.shape[0] == k_val.shape[2]
        k_out = self.k_cache
        v_out = self.v_cache
        k_out[:, :, input_pos] = k_val
        v_out[:, :, input_pos] = v_val
        return k_out, v_out


class RMSNorm(nn.Module):

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def _norm(self, x):
        return x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + self.eps)

    def forward(self, x: Tensor) -> Tensor:
        output = self._norm(x.float()).type_as(x)
        return output * self.weight


class Attention(nn.Module):

    def __init__(self, config: ModelArgs, linear_class=nn.Linear, linear_kwargs=None, fuse_linears=True) -> None:
        super().__init__()
        assert config.dim % config.n_head == 0
        total_head_dim = (config.n_head + 2 * config.n_local_heads) * config.head_dim
        if fuse_linears:
            self.wqkv = linear_class(config.dim, total_head_dim, bias=False, **(linear_kwargs or {}))
        else:
            self.wq = linear_class(config.dim, config.n_head * config.head_dim, bias=False, **(linear_kwargs or {}))
            self.wk = linear_class(config.dim, config.n_local_heads * config.head_dim, bias=False, **(linear_kwargs or {}))
            self.wv = linear_class(config.dim, config.n_local_heads * config.head_dim, bias=False, **(linear_kwargs or {}))
        self.wo = linear_class(config.dim, config.dim, bias=False, **(linear_kwargs or {}))
        self.kv_cache = None
        self.n_head = config.n_head
        self.head_dim = config.head_dim
        self.n_local_heads = config.n_local_heads
        self.dim = config.dim
        self.config = config
        self.fuse_linears = fuse_linears

    def forward(self, x: Tensor, mask: Tensor, input_pos: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
        bsz, seqlen, _ = x.shape
        kv_size = self.n_local_heads * self.head_dim
        if self.fuse_linears:
            q, k, v = self.wqkv(x).split([self.dim, kv_size, kv_size], dim=-1)
        else:
            q, k, v = self.wq(x), self.wk(x), self.wv(x)
        q = q.view(bsz, seqlen, self.n_head, self.head_dim)
        k = k.view(bsz, seqlen, self.n_local_heads, self.head_dim)
        v = v.view(bsz, seqlen, self.n_local_heads, self.head_dim)
        q, k, v = map(lambda t: t.transpose(1, 2), (q, k, v))
        q, k = apply_rotary_pos_emb(q, k, cos[input_pos].unsqueeze(0), sin[input_pos].unsqueeze(0))
        if self.kv_cache is not None:
            k, v = self.kv_cache.update(input_pos, k, v)
        k = k.repeat_interleave(self.n_head // self.n_local_heads, dim=1)
        v = v.repeat_interleave(self.n_head // self.n_local_heads, dim=1)
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0)
        y = y.transpose(1, 2).contiguous().view(bsz, seqlen, -1)
        return self.wo(y)


class FeedForward(nn.Module):

    def __init__(self, config: ModelArgs, linear_class=nn.Linear, linear_kwargs=None, fuse_linears=True) -> None:
        super().__init__()
        self.config = config
        self.fuse_linears = fuse_linears
        if fuse_linears:
            self.w1w3 = linear_class(config.dim, config.intermediate_size * 2, bias=False, **(linear_kwargs or {}))
            if hasattr(self.w1w3, "lut"):  # Any-Precision linear: the fused decode step may pair its rows in place
                self.w1w3._register_state_dict_hook(_unpair_on_export)
                self.w1w3._register_load_state_dict_pre_hook(_incoming_is_reference_layout, with_module=True)
        else:
            self.w1 = linear_class(config.dim, config.intermediate_size, bias=False, **(linear_kwargs or {}))
            self.w3 = linear_class(config.dim, config.intermediate_size, bias=False, **(linear_kwargs or {}))
        self.w2 = linear_class(config.intermediate_size, config.dim, bias=False, **(linear_kwargs or {}))
        self.act_fn = F.silu

    def forward(self, x: Tensor) -> Tensor:
        if self.fuse_linears and getattr(self.w1w3, "gq_row_pairs", False):
            # the fused decode step re-ordered this tensor's rows to (gate_0, up_0, gate_1, up_1, ..) in place
            # (`pair_gate_up_rows_`): the module forward (prefill, tests) reads the interleaved output back
            y = self.w1w3(x)
            w1_out, w3_out = y[..., 0::2], y[..., 1::2]
        elif self.fuse_linears:
            # .clone(): the quantized linears return their persistent output buffer by reference
            w1_out, w3_out = self.w1w3(x).split([self.config.intermediate_size, self.config.intermediate_size], dim=-1)
        else:
            w1_out = self.w1(x).clone()
            w3_out = self.w3(x)
        return self.w2(self.act_fn(w1_out) * w3_out)


def _pair_perm(inter: int, device):
    """row order (gate_0, up_0, gate_1, up_1, ..) of a fused [w1; w3] tensor"""
    return torch.stack((torch.arange(inter, device=device), torch.arange(inter, 2 * inter, device=device)), dim=1).reshape(-1)


def pair_gate_up_rows_(m) -> None:
    """Re-order the rows of a fused gate/up Any-Precision linear IN PLACE to interleaved (gate_i, up_i) pairs, the layout
    the GQ_EPI_SILU_PAIRS epilogue consumes (the w1w3 GEMV then writes silu(gate) * up directly).  No second copy of the
    tensor is kept (round 1 held one: +0.94 GB at 8B 2-bit, +3.8 GB at 70B).  The state-dict contract is unchanged:
    `state_dict()` exports the reference layout [w1; w3] (hook below) and `load_state_dict` takes it."""
    if getattr(m, "gq_row_pairs", False):
        return
    perm = _pair_perm(m.out_features // 2, m.qweight.device)
    m.qweight.data = m.qweight.data[:, perm, :].contiguous()
    m.lut.data = m.lut.data[perm].contiguous()
    m.gq_row_pairs = True
    m.gq_realloc_gen = getattr(m, "gq_realloc_gen", 0) + 1  # (new tensors: captured graphs must re-validate, Transformer._alloc_gen)


def _unpair_on_export(module, state_dict, prefix, local_metadata):
    if getattr(module, "gq_row_pairs", False):
        inter = module.out_features // 2
        inv = torch.argsort(_pair_perm(inter, state_dict[prefix + "lut"].device))
        state_dict[prefix + "qweight"] = state_dict[prefix + "qweight"][:, inv, :].contiguous()
        state_dict[prefix + "lut"] = state_dict[prefix + "lut"][inv].contiguous()


def _incoming_is_reference_layout(module, state_dict, prefix, *args, **kwargs):
    # load_state_dict brings [w1; w3]; the native state is rebuilt (and re-paired) lazily.  Only when this module's tensors are
    # actually in the incoming dict: a strict=False load without them leaves the paired rows (and the flag) as they are.
    if prefix + "qweight" in state_dict or prefix + "lut" in state_dict:
        module.gq_row_pairs = False


class TransformerBlock(nn.Module):

    def __init__(self, config: ModelArgs, linear_class=nn.Linear, linear_kwargs=None, fuse_linears=True) -> None:
        super().__init__()
        self.attention = Attention(config, linear_class, linear_kwargs, fuse_linears)
        self.feed_forward = FeedForward(config, linear_class, linear_kwargs, fuse_linears)
        if "llama" in config.model_name.lower():
            self.input_layernorm = RMSNorm(config.dim, config.norm_eps)
            self.post_attention_layernorm = RMSNorm(config.dim, config.norm_eps)
        else:
            raise NotImplementedError

    def forward(self, x: Tensor, input_pos: Tensor, mask: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
        h = x + self.attention(self.input_layernorm(x), mask, input_pos, cos, sin)
        out = self.feed_forward(self.post_attention_layernorm(h))
        return h + out


class Transformer(nn.Module):

    def __init__(self, dtype, config: ModelArgs, linear_class=nn.Linear, linear_kwargs=None, halve_layers=False,
                 fuse_linears=True) -> None:
        super().__init__()
        self.config = config
        self.dtype = dtype
        if halve_layers:
            config.n_layer = config.n_layer // 2
        self.tok_embeddings = nn.Embedding(config.vocab_size, config.dim)
        self.layers = nn.ModuleList(
            TransformerBlock(config, linear_class, linear_kwargs, fuse_linears) for _ in range(config.n_layer))
        self.norm = RMSNorm(config.dim, eps=config.norm_eps)
        self.output = nn.Linear(config.dim, config.vocab_size, bias=False)
        self.max_batch_size = -1
        self.max_seq_length = -1
        self.cache_initialized = False
        self.fuse_linears = fuse_linears
        self._native = None
        # new weights (copied in place or assigned): the native step's buffers / launch plans / row pairing are rebuilt lazily
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module._reset_native())

    def _reset_native(self):
        self._native = None
        self._native_kind_cache = None
        self._alloc_gen = getattr(self, "_alloc_gen", 0) + 1  # captured graphs re-validate their pointers (generate.DecodeGraph.step)

    def _apply(self, fn, *a, **k):
        # .to() / .half() / .cuda() move or re-allocate every tensor a captured decode graph points at
        r = super()._apply(fn, *a, **k)
        self._reset_native()
        return r

    @classmethod
    def from_name(cls, dtype, name: str, linear_class=nn.Linear, linear_kwargs=None, halve_layers=False,
                  fuse_linears=True) -> "Transformer":
        return cls(dtype, ModelArgs.from_name(name), linear_class=linear_class, linear_kwargs=linear_kwargs,
                   halve_layers=halve_layers, fuse_linears=fuse_linears)

    def setup_caches(self, max_batch_size, max_seq_length):
        if self.max_seq_length >= max_seq_length and self.max_batch_size >= max_batch_size:
            return
        head_dim = self.config.dim // self.config.n_head
        max_seq_length = find_multiple(max_seq_length, 8)
        self.max_seq_length = max_seq_length
        self.max_batch_size = max_batch_size
        dtype = self.output.weight.dtype
        device = self.output.weight.device
        for b in self.layers:
            b.attention.kv_cache = KVCache(max_batch_size, max_seq_length, self.config.n_local_heads, head_dim, dtype, device)
        self.causal_mask = torch.tril(torch.ones(self.max_seq_length, self.max_seq_length, dtype=torch.bool, device=device))
        self.rope_cos, self.rope_sin = rope_tables(head_dim, max_seq_length, self.config.rope_base, device, dtype,
                                                   rope_scaling=self.config.rope_scaling)
        self.cache_initialized = True
        self._reset_native()

    def forward(self, idx: Tensor, input_pos: Optional[Tensor] = None) -> Tensor:
        assert self.cache_initialized, "Caches must be initialized first"
        mask = self.causal_mask[None, None, input_pos]
        x = self.tok_embeddings(idx)
        for layer in self.layers:
            x = layer(x, input_pos, mask, self.rope_cos, self.rope_sin)
        x = self.norm(x)
        return self.output(x)

    # ------------------------------------------------------------------------------------------ fused HIP decode step
    def native_ready(self) -> bool:
        return self._native_kind() is not None

    def _native_kind(self):
        """'ap': fused-linear Any-Precision model (<= 4 bit) -> gq_anyprec_gemv_fused chain; 'qtip': unfused QTIP model
        with power-of-two widths -> gq_qtip_linear_in / _out chain; None: the module-by-module forward."""
        from .APLinear import APLinear
        if not (self.cache_initialized and self.output.weight.is_cuda):
            return None
        if self.output.weight.dtype != torch.float16 or self.max_batch_size < 1:
            return None
        kind = getattr(self, "_native_kind_cache", None)
        if kind is not None:
            return kind or None
        kind = ""
        if self.fuse_linears:
            if all(isinstance(m, APLinear) and m.bias is None and m.bitwidth <= 4 and m.in_features % 128 == 0
                   for b in self.layers for m in (b.attention.wqkv, b.attention.wo, b.feed_forward.w1w3, b.feed_forward.w2)):
                kind = "ap"
        elif os.environ.get("GQ_NATIVE_QTIP", "1") != "0":
            from .qtip import QuantizedLinear

            p2 = lambda n: n > 0 and (n & (n - 1)) == 0  # noqa: E731

            def ok(m):
                # widths: power of two (fused transform), or Kf * 2^p with the caller's Hadamard factor table loaded
                # (gq_qtip_transform; e.g. 11008 = 172 * 64, 5120 = 20 * 256)
                def side(nf, K, had):
                    return p2(nf) if K == 1 else (had is not None and p2(nf // K) and nf // K >= 64 and nf <= 32768)
                return (isinstance(m, QuantizedLinear) and m.bias is None and m.has_kernel and int(m.rcp.item()) == 0
                        and side(m.in_features, m.K_left, m.had_left) and side(m.out_features, m.K_right, m.had_right)
                        and 32 <= m.in_features and (m.in_features <= 16384 or m.K_left != 1) and m.out_features <= 32768)
            if self.config.head_dim in (64, 128) and all(
                    ok(m) for b in self.layers for m in (b.attention.wq, b.attention.wk, b.attention.wv, b.attention.wo,
                                                         b.feed_forward.w1, b.feed_forward.w3, b.feed_forward.w2)):
                kind = "qtip"
        self._native_kind_cache = kind
        return kind or None

    def _native_state(self):
        if self._native is None:
            dev = self.output.weight.device
            c = self.config
            f16 = dict(dtype=torch.float16, device=dev)
            self._native = dict(
                x=torch.zeros(c.dim, **f16), h=torch.zeros(c.dim, **f16), y=torch.zeros(c.dim, **f16),
                qkv=torch.zeros((c.n_head + 2 * c.n_local_heads) * c.head_dim, **f16),
                ssq=torch.zeros(_lib.SSQ_SLOTS, dtype=torch.float32, device=dev),  # statistics hand-over slots (gq_hip.h GQ_SSQ_SLOTS)
                # one flag line per query head for the attention heads that run inside the wqkv launch (gq_anyprec_gemv_qkv_rope_attn:
                # zero between launches; one buffer for all layers)
                attn_flags=torch.zeros(c.n_head * _lib.ATTN_FLAG_STRIDE, dtype=torch.int32, device=dev),
                gu=torch.zeros(2 * c.intermediate_size, **f16), logits=torch.zeros(1, 1, c.vocab_size, **f16))
            # long caches: split-KV attention (gq_attn_decode_split), n_split blocks per head + a combine launch; a context of
            # up to 256 positions is still finished by one block per head at run time
            S = self.max_seq_length
            ns = 1 if S <= 1024 else (4 if S <= 2048 else 8)
            # grouped-query models whose wqkv launch rotates q / k (gq_attn_decode_roped): the four query heads of a KV group share a
            # block, so the splits can be as short as one 128-position pass -- n_kv_head x n_split blocks ~ one per CU
            l0 = self.layers[0].attention
            if (S > 1024 and c.n_head % (4 * c.n_local_heads) == 0 and self._native_kind() != "qtip" and os.environ.get("GQ_ATTN_GQA", "1") != "0"
                    and _lib.lib().gq_anyprec_qkv_rope_supported(l0.wqkv.out_features, c.dim, l0.wqkv.bitwidth, c.head_dim)):
                ns = max(4, min(32, (S + 127) // 128, 256 // max(1, c.n_head // 4)))
            ns = int(os.environ.get("GQ_ATTN_SPLIT", ns))
            self._native["attn_split"] = ns
            self._native["attn_ws"] = torch.zeros(c.n_head * ns * (c.head_dim + 2), dtype=torch.float32, device=dev) if ns > 1 else None
            # Gate/up pairing (GQ_EPI_SILU_PAIRS): every fused w1w3 tensor is re-ordered in place to (gate_0, up_0, gate_1,
            # up_1, ..) rows so that the w1w3 GEMV writes silu(gate) * up directly (model.py:266 of the reference) and w2
            # reads a plain vector; state_dict() still exports the reference layout (pair_gate_up_rows_).
            self._native["pairs"] = False
            self._native["ap_ws"] = None
            if self._native_kind() == "qtip":
                self._native_qtip_state(self._native)
            else:
                # workspace of the down projection where the library splits its rows along K over blocks (K > 16384: 70B)
                wb = max(int(_lib.lib().gq_anyprec_gemv_fused_ws_bytes(c.dim, c.intermediate_size, b.feed_forward.w2.bitwidth, 1)) for b in self.layers)
                if wb:
                    self._native["ap_ws"] = torch.zeros(wb // 4, dtype=torch.float32, device=dev)
            if self._native_kind() == "qtip":
                pass
            elif os.environ.get("GQ_NATIVE_PAIRS", "1") != "0":
                for b in self.layers:
                    pair_gate_up_rows_(b.feed_forward.w1w3)
                self._native["pairs"] = True
            else:
                assert not any(getattr(b.feed_forward.w1w3, "gq_row_pairs", False) for b in self.layers), \
                    "GQ_NATIVE_PAIRS=0 on a model whose gate/up rows were already paired"
        return self._native

    def _native_qtip_state(self, st):
        """launch plans of the QTIP linears, per layer four groups of linears that share an input -- (q, k, v), (o),
        (gate, up), (down) -- each a list of (entry point name, argument tuple) built once: SU as fp32, SV * 32 as fp32 (the
        values BitshiftLinear.forward multiplies with, bitshift.py:441,470), q/k/v landing in the packed buffer the
        attention kernel reads.  Per side of a linear: power-of-two width -> the fused kernels (gq_qtip_linear_in / _out);
        width with a Hadamard factor -> gq_qtip_transform around the bare matvec (GQ_QPRO_PRETRANSFORMED)."""
        c = self.config
        dev = self.output.weight.device
        kv = c.n_local_heads * c.head_dim
        st["g"] = torch.zeros(c.intermediate_size, dtype=torch.float16, device=dev)
        st["u"] = torch.zeros(c.intermediate_size, dtype=torch.float16, device=dev)
        mmax = max(c.dim, c.intermediate_size)
        # [linear][split-K part][M]; rows 3 / 4: the sums of wo / down while their transform-out is folded into the next launch
        st["y32"] = torch.zeros(5, 4 * mmax, dtype=torch.float32, device=dev)
        st["xs16"] = torch.zeros(3, mmax, dtype=torch.float16, device=dev)     # transformed inputs (factor widths)
        keep = []    # fp32 copies and descriptor arrays the plans point into
        tables = {}  # one fp32 device copy per distinct Hadamard factor table (every layer's module holds its own buffer)

        def f32(t, mul=1.0):
            t = (t.detach().to(dev).float() * mul).contiguous()
            keep.append(t)
            return t.data_ptr()

        def table(t):
            key = (tuple(t.shape), hash(t.detach().float().cpu().numpy().tobytes()))
            if key not in tables:
                tables[key] = f32(t)
            return tables[key]

        def table16(t):  # (gq_qtip_mlp_mid reads its factor tables as fp16: +-1 entries, checked by pm1)
            t16 = t.detach().to(dev).half().contiguous()
            key = ("h", tuple(t16.shape), hash(t16.cpu().numpy().tobytes()))
            if key not in tables:
                keep.append(t16)
                tables[key] = t16.data_ptr()
            return tables[key]

        def pm1(t):
            return bool((t.detach().float().abs() == 1.0).all())

        def ksplit(m):  # K ranges per band: the split with the fewest band-equivalents per block (wo, down: 128 bands on 256 units -> 2)
            if os.environ.get("GQ_QTIP_KSPLIT", "1") == "0":
                return 1
            return int(_lib.lib().gq_qtip_plan_ksplit(1, (ctypes.c_uint32 * 1)(m.out_features), m.in_features, 2))

        def ksplit_group(mods):  # the same for linears that share a launch (q / k / v: 384 bands on 256 units -> 2 K ranges, 3 rounds of half a band)
            if os.environ.get("GQ_QTIP_KSPLIT", "1") == "0" or os.environ.get("GQ_QTIP_KSPLIT_GROUP", "1") == "0":
                return 1
            return int(_lib.lib().gq_qtip_plan_ksplit(len(mods), (ctypes.c_uint32 * len(mods))(*[m.out_features for m in mods]), mods[0].in_features, 4))

        y32, xs16 = st["y32"], st["xs16"]

        def group(mods, xp, x2p, normw, pro, outs, resid, prev=None, defer=None, out_to=None, parts_ok=False):
            """launches of one group: mods share the input vector xp (x2p for silu*mul); outs[i] fp16 destinations.
            prev: GqQtipOut of the linear that PRODUCES xp, its transform-out folded into this group's first launch (which then
            stores xp itself); defer = row of y32: this group's own transform-out is left to the consumer (-> returned descriptor)"""
            R, K = mods[0].K, mods[0].in_features
            plan = []
            # split-K partial sums are added by the consumer of the sums: the fused transform-out (single linears with a power-of-two
            # output width), the attention launch with the q / k / v transform-out folded in (out_to), gq_qtip_mlp_mid (parts_ok)
            if len(mods) == 1:
                ks = ksplit(mods[0]) if mods[0].K_right == 1 else 1
            elif mods[0].K_left == 1 and ((out_to is not None and all(m.K_right == 1 for m in mods)) or parts_ok):
                ks = ksplit_group(mods)
            else:
                ks = 1
            ysl = [y32[defer]] if defer is not None else [y32[i] for i in range(len(mods))]
            if mods[0].K_left == 1:  # fused transform-in + matvec, all linears in one launch
                arr = (_lib.GqQtipIn * len(mods))(*[_lib.GqQtipIn(m.trellis.data_ptr(), f32(m.SU), m.tlut.data_ptr(), ysl[i].data_ptr(), m.out_features)
                                                    for i, m in enumerate(mods)])
                keep.append(arr)
                if prev is not None:
                    parr = (_lib.GqQtipOut * 1)(prev)
                    keep.append(parr)
                    plan.append(("gq_qtip_linear_in", (None, None, normw, c.norm_eps, pro, K, R, len(mods), arr, 1, parr, ks)))
                else:
                    plan.append(("gq_qtip_linear_in", (xp, x2p, normw, c.norm_eps, pro, K, R, len(mods), arr, 0, None, ks)))
            else:  # factor transform of the shared input (one launch), then the bare matvec per linear
                xf = (_lib.GqQtipXf * len(mods))(*[_lib.GqQtipXf(None, f32(m.SU), table(m.had_left), None, xs16[i].data_ptr())
                                                   for i, m in enumerate(mods)])
                keep.append(xf)
                plan.append(("gq_qtip_transform", (1, xp, x2p, normw, c.norm_eps, pro, len(mods), xf, K, mods[0].K_left, 1)))
                assert prev is None
                for i, m in enumerate(mods):
                    arr = (_lib.GqQtipIn * 1)(_lib.GqQtipIn(m.trellis.data_ptr(), None, m.tlut.data_ptr(), ysl[i].data_ptr(), m.out_features))
                    keep.append(arr)
                    plan.append(("gq_qtip_linear_in", (xs16[i].data_ptr(), None, None, 0.0, 3, K, R, 1, arr, 0, None, ks)))
            p2 = [i for i, m in enumerate(mods) if m.K_right == 1]
            fac = [i for i, m in enumerate(mods) if m.K_right != 1]
            # One launch for transform-in + matvec + transform-out (gq_qtip_linear: the block that finishes a linear last
            # transforms it): every linear of the group has a power-of-two output width, nothing is folded or deferred
            if (one_launch and defer is None and prev is None and not fac and plan and plan[-1][0] == "gq_qtip_linear_in"
                    and all(m.out_features <= 16384 for m in mods)):
                name, args = plan.pop()
                fin = (_lib.GqQtipOut * len(mods))(*[_lib.GqQtipOut(ysl[i].data_ptr(), f32(mods[i].SV, 32.0), resid, outs[i], mods[i].out_features, ks)
                                                     for i in range(len(mods))])
                ctr = torch.zeros(4, dtype=torch.int32, device=dev)
                keep.extend([fin, ctr])
                plan.append(("gq_qtip_linear", args[:9] + (fin, ks, ctr.data_ptr())))
                return plan
            if out_to is not None and not fac:
                # the consumer rebuilds the outputs itself (q / k / v: gq_attn_decode_qtip, the transform-out inside the attention
                # launch): descriptors instead of the gq_qtip_linear_out launch
                arr = (_lib.GqQtipOut * len(mods))(*[_lib.GqQtipOut(ysl[i].data_ptr(), f32(mods[i].SV, 32.0), None, outs[i], mods[i].out_features, ks)
                                                     for i in range(len(mods))])
                keep.append(arr)
                out_to.append(arr)
                return plan
            if defer is not None:  # (a single linear with a power-of-two output width)
                desc = _lib.GqQtipOut(ysl[0].data_ptr(), f32(mods[0].SV, 32.0), resid, outs[0], mods[0].out_features, ks)
                arr = (_lib.GqQtipOut * 1)(desc)
                keep.append(arr)
                return plan, desc, [("gq_qtip_linear_out", (1, arr))]
            if p2:
                arr = (_lib.GqQtipOut * len(p2))(*[_lib.GqQtipOut(ysl[i].data_ptr(), f32(mods[i].SV, 32.0), resid, outs[i], mods[i].out_features, ks)
                                                   for i in p2])
                keep.append(arr)
                # M / 128 blocks per linear instead of one (GQ_QTIP_OUT_SEG, default ON; equal up to fp32 rounding): 2.9 vs 4.6 us
                seg = out_seg and all(128 <= mods[i].out_features <= 8192 for i in p2)
                plan.append(("gq_qtip_linear_out_seg" if seg else "gq_qtip_linear_out", (len(p2), arr)))
            for Kf, M in sorted({(mods[i].K_right, mods[i].out_features) for i in fac}):
                idx = [i for i in fac if (mods[i].K_right, mods[i].out_features) == (Kf, M)]
                xf = (_lib.GqQtipXf * len(idx))(*[_lib.GqQtipXf(ysl[i].data_ptr(), f32(mods[i].SV, 32.0), table(mods[i].had_right), resid, outs[i])
                                                  for i in idx])
                keep.append(xf)
                plan.append(("gq_qtip_transform", (0, None, None, None, 0.0, 0, len(idx), xf, M, Kf, 0)))
            return plan

        x, h, y, qkv = st["x"], st["h"], st["y"], st["qkv"]
        e = qkv.element_size()
        # Folding (GQ_QTIP_FOLD=1, default OFF): the transform-out of wo (+ residual) is rebuilt by the gate / up launch, that of
        # down (+ residual) by the NEXT layer's q / k / v launch (gq_qtip_linear_in with n_prev = 1; bit-identical, the vector is
        # stored once for the residual stream): two launches per layer less, but every one of the 256 blocks repeats the
        # 4096-point transform and takes the slower prologue path -- measured 365 vs 385 tokens/s on the Llama-2-7b shape
        # (400 vs 422 with a power-of-two MLP), so it stays off.  Needs power-of-two widths on both sides of the fold.
        fold = os.environ.get("GQ_QTIP_FOLD", "0") != "0"
        # GQ_QTIP_ONE_LAUNCH=1 (default OFF): transform-out inside the matvec launch (gq_qtip_linear, see group()).  Bit-identical,
        # three launches per layer less -- and measured 302 vs 384 tokens/s on the Llama-2-7b shape: the device-scope release /
        # acquire fences around the per-linear counter (L2 write-back + invalidate on 8 XCDs) cost ~7 us per launch, more than the
        # launch they save.
        one_launch = os.environ.get("GQ_QTIP_ONE_LAUNCH", "0") != "0"
        out_seg = os.environ.get("GQ_QTIP_OUT_SEG", "1") != "0"
        mlp_mid = os.environ.get("GQ_QTIP_MLP_MID", "1") != "0"
        # GQ_QTIP_ATTN_FOLD (default ON): the transform-out of q / k / v runs inside the attention launch -- every head block needs
        # head_dim of the outputs: the segments combined with the signs of its row, then one head_dim-point transform (equal to
        # gq_qtip_linear_out up to fp32 rounding) --: one launch (4.8 us) per layer less.  Needs power-of-two q / k / v widths.
        # (gq_attn_decode_qtip serves n_head * head_dim <= 8192, a power of two; wider models keep gq_qtip_linear_out + attention)
        qw = c.n_head * c.head_dim
        attn_fold = (os.environ.get("GQ_QTIP_ATTN_FOLD", "1") != "0" and not one_launch and c.head_dim in (64, 128)
                     and qw <= 8192 and (qw & (qw - 1)) == 0)
        layers = []
        prev_down = None
        for b in self.layers:
            at, ff = b.attention, b.feed_forward
            qkv_outs = [qkv.data_ptr(), qkv.data_ptr() + c.dim * e, qkv.data_ptr() + (c.dim + kv) * e]
            can_o = fold and at.wo.K_right == 1 and ff.w1.K_left == 1
            # GQ_QTIP_MLP_MID (default ON): a factor MLP width n = Kf * 64 (Llama-2-7b: 172 * 64) -- the two gq_qtip_transform launches
            # between the matvecs of gate / up and down (output side, then input side with silu * up) become ONE launch that works
            # column by column (gq_qtip_mlp_mid), the 64-point row transforms of the input side move into the prologue of down's
            # matvec launch (gq_qtip_linear_in_rows).  gate / up bit-identical, the input of down equal up to fp32 rounding.
            w1, w3, w2 = ff.w1, ff.w3, ff.w2
            n_mlp, Kf = w2.in_features, w2.K_left
            mid_ok = (mlp_mid and not can_o and not one_launch and Kf != 1 and w1.K_right == Kf and w3.K_right == Kf and n_mlp == Kf * 64
                      and Kf <= 176 and Kf % 4 == 0 and w1.K_left == 1 and w3.K_left == 1 and w2.K_right == 1
                      and w1.out_features == n_mlp and w3.out_features == n_mlp
                      and torch.equal(w1.had_right, w3.had_right) and pm1(w1.had_right) and pm1(w2.had_left))
            can_d = fold and ff.w2.K_right == 1 and at.wq.K_left == 1
        
            fold_here = attn_fold and all(m.K_right == 1 for m in (at.wq, at.wk, at.wv))
            desc = [] if fold_here else None
            d = dict(qkv=group([at.wq, at.wk, at.wv], x.data_ptr(), None, b.input_layernorm.weight.data_ptr(), 1, qkv_outs, None, out_to=desc))
            # (q / k / v of this layer with the previous layer's down folded in; the first layer of a range takes the plain form)
            desc_f = [] if fold_here else None
            d["qkv_f"] = group([at.wq, at.wk, at.wv], None, None, b.input_layernorm.weight.data_ptr(), 1, qkv_outs, None, prev=prev_down, out_to=desc_f) \
                if prev_down is not None else None
            d["attn_qt"] = desc[0] if fold_here else None
            if can_o:
                d["o"], desc_o, _ = group([at.wo], y.data_ptr(), None, None, 0, [h.data_ptr()], x.data_ptr(), defer=3)
                d["gu"] = group([ff.w1, ff.w3], None, None, b.post_attention_layernorm.weight.data_ptr(), 1,
                                [st["g"].data_ptr(), st["u"].data_ptr()], None, prev=desc_o)
            else:
                d["o"] = group([at.wo], y.data_ptr(), None, None, 0, [h.data_ptr()], x.data_ptr())
                d["gu"] = group([ff.w1, ff.w3], h.data_ptr(), None, b.post_attention_layernorm.weight.data_ptr(), 1,
                                [st["g"].data_ptr(), st["u"].data_ptr()], None, parts_ok=mid_ok)
            if can_d:
                d["d"], prev_down, d["d_out"] = group([ff.w2], st["g"].data_ptr(), st["u"].data_ptr(), None, 2, [x.data_ptr()], h.data_ptr(), defer=4)
            else:
                d["d"], prev_down, d["d_out"] = group([ff.w2], st["g"].data_ptr(), st["u"].data_ptr(), None, 2, [x.data_ptr()], h.data_ptr()), None, None
            if mid_ok:
                assert (d["gu"][-1][0] == "gq_qtip_transform" and d["d"][0][0] == "gq_qtip_transform"
                        and d["d"][1][0] == "gq_qtip_linear_in" and d["d"][1][1][4] == 3), "unexpected launch plan of a factor-width MLP"
                ks_gu = d["gu"][0][1][11]  # (split-K parts of the gate / up sums)
                if "z32" not in st:
                    st["z32"] = torch.zeros(mmax, dtype=torch.float32, device=dev)
                mid = _lib.GqQtipMid(y32[0].data_ptr(), y32[1].data_ptr(), f32(w1.SV, 32.0), f32(w3.SV, 32.0),
                                     table16(w1.had_right.t()), f32(w2.SU), table16(w2.had_left), st["z32"].data_ptr(), None, None)
                keep.append(mid)
                a_in = d["d"][1][1]  # (xs16, None, None, 0.0, 3, K, R, 1, arr, 0, None, ks)
                d["gu"] = d["gu"][:-1]
                d["d"] = [("gq_qtip_mlp_mid", (ctypes.pointer(mid), ks_gu, n_mlp, Kf)),
                          ("gq_qtip_linear_in_rows", (st["z32"].data_ptr(), n_mlp, 64, a_in[6], 1, a_in[8], a_in[11]))] + d["d"][2:]
            layers.append(d)
        # Round 5 (GQ_QTIP_PRE=1; default OFF -- measured 421 vs 427 tokens/s on Llama-2-7b, profiles/r05_qtip_pre.txt: the one-block
        # launch grows by more than the 256-block launch shrinks, whose prologue ran under its first tile requests anyway): the
        # transform-out launch of wo / down ALSO runs the transform-in of the linears that read
        # its output through an RMSNorm -- gate / up, the next layer's q / k / v -- one block per consumer (gq_qtip_linear_out_in), and
        # their matvec launch takes the pre-transformed vectors as they are: its 256 blocks no longer repeat RMSNorm . SU . Hadamard.
        # Needs power-of-two widths on that edge and the plain launch forms (no folding, no one-launch form).
        st["xt"] = torch.zeros(3, c.dim, dtype=torch.float16, device=dev)
        pre_on = os.environ.get("GQ_QTIP_PRE", "0") != "0" and not fold and not one_launch and 256 <= c.dim <= 8192 and (c.dim & (c.dim - 1)) == 0

        def with_pre(plan_out, plan_in, normw, mods):
            """(plan of the producer with its last launch -- the transform-out -- replaced, plan of the consumers' matvec launch on the
            pre-transformed vectors), or None when the launches are not of the plain form"""
            if not (pre_on and plan_out and plan_in and plan_out[-1][0] in ("gq_qtip_linear_out", "gq_qtip_linear_out_seg") and plan_out[-1][1][0] == 1
                    and plan_in[0][0] == "gq_qtip_linear_in" and plan_in[0][1][4] == 1 and plan_in[0][1][9] == 0
                    and all(m.K_left == 1 and m.in_features == c.dim for m in mods)):
                return None
            desc = plan_out[-1][1][1]  # GqQtipOut array of one element
            n = len(mods)
            su = (ctypes.c_void_p * n)(*[f32(m.SU) for m in mods])
            xt = (ctypes.c_void_p * n)(*[st["xt"][i].data_ptr() for i in range(n)])
            a_in = plan_in[0][1]  # (xp, x2p, normw, eps, pro, K, R, n, arr, 0, None, ks)
            arr = (_lib.GqQtipIn * n)(*[_lib.GqQtipIn(a_in[8][i].trellis, st["xt"][i].data_ptr(), a_in[8][i].tlut, a_in[8][i].y32, a_in[8][i].M) for i in range(n)])
            keep.extend([su, xt, arr])
            return (plan_out[:-1] + [("gq_qtip_linear_out_in", (desc, normw, c.norm_eps, n, su, xt))],
                    [("gq_qtip_linear_in", (None, None, None, 0.0, 3, a_in[5], a_in[6], n, arr, 0, None, a_in[11]))] + plan_in[1:])

        for li, (d, b) in enumerate(zip(layers, self.layers)):
            at, ff = b.attention, b.feed_forward
            r = with_pre(d["o"], d["gu"], b.post_attention_layernorm.weight.data_ptr(), [ff.w1, ff.w3])
            if r is not None:
                d["o"], d["gu"] = r
            d["d_pre"] = d["qkv_pre"] = None
        for li in range(len(layers) - 1):
            nb = self.layers[li + 1]
            r = with_pre(layers[li]["d"], layers[li + 1]["qkv"], nb.input_layernorm.weight.data_ptr(), [nb.attention.wq, nb.attention.wk, nb.attention.wv])
            if r is not None and layers[li]["d_out"] is None:
                layers[li]["d_pre"], layers[li + 1]["qkv_pre"] = r
        st["qtip_layers"] = layers
        st["qtip_keep"] = keep

    def _native_layers_qtip(self, x: Tensor, pos: Tensor, l0: int, l1: int, slot: int = 0):
        """one decode step of layers [l0, l1) of an unfused QTIP model (A = transform-in + trellis matvec, B = transform-out):
        A(q,k,v | RMSNorm) B(q,k,v) attention A(o) B(o + residual) A(gate,up | RMSNorm) B(gate,up) A(down | silu*mul)
        B(down + residual) -- 9 launches per layer with power-of-two widths; a width with a Hadamard factor replaces the A / B
        on its side by gq_qtip_transform (+ the bare matvec): 11 launches for Llama-2-7b / 70b (MLP width only)."""
        L = _lib.lib()
        sp = _lib.current_stream_ptr()
        c = self.config
        b = self._native_state()
        assert x.data_ptr() == b["x"].data_ptr(), "the QTIP launch plans are bound to the model's own hidden-state buffer"
        y, qkv = b["y"], b["qkv"]
        ck = _lib.check
        scale = 1.0 / math.sqrt(c.head_dim)
        kv_stride = c.n_local_heads * self.max_seq_length * c.head_dim * 2

        def run(plan):
            for name, args in plan:
                ck(getattr(L, name)(*args, sp), name)

        pending = None  # transform-out of the previous layer's down projection, not yet run
        pre_in = False  # the previous layer's down launch left this layer's q / k / v inputs pre-transformed
        for li in range(l0, l1):
            d, at = b["qtip_layers"][li], self.layers[li].attention
            if pre_in:
                run(d["qkv_pre"])
            elif pending is not None and d["qkv_f"] is not None:
                run(d["qkv_f"])  # (rebuilds and stores the hidden state itself)
            else:
                if pending is not None:
                    run(pending)
                run(d["qkv"])
            if d["attn_qt"] is not None:
                ck(L.gq_attn_decode_qtip(d["attn_qt"], pos.data_ptr(), self.rope_cos.data_ptr(), self.rope_sin.data_ptr(),
                                         at.kv_cache.k_cache.data_ptr() + slot * kv_stride, at.kv_cache.v_cache.data_ptr() + slot * kv_stride,
                                         y.data_ptr(), c.n_head, c.n_local_heads, c.head_dim, self.max_seq_length, scale, b["attn_split"],
                                         b["attn_ws"].data_ptr() if b["attn_ws"] is not None else None, sp), "attn_qtip")
            else:
                ck(L.gq_attn_decode_split(qkv.data_ptr(), pos.data_ptr(), self.rope_cos.data_ptr(), self.rope_sin.data_ptr(),
                                          at.kv_cache.k_cache.data_ptr() + slot * kv_stride, at.kv_cache.v_cache.data_ptr() + slot * kv_stride,
                                          y.data_ptr(), c.n_head, c.n_local_heads, c.head_dim, self.max_seq_length, scale, b["attn_split"],
                                          b["attn_ws"].data_ptr() if b["attn_ws"] is not None else None, sp), "attn")
            run(d["o"])
            run(d["gu"])
            pre_in = li + 1 < l1 and d["d_pre"] is not None
            run(d["d_pre"] if pre_in else d["d"])
            pending = d["d_out"]
        if pending is not None:
            run(pending)

    def native_embed(self, tok: Tensor, x: Tensor, ssq: Optional[Tensor] = None):
        """x = tok_embeddings[tok]; with `ssq` (the hand-over slots of _native_state) also the statistics of x for layer 0's RMSNorm"""
        _lib.check(_lib.lib().gq_embed_lookup_ho(tok.data_ptr(), self.tok_embeddings.weight.data_ptr(), x.data_ptr(), self.config.dim,
                                                 self.config.vocab_size, ssq.data_ptr() if ssq is not None else None,
                                                 _lib.current_stream_ptr()), "gq_embed_lookup")

    def _handover_plan(self, blk):
        """Statistics hand-over (include/gq_hip.h, round 5) on the two RMSNorm edges of a layer -- (w2 or the embedding) -> wqkv and
        wo -> w1w3: the producer's residual epilogue leaves the partial sums of squares of the hidden state it writes, the consumer's
        RMSNorm prologue adds them instead of exchanging per-wave sums.  An edge is used only when BOTH ends have the form (the plan is
        the library's own dispatch run dry): 8B-class 2-bit models.  OFF by default (GQ_SSQ_HANDOVER=1 turns it on): measured on the
        8B decode it LOSES 1.4 % (855 vs 867 tokens/s, profiles/r05_handover.txt) -- the partial sums come out of memory no earlier
        than the activations themselves, the wave that adds them holds the launch barrier ~700 cycles, and the producers pay 0.1 us."""
        # (not cached: the answer follows gq_set_ap_mode / the environment like the dispatch itself; 4 host calls per layer, paid by
        # eager steps and graph captures only)
        L = _lib.lib()
        c, at, ff = self.config, blk.attention, blk.feed_forward
        on = os.environ.get("GQ_SSQ_HANDOVER", "0") != "0"
        if not on:
            return dict(qkv_in=False, w13=False, w2_out=False)
        qkv_in = on and bool(L.gq_anyprec_handover_plan(at.wqkv.out_features, c.dim, at.wqkv.bitwidth, 1, 0) & 1)
        w13_in = on and bool(L.gq_anyprec_handover_plan(2 * c.intermediate_size, c.dim, ff.w1w3.bitwidth, 1, 4) & 1)
        wo_out = bool(L.gq_anyprec_handover_plan(c.dim, c.dim, at.wo.bitwidth, 0, 1) & 2)
        w2_out = bool(L.gq_anyprec_handover_plan(c.dim, c.intermediate_size, ff.w2.bitwidth, 0, 1) & 2)
        return dict(qkv_in=qkv_in, w13=w13_in and wo_out, w2_out=w2_out)

    def native_layers(self, x: Tensor, pos: Tensor, l0: int, l1: int, slot: int = 0, ssq_ready: bool = False):
        """layers [l0, l1) of one decode step, in place on the hidden state `x` (fp16 [dim]); `slot` = batch index of
        the KV caches to use (layer-pipelined decode keeps one sequence per slot).  ssq_ready: the hand-over slots hold the
        statistics of `x` (native_embed(..., ssq) ran on it)."""
        if self._native_kind() == "qtip":
            return self._native_layers_qtip(x, pos, l0, l1, slot)
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        c = self.config
        b = self._native_state()
        h, y, qkv, gu, pairs = b["h"], b["y"], b["qkv"], b["gu"], b["pairs"]
        apws = b["ap_ws"]
        if pairs:  # a load_state_dict into a sub-module bypasses _reset_native: the rows must still be paired at launch
            for blk in self.layers[l0:l1]:
                if not getattr(blk.feed_forward.w1w3, "gq_row_pairs", False):
                    pair_gate_up_rows_(blk.feed_forward.w1w3)
                    self._alloc_gen = getattr(self, "_alloc_gen", 0) + 1  # (the re-pair assigned new tensors)
        ck = _lib.check
        scale = 1.0 / math.sqrt(c.head_dim)
        kv_stride = c.n_local_heads * self.max_seq_length * c.head_dim * 2  # bytes per batch slot
        ssq = b["ssq"].data_ptr()
        x_has_ssq = ssq_ready  # the slots hold the statistics of the current x
        for li, blk in enumerate(self.layers[l0:l1]):
            at, ff = blk.attention, blk.feed_forward
            kc, vc = at.kv_cache.k_cache.data_ptr() + slot * kv_stride, at.kv_cache.v_cache.data_ptr() + slot * kv_stride
            ws = b["attn_ws"].data_ptr() if b["attn_ws"] is not None else None
            ho = self._handover_plan(blk)
            nxt = self.layers[l0 + li + 1] if l0 + li + 1 < len(self.layers) else None
            # (w2 writes the statistics only when the NEXT layer's wqkv reads them: the last layer feeds the lm_head's own norm)
            w2_ssq = ssq if (pairs and ho["w2_out"] and nxt is not None and l0 + li + 1 < l1 and self._handover_plan(nxt)["qkv_in"]) else None
            # RoPE + KV-cache write in the epilogue of the wqkv GEMV, attention without them, where the library serves the layer's
            # wqkv that way (fast mode, 2-bit, K <= 4096: csrc/ap_stream.hip); else the two launches of rounds 1-3
            use_ssq = x_has_ssq and ho["qkv_in"]
            if (b["attn_split"] == 1 and not use_ssq
                    and L.gq_anyprec_qkv_rope_attn_supported(at.wqkv.out_features, c.dim, at.wqkv.bitwidth, c.head_dim, c.n_head, c.n_local_heads)):
                # round 6: the attention heads as extra blocks of the wqkv launch (they wait on device flags for q / the new cache row):
                # one launch and one kernel boundary less per layer, outputs bit-identical to the two launches below
                ck(L.gq_anyprec_gemv_qkv_rope_attn(x.data_ptr(), qkv.data_ptr(), at.wqkv.qweight.data_ptr(), at.wqkv.lut.data_ptr(),
                                                   at.wqkv.out_features, c.dim, at.wqkv.bitwidth, blk.input_layernorm.weight.data_ptr(),
                                                   c.norm_eps, pos.data_ptr(), self.rope_cos.data_ptr(), self.rope_sin.data_ptr(), kc, vc,
                                                   c.n_head, c.n_local_heads, c.head_dim, self.max_seq_length, y.data_ptr(), scale,
                                                   b["attn_flags"].data_ptr(), st), "wqkv+rope+attention")
            elif L.gq_anyprec_qkv_rope_supported(at.wqkv.out_features, c.dim, at.wqkv.bitwidth, c.head_dim):
                ck(L.gq_anyprec_gemv_qkv_rope_ho(x.data_ptr(), qkv.data_ptr(), at.wqkv.qweight.data_ptr(), at.wqkv.lut.data_ptr(),
                                                 at.wqkv.out_features, c.dim, at.wqkv.bitwidth, blk.input_layernorm.weight.data_ptr(),
                                                 c.norm_eps, pos.data_ptr(), self.rope_cos.data_ptr(), self.rope_sin.data_ptr(), kc, vc,
                                                 c.n_head, c.n_local_heads, c.head_dim, self.max_seq_length,
                                                 ssq if (x_has_ssq and ho["qkv_in"]) else None, st), "wqkv+rope")
                ck(L.gq_attn_decode_roped(qkv.data_ptr(), pos.data_ptr(), kc, vc, y.data_ptr(), c.n_head, c.n_local_heads, c.head_dim,
                                          self.max_seq_length, scale, b["attn_split"], ws, st), "attn")
            else:
                ck(L.gq_anyprec_gemv_fused(x.data_ptr(), qkv.data_ptr(), at.wqkv.qweight.data_ptr(), at.wqkv.lut.data_ptr(),
                                           at.wqkv.out_features, c.dim, at.wqkv.bitwidth, blk.input_layernorm.weight.data_ptr(),
                                           c.norm_eps, None, 0, st), "wqkv")
                ck(L.gq_attn_decode_split(qkv.data_ptr(), pos.data_ptr(), self.rope_cos.data_ptr(), self.rope_sin.data_ptr(), kc, vc,
                                          y.data_ptr(), c.n_head, c.n_local_heads, c.head_dim, self.max_seq_length, scale, b["attn_split"],
                                          ws, st), "attn")
            x_has_ssq = False
            if pairs:
                w13 = ssq if ho["w13"] else None
                ck(L.gq_anyprec_gemv_fused_ho(y.data_ptr(), h.data_ptr(), at.wo.qweight.data_ptr(), at.wo.lut.data_ptr(), c.dim, c.dim,
                                              at.wo.bitwidth, None, 0.0, x.data_ptr(), 1, None, 0, None, w13, st), "wo")
                ck(L.gq_anyprec_gemv_fused_ho(h.data_ptr(), gu.data_ptr(), ff.w1w3.qweight.data_ptr(), ff.w1w3.lut.data_ptr(), 2 * c.intermediate_size,
                                              c.dim, ff.w1w3.bitwidth, blk.post_attention_layernorm.weight.data_ptr(), c.norm_eps, None, 4,
                                              None, 0, w13, None, st), "w1w3")
                ck(L.gq_anyprec_gemv_fused_ho(gu.data_ptr(), x.data_ptr(), ff.w2.qweight.data_ptr(), ff.w2.lut.data_ptr(), c.dim,
                                              c.intermediate_size, ff.w2.bitwidth, None, 0.0, h.data_ptr(), 1,
                                              apws.data_ptr() if apws is not None else None, apws.numel() * 4 if apws is not None else 0,
                                              None, w2_ssq, st), "w2")
                x_has_ssq = w2_ssq is not None
                continue
            ck(L.gq_anyprec_gemv_fused(y.data_ptr(), h.data_ptr(), at.wo.qweight.data_ptr(), at.wo.lut.data_ptr(), c.dim, c.dim,
                                       at.wo.bitwidth, None, 0.0, x.data_ptr(), 1, st), "wo")
            ck(L.gq_anyprec_gemv_fused(h.data_ptr(), gu.data_ptr(), ff.w1w3.qweight.data_ptr(), ff.w1w3.lut.data_ptr(),
                                       2 * c.intermediate_size, c.dim, ff.w1w3.bitwidth,
                                       blk.post_attention_layernorm.weight.data_ptr(), c.norm_eps, None, 0, st), "w1w3")
            ck(L.gq_anyprec_gemv_fused(gu.data_ptr(), x.data_ptr(), ff.w2.qweight.data_ptr(), ff.w2.lut.data_ptr(), c.dim,
                                       c.intermediate_size, ff.w2.bitwidth, None, 0.0, h.data_ptr(), 1 | 2, st), "w2")

    def native_head(self, x: Tensor) -> Tensor:
        b = self._native_state()
        c = self.config
        _lib.check(_lib.lib().gq_dense_gemv_f16(x.data_ptr(), self.output.weight.data_ptr(), b["logits"].data_ptr(), c.vocab_size,
                                                c.dim, self.norm.weight.data_ptr(), c.norm_eps, _lib.current_stream_ptr()), "lm_head")
        return b["logits"]

    # ------------------------------------------------------------------------------------------------ prompt pass
    def prefill_ready(self, idx: Tensor) -> bool:
        """the HIP prompt pass serves what the fused decode step serves for Any-Precision models: fp16, batch 1, on the GPU"""
        return (self._native_kind() == "ap" and idx.is_cuda and idx.numel() > 1 and (idx.dim() == 1 or idx.shape[0] == 1)
                and self.config.head_dim % 16 == 0 and self.config.dim % 8 == 0 and self.config.dim <= 16384
                and self.config.intermediate_size % 8 == 0)

    def prefill_native(self, idx: Tensor, input_pos: Tensor, start: int = 0, last_only: bool = True) -> Tensor:
        """The prompt pass (`Transformer.forward` with seq_len > 1, inference/model.py:206-266 semantics) with the element-wise
        steps between the linears as ONE HIP launch each (csrc/prefill.hip: RMSNorm rows, RoPE + KV-cache write, silu * up)
        instead of ~45 eager tensor ops per layer, the linears through `APLinear.forward` (fused prefill GEMM / split K /
        the reference's two steps by size), attention over the keys [0, start + S) only instead of the whole cache.
        `input_pos` must be arange(start, start + S) (what generate() passes; `start` is the host copy of its first element).
        Same fp16 rounding points as the module forward: logits agree up to the summation order of the fp32 sums.
        last_only: logits of the last prompt token only, [1, 1, V] -- all generate() samples from -- else [1, S, V]."""
        from . import _lib
        L = _lib.lib()
        cfg = self.config
        S, D, H, Hkv, hd, inter = idx.numel(), cfg.dim, cfg.n_head, cfg.n_local_heads, cfg.head_dim, cfg.intermediate_size
        T = start + S
        assert self.prefill_ready(idx) and input_pos.numel() == S and input_pos.dtype == torch.int32 and T <= self.max_seq_length
        dev = idx.device
        x = self.tok_embeddings(idx.view(1, S)).view(S, D).contiguous()
        xn = torch.empty_like(x)
        q = torch.empty((H, S, hd), dtype=torch.float16, device=dev)
        hbuf = torch.empty((S, inter), dtype=torch.float16, device=dev)
        mask = None if start == 0 else self.causal_mask[None, None, input_pos.long(), :T]
        rep = H // Hkv
        pending = None  # the previous block's MLP output: its residual add rides in the next RMSNorm launch
        with torch.cuda.device(dev):
            st = _lib.current_stream_ptr()  # (the model's device's current stream: inside the guard)
            for b in self.layers:
                att, ff = b.attention, b.feed_forward
                _lib.check(L.gq_rmsnorm_rows(x.data_ptr(), pending.data_ptr() if pending is not None else None, b.input_layernorm.weight.data_ptr(), xn.data_ptr(),
                                             S, D, b.input_layernorm.eps, st), "gq_rmsnorm_rows")
                qkv = att.wqkv(xn.view(1, S, D)).view(S, -1)
                kc, vc = att.kv_cache.k_cache, att.kv_cache.v_cache
                _lib.check(L.gq_rope_cache_rows(qkv.data_ptr(), input_pos.data_ptr(), self.rope_cos.data_ptr(), self.rope_sin.data_ptr(), q.data_ptr(),
                                                kc.data_ptr(), vc.data_ptr(), S, H, Hkv, hd, kc.shape[2], st), "gq_rope_cache_rows")
                y = _sdpa_gqa(q.unsqueeze(0), kc[:1, :, :T], vc[:1, :, :T], mask, rep)
                y = y.transpose(1, 2).reshape(1, S, H * hd)
                o = att.wo(y).view(S, D)
                _lib.check(L.gq_rmsnorm_rows(x.data_ptr(), o.data_ptr(), b.post_attention_layernorm.weight.data_ptr(), xn.data_ptr(), S, D,
                                             b.post_attention_layernorm.eps, st), "gq_rmsnorm_rows")
                gu = ff.w1w3(xn.view(1, S, D)).view(S, 2 * inter)
                _lib.check(L.gq_silu_mul_rows(gu.data_ptr(), hbuf.data_ptr(), S, inter, 1 if getattr(ff.w1w3, "gq_row_pairs", False) else 0, st), "gq_silu_mul_rows")
                pending = ff.w2(hbuf.view(1, S, inter)).view(S, D)
        x = x + pending
        x = x[-1:] if last_only else x
        return self.output(self.norm(x)).view(1, -1, cfg.vocab_size)

    def decode_native(self, tok: Tensor, pos: Tensor) -> Tensor:
        """One bs=1 decode step.  tok, pos: int32 device tensors with one element.  Returns logits fp16 [1,1,V]
        (a persistent buffer, like the quantized linears' outputs).  Enqueues on the current stream only."""
        assert tok.dtype == torch.int32 and pos.dtype == torch.int32 and tok.is_cuda and pos.is_cuda
        with torch.cuda.device(self.output.weight.device):  # (launches go to the model's device's current stream)
            st = self._native_state()
            x = st["x"]
            ho0 = self._native_kind() == "ap" and self._handover_plan(self.layers[0])["qkv_in"]
            self.native_embed(tok, x, st["ssq"] if ho0 else None)
            self.native_layers(x, pos, 0, len(self.layers), ssq_ready=bool(ho0))
            return self.native_head(x)
