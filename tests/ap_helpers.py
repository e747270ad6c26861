"""Shared helpers of the Any-Precision GEMV GPU parity tests (fast / plane-MFMA mode envelopes, reference-rounding
restatements of the decode step's element-wise ops, LNQ-like synthetic layers)."""
import os

import numpy as np


def _fast(force_plane=True, local=1):
    """Fast mode.  By default the dispatcher sends only the shapes on which the plane-MFMA kernel wins to it
    (DESIGN.md section 7); the parity tests of that kernel lift the thresholds so that every shape runs on it.
    local = 0 keeps the shapes that would run the local-image variant (<= 16 rows per CU, 2/3-bit, no RMSNorm) on the
    shared-image kernel, so both are checked on the same inputs."""
    from guidedquant_amd import _lib
    _lib.check(_lib.lib().gq_set_ap_mode(0), "gq_set_ap_mode")
    if force_plane:
        os.environ["GQ_PL_MIN_MWEIGHTS"] = "0"
        os.environ["GQ_PL_MAX_BITS"] = "4"
        os.environ["GQ_PL_LOCAL"] = str(local)
        _lib.lib().gq_reset_env_cache()


def _check_fast(got, x, q, lut, bits, oracle, rows=None, nround=None):
    """Fast (plane-MFMA) mode.  The reference kernel accumulates in fp16 and is itself ~1e-3 (rms, relative) away
    from the exact product, so a kernel that is MORE accurate cannot be elementwise within 1e-3 of it.  What is
    asserted instead, per element:
      (a) accuracy: |got - exact| <= one fp16 rounding of the exact value + 1e-5 * sum|w||x|  (fp32-class);
      (b) parity:   got is as close to the reference-order result as the correctly rounded exact result is,
                    |got - ref| <= |fp16(exact) - ref| + 2 ulp + 1e-5 * sum|w||x|, and normwise
                    ||got - ref|| <= 1.05 * ||fp16(exact) - ref|| (+eps): all of the distance to the reference is
                    the reference's own fp16 accumulation error (anyprec.cu:495-512);
      (c) shapes the fast path does not serve (K % 256 != 0 or K > 32768) fall back to the exact kernels: bit-identical;
          16384 < K <= 32768 runs as two K-halves chained through the residual epilogue (two fp16 roundings).
    No allowance for the dynamic range of x: elements above 8 x rms are taken out of the MFMA image and multiplied on their
    own (ap_plane.hip hot_step), so massive-activation channels cost their group mates nothing (round 2 needed a slack term
    here); test_hot_channels_* pins that against sum|w||x| of the NON-hot elements."""
    if rows is not None:
        q = np.ascontiguousarray(q[:, rows, :])
        lut = lut[rows]
        got = got[rows]
    K = q.shape[2] * 32
    ref16h = oracle.ap_gemv_f16(x, q, lut, bits)[0]
    if K % 256 or K > 32768:
        assert np.array_equal(got.view(np.uint16), ref16h.view(np.uint16))
        return
    # 16384 < K <= 32768 is served as two K-halves, the second added to the fp16 result of the first: two roundings
    # (with a workspace -- gq_anyprec_gemv_fused_ws -- the K slices meet in fp32 and are rounded once: the caller passes nround = 1)
    if nround is None:
        nround = 2.0 if K > 16384 else 1.0
    y64 = oracle.ap_gemv_f64(x, q, lut, bits)[0]
    ref16 = ref16h.astype(np.float64)
    W = np.abs(oracle.ap_dequant(q, lut, bits).astype(np.float64))
    scale = W @ np.abs(np.asarray(x, dtype=np.float64).reshape(-1))
    g = got.astype(np.float64)
    err_exact = np.abs(g - y64)
    assert (err_exact <= nround * 2.0**-11 * np.abs(y64) * 1.001 + 1e-5 * scale + 1e-7).all(), (err_exact / (scale + 1e-30)).max()
    e16 = y64.astype(np.float16).astype(np.float64)
    ulp = np.maximum(np.abs(np.spacing(y64.astype(np.float16))).astype(np.float64), 2.0**-24)
    assert (np.abs(g - ref16) <= np.abs(e16 - ref16) + 2 * nround * ulp + 1e-5 * scale).all()
    assert np.linalg.norm(g - ref16) <= (1.05 if nround == 1.0 else 1.15) * np.linalg.norm(e16 - ref16) + 1e-6 * np.linalg.norm(scale) + 1e-7


def check_nonhot_accuracy(got, x, hot, q, lut, bits, oracle, tol=2e-5, nround=1.0):
    """|got - exact| in excess of the fp16 output rounding, in units of sum|w||x| over the NON-hot elements (the criterion of
    tools/plane_dynrange_probe.py): the elements next to a massive channel keep fp32-class accuracy"""
    y64 = oracle.ap_gemv_f64(x, q, lut, bits)[0]
    xs = np.abs(np.asarray(x, dtype=np.float64).reshape(-1)).copy()
    xs[hot] = 0
    base = np.abs(oracle.ap_dequant(q, lut, bits).astype(np.float64)) @ xs
    # (nround: fp16 roundings on the way -- 2 for the K > 16384 split, whose first half is rounded before the second is added)
    over = np.maximum(np.abs(got.astype(np.float64) - y64) - nround * 2.0**-11 * np.abs(y64) * 1.001, 0)
    assert (over <= tol * base + 1e-9).all(), (over / (base + 1e-30)).max()


# ----------------------------------------------------------------------------- the decode step's element-wise ops, restated
def rmsnorm_ref(x16, w16, eps):
    """RMSNorm.forward of the reference (inference/model.py:281-292) with its rounding points: fp32 x * rsqrt(mean(x^2) +
    eps), rounded to fp32, cast to fp16 (`type_as`), then an fp16 multiply by the fp16 weight."""
    xf = np.asarray(x16, dtype=np.float16).astype(np.float32)
    r = np.float32(1.0 / np.sqrt(np.mean(xf.astype(np.float64)**2) + np.float64(eps)))
    n = (xf * r).astype(np.float16)
    return (n.astype(np.float32) * np.asarray(w16, dtype=np.float16).astype(np.float32)).astype(np.float16)


def silu_mul_ref(gate16, up16):
    """F.silu(w1_out) * w3_out on fp16 tensors (inference/model.py:259-266): silu evaluated in fp32 and rounded to fp16,
    then an fp16 multiply."""
    g = np.asarray(gate16, dtype=np.float16).astype(np.float32)
    s = (g / (np.float32(1.0) + np.exp(-g))).astype(np.float16)
    return (s.astype(np.float32) * np.asarray(up16, dtype=np.float16).astype(np.float32)).astype(np.float16)


def half_add(a16, b16):
    """fp16 + fp16 -> fp16 (the residual adds of inference/model.py:311-313): the fp32 sum of two halves is exact"""
    return (np.asarray(a16, dtype=np.float16).astype(np.float32) + np.asarray(b16, dtype=np.float16).astype(np.float32)).astype(np.float16)


def lnq_like_layer(N, K, bits, seed, oracle=None):
    """A layer shaped like what LNQ + GuidedQuant emits rather than uniform noise: a skewed code histogram (the inner
    centroids carry most of the mass), per-row centroids spread like a weight row's quantiles with a few rows holding
    outlier centroids 8-40x larger, and a heavy-tailed activation vector (Student-t, plus a handful of massive-activation
    channels, 2^10 .. 2^12 times the typical element, as Llama hidden states have).  Returns (qweight, lut, x)."""
    from guidedquant_amd import pack
    rng = np.random.default_rng(seed)
    L = 1 << bits
    centre = (L - 1) / 2.0
    p = np.exp(-0.5 * ((np.arange(L) - centre) / (0.28 * L))**2)
    p /= p.sum()
    codes = rng.choice(L, size=(N, K), p=p).astype(np.uint8)
    q = pack.pack_codes(codes, bits)
    base = np.sort(rng.normal(0, 0.02, (N, L)), axis=1)
    out_rows = rng.random(N) < 0.01
    base[out_rows, 0] *= rng.uniform(8, 40, out_rows.sum())
    base[out_rows, -1] *= rng.uniform(8, 40, out_rows.sum())
    lut = base.astype(np.float16)
    x = rng.standard_t(3, K) * 0.5
    hot = rng.choice(K, 6, replace=False)
    x[hot] = np.sign(x[hot]) * 0.5 * 2.0**rng.uniform(10, 12, 6)   # 2^10 .. 2^12 x the typical element (Llama: 2^8 .. 2^11)
    return q, lut, np.clip(x, -6e4, 6e4).astype(np.float16)


def run_fused(x, q, lut, bits, norm_weight=None, eps=1e-5, residual=None, flags=0, out_elems=None, workspace=False):
    """gq_anyprec_gemv_fused through the C ABI on cuda:0 (numpy in, numpy out); out is pre-filled with NaN.
    workspace: gq_anyprec_gemv_fused_ws with the bytes gq_anyprec_gemv_fused_ws_bytes asks for (NaN-filled)"""
    import torch
    from guidedquant_amd import _lib
    d = torch.device("cuda:0")
    N, K = q.shape[1], q.shape[2] * 32
    t = lambda a, dt: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(d)  # noqa: E731
    xt, qt, lt = t(x, np.float16), torch.from_numpy(np.ascontiguousarray(q)).to(d), t(lut, np.float16)
    nw, rs = t(norm_weight, np.float16), t(residual, np.float16)
    out = torch.full((out_elems or N, ), float("nan"), dtype=torch.float16, device=d)
    if workspace:
        nb = int(_lib.lib().gq_anyprec_gemv_fused_ws_bytes(N, K, bits, flags))
        assert nb > 0, "no workspace form for this shape"
        ws = torch.full((nb // 4, ), float("nan"), dtype=torch.float32, device=d)
        rc = _lib.lib().gq_anyprec_gemv_fused_ws(xt.data_ptr(), out.data_ptr(), qt.data_ptr(), lt.data_ptr(), N, K, bits,
                                                 nw.data_ptr() if nw is not None else None, eps, rs.data_ptr() if rs is not None else None,
                                                 flags, ws.data_ptr(), nb, _lib.current_stream_ptr())
        _lib.check(rc, "gq_anyprec_gemv_fused_ws")
        torch.cuda.synchronize()
        return out.cpu().numpy()
    rc = _lib.lib().gq_anyprec_gemv_fused(xt.data_ptr(), out.data_ptr(), qt.data_ptr(), lt.data_ptr(), N, K, bits,
                                          nw.data_ptr() if nw is not None else None, eps, rs.data_ptr() if rs is not None else None,
                                          flags, _lib.current_stream_ptr())
    _lib.check(rc, "gq_anyprec_gemv_fused")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def tiny_hf_anyprec_checkpoint(path, D=128, I=256, H=4, KV=2, Lr=2, V=96, seed=2):
    """write an HF-layout Any-Precision checkpoint directory (config.json with the `anyprec` section incl. arch_config,
    pack.py:190-195; model.safetensors with `...{qweight,lut2,lut3}` keys, pack.py:112-123) of a tiny Llama; returns
    (LlamaConfig, state dict, module names, dims)"""
    import json
    import torch
    import transformers
    from safetensors.torch import save_file
    hf_cfg = transformers.LlamaConfig(hidden_size=D, intermediate_size=I, num_hidden_layers=Lr, num_attention_heads=H, num_key_value_heads=KV,
                                      vocab_size=V, max_position_embeddings=64, rms_norm_eps=1e-5, tie_word_embeddings=False)
    names = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]
    cfg = hf_cfg.to_dict()
    cfg["anyprec"] = dict(seed_precision=2, parent_precision=3, group_count=1,
                          arch_config=dict(module_names=names, model_name="model", layers_name="layers"))
    (path / "config.json").write_text(json.dumps(cfg))
    g = torch.Generator().manual_seed(seed)
    hd = D // H
    shapes = dict(zip(names, [(D, D), (KV * hd, D), (KV * hd, D), (D, D), (I, D), (I, D), (D, I)]))
    sd = {"model.embed_tokens.weight": (torch.randn(V, D, generator=g) * 0.5).half(), "model.norm.weight": torch.ones(D).half(),
          "lm_head.weight": (torch.randn(V, D, generator=g) * 0.2).half()}
    for i in range(Lr):
        sd[f"model.layers.{i}.input_layernorm.weight"] = torch.ones(D).half()
        sd[f"model.layers.{i}.post_attention_layernorm.weight"] = torch.ones(D).half()
        for name, (n, k) in shapes.items():
            sd[f"model.layers.{i}.{name}.qweight"] = torch.randint(-2**31, 2**31 - 1, (3, n, k // 32), dtype=torch.int32, generator=g)
            for b in (2, 3):
                sd[f"model.layers.{i}.{name}.lut{b}"] = (torch.randn(n, 2**b, generator=g) * 0.08).sort(dim=1).values.half().contiguous()
    save_file(sd, str(path / "model.safetensors"))
    return hf_cfg, sd, names, (D, I, H, KV, Lr, V)


# ------------------------------------------------------------------------------------------------ converter goldens (section 8 f-1)
CONVERT_DIMS = dict(D=128, I=256, H=2, KV=1, Lr=32, V=256, parent=4, seed_bits=2)  # 32 layers: the reference script's "Llama-2-7b" (:62-69)


def convert_input_state_dict(seed=7):
    """the HF-layout multi-precision Any-Precision checkpoint (keys / dtypes as pack.py:112-123 + HF Llama emit them) the converter
    goldens were generated from -- regenerated from the seed (numpy PCG64: stable across versions), not stored: a parent of 4 planes,
    lut2 / lut3 / lut4 per linear, bf16 embeddings and norms (the script casts those to fp16)"""
    import torch
    c = CONVERT_DIMS
    D, I, hd = c["D"], c["I"], c["D"] // c["H"]
    rng = np.random.default_rng(seed)
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16)  # noqa: E731
    sd = {"model.embed_tokens.weight": bf(rng.normal(0, 0.5, (c["V"], D))), "model.norm.weight": bf(1 + 0.1 * rng.normal(0, 1, D)),
          "lm_head.weight": torch.from_numpy(rng.normal(0, 0.2, (c["V"], D)).astype(np.float16))}
    shapes = {"self_attn.q_proj": (D, D), "self_attn.k_proj": (c["KV"] * hd, D), "self_attn.v_proj": (c["KV"] * hd, D), "self_attn.o_proj": (D, D),
              "mlp.gate_proj": (I, D), "mlp.up_proj": (I, D), "mlp.down_proj": (D, I)}
    for i in range(c["Lr"]):
        sd[f"model.layers.{i}.input_layernorm.weight"] = bf(1 + 0.1 * rng.normal(0, 1, D))
        sd[f"model.layers.{i}.post_attention_layernorm.weight"] = bf(1 + 0.1 * rng.normal(0, 1, D))
        for name, (n, k) in shapes.items():
            sd[f"model.layers.{i}.{name}.qweight"] = torch.from_numpy(
                rng.integers(-2**31, 2**31 - 1, (c["parent"], n, k // 32), dtype=np.int64).astype(np.int32))
            for b in range(c["seed_bits"], c["parent"] + 1):
                sd[f"model.layers.{i}.{name}.lut{b}"] = torch.from_numpy(np.sort(rng.normal(0, 0.08, (n, 1 << b)).astype(np.float16), axis=1))
    return sd


def qtip_convert_input_state_dict(seed=8, Lr=2, D=64, I=128, V=64):
    """an hfized QTIP checkpoint's key set (qtip/model/llama.py QuantizedLinear buffers per projection: trellis, SU, SV, tlut, rcp, tp_rank)"""
    import torch
    rng = np.random.default_rng(seed)
    sd = {"model.embed_tokens.weight": torch.from_numpy(rng.normal(0, 0.5, (V, D)).astype(np.float16)),
          "model.norm.weight": torch.from_numpy(np.ones(D, np.float16)), "lm_head.weight": torch.from_numpy(rng.normal(0, 0.2, (V, D)).astype(np.float16))}
    shapes = {"self_attn.q_proj": (D, D), "self_attn.k_proj": (D, D), "self_attn.v_proj": (D, D), "self_attn.o_proj": (D, D),
              "mlp.gate_proj": (I, D), "mlp.up_proj": (I, D), "mlp.down_proj": (D, I)}
    for i in range(Lr):
        sd[f"model.layers.{i}.input_layernorm.weight"] = torch.from_numpy((1 + 0.1 * rng.normal(0, 1, D)).astype(np.float16))
        sd[f"model.layers.{i}.post_attention_layernorm.weight"] = torch.from_numpy((1 + 0.1 * rng.normal(0, 1, D)).astype(np.float16))
        for name, (n, k) in shapes.items():
            p = f"model.layers.{i}.{name}."
            sd[p + "trellis"] = torch.from_numpy(rng.integers(-2**15, 2**15 - 1, ((n // 16) * (k // 16), 16 * 16 * 2 // 16), dtype=np.int64).astype(np.int16))
            sd[p + "SU"] = torch.from_numpy(rng.choice([-1.0, 1.0], k).astype(np.float16))
            sd[p + "SV"] = torch.from_numpy((rng.choice([-1.0, 1.0], n) * 0.02).astype(np.float32))
            sd[p + "tlut"] = torch.from_numpy(rng.normal(0, 0.5, (512, 2)).astype(np.float16))
            sd[p + "rcp"] = torch.tensor(0)
            sd[p + "tp_rank"] = torch.tensor(8)
    return sd


def tensor_digest(t):
    """(dtype name, shape, sha256 of the raw little-endian bytes) of a torch tensor"""
    import hashlib
    import torch
    t = t.detach().cpu().contiguous()
    raw = t.view(torch.int16).numpy().tobytes() if t.dtype == torch.bfloat16 else t.numpy().tobytes()
    return str(t.dtype).replace("torch.", ""), tuple(int(s) for s in t.shape), hashlib.sha256(raw).hexdigest()
