"""AnyPrecisionForCausalLM -- the HF-path entry of the reference (any_precision/modules/AnyPrecisionForCausalLM.py:28-210) on
MI355X: an HF causal LM whose decoder linears are `AnyPrecisionLinear` modules (multi-precision parent tensor + one LUT per
precision), with `from_quantized(path, precisions=...)`, `forward(..., precision=b)`, `generate(..., precision=b)`,
`set_precision`, `prune_precisions` -- the harness the reference's only published number (130 tok/s, README.md:95-97,
inference_example.py:34-77) runs on.

Same public surface; the construction is this package's own:
  * the skeleton comes from `AutoModelForCausalLM.from_config` on the meta device; which modules to swap is read from the
    checkpoint's `config.anyprec["arch_config"]` (`model_name`, `layers_name`, `module_names`, pack.py:190-195) -- or, when a
    checkpoint lacks it, every nn.Linear inside the decoder layers -- instead of the reference's analyzer package;
  * weights are read with plain torch / safetensors (hf_loader.read_hf_state_dict) and placed on ONE device: one process
    per GPU is this framework's model (a 70B 2-bit model is 19 GB of a 288 GB HBM), there is no accelerate `device_map`;
  * a rows == 1 call runs the HIP LUT-GEMV (plugin::anyprec_gemv), more rows the dequant + matmul branch, host tensors the
    CPU twins (AnyPrecisionLinear.py).
`native_decoder(bitwidth)` additionally returns the fused gpt-fast `Transformer` of the same checkpoint (hf_loader), whose
decode step is the 5-launches-per-layer hipGraph path bench.py measures.
"""
import gc
import json
import os
import weakref
from typing import List, Optional

import torch
import torch.nn as nn

from .AnyPrecisionLinear import AnyPrecisionLinear
from .hf_loader import read_hf_state_dict


def replace_module_by_name(layer, module_name, new_module):
    levels = module_name.split('.')
    module = layer
    for level in levels[:-1]:
        module = getattr(module, level) if not level.isdigit() else module[int(level)]
    setattr(module, levels[-1], new_module)


def _get_by_path(root, dotted):
    for name in [p for p in dotted.split('.') if p]:
        root = getattr(root, name) if not name.isdigit() else root[int(name)]
    return root


class _WeakRestore:
    """`wrapper._restore_module_tree()` through a weak reference (a state_dict pre-hook of the inner model / the `_gq_restore` of a
    released linear): no reference cycle through the wrapper"""
    __slots__ = ("ref", )

    def __init__(self, wrapper):
        self.ref = weakref.ref(wrapper)

    def __call__(self, *args, **kwargs):
        w = self.ref()
        if w is not None:
            w._restore_module_tree()


class _CapturedStep:
    """State of route 2 (`generate(capture=True)`): the StaticCache, the device words the captured step reads and writes, and the
    hipGraph of one token step of the HF module tree.  A plain object -- no closure over its own state, so no reference cycle owns the
    graph -- with an explicit `close()`: the owner (`AnyPrecisionForCausalLM._evict`) destroys the graph when it drops the entry."""
    __slots__ = ("model", "V", "top_k", "top_p", "temperature", "seed", "cache", "tok64", "tok", "pos", "nxt", "ctr", "wv", "wi", "ban", "seq", "logits", "graph")

    def __init__(self, model, config, dev, total, temperature, top_k, seed, top_p=1.0):
        from transformers import StaticCache
        self.model, self.V, self.top_k, self.temperature, self.seed = model, int(config.vocab_size), int(top_k), float(temperature), int(seed)
        self.top_p = float(top_p)
        self.cache = StaticCache(config=config, max_cache_len=total)
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)  # noqa: E731
        self.tok64 = z((1, 1), torch.long)
        self.tok, self.pos, self.nxt, self.ctr = z((1, ), torch.int32), z((1, ), torch.int32), z((1, ), torch.int32), z((1, ), torch.int32)
        self.wv, self.wi, self.ban = z(128 * 64, torch.float32), z(128 * 64, torch.int32), z(6, torch.int32)
        self.seq, self.logits = z(total + 1, torch.int32), z(self.V, torch.float16)
        self.graph = None

    def step(self):
        from . import _lib
        # (positions come from the cache itself: StaticLayer.cumulative_length is a device word the layer advances in place)
        out = self.model(input_ids=self.tok64, past_key_values=self.cache, use_cache=True)
        self.logits.copy_(out.logits.view(-1))
        _lib.check(_lib.lib().gq_sample_topk_p(self.logits.data_ptr(), self.V, self.top_k, self.top_p, self.temperature, self.seed, self.ctr.data_ptr(),
                                              self.wv.data_ptr(), self.wi.data_ptr(), self.tok.data_ptr(), self.pos.data_ptr(), self.nxt.data_ptr(),
                                              self.ban.data_ptr(), self.seq.data_ptr(), self.seq.numel(), None, None, 0, None,
                                              _lib.current_stream_ptr()), "gq_sample_topk_p")
        self.tok64.copy_(self.tok.view(1, 1))

    def _set_cache_length(self, n):
        for layer in self.cache.layers:
            if torch.is_tensor(getattr(layer, "cumulative_length", None)):
                layer.cumulative_length.fill_(n)

    def capture(self, T):
        """two eager steps on a side stream, then the capture; the cache and the positions are re-set behind them"""
        from ._graphs import capture
        keep = (self.tok64.clone(), self.tok.clone(), self.ctr.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture(g):
            self.step()
        torch.cuda.synchronize()
        self.graph = g
        # the warm-up steps wrote cache rows T-1 .. T+1 and moved the positions: rows past the position are overwritten before they are
        # read (causal), the rest is restored
        self.tok64.copy_(keep[0])
        self.tok.copy_(keep[1])
        self.ctr.copy_(keep[2])
        self.pos.fill_(T - 1)
        if T == 1:  # (no prompt pass ran: the layers allocated their tensors inside the warm-up)
            self.cache.reset()
        self._set_cache_length(T - 1)

    def close(self):
        from ._graphs import release
        g, self.graph = self.graph, None
        release(g)
        self.cache = None


class AnyPrecisionForCausalLM(nn.Module):

    def __init__(self, model_path, config, precisions: Optional[List[int]] = None, torch_dtype=torch.float16, fuse_layers=False,
                 trust_remote_code=True, local_dir=None, device=None, random_init_seed: Optional[int] = None):
        super().__init__()
        from transformers import AutoModelForCausalLM
        if torch_dtype != torch.float16:
            raise RuntimeError('Only float16 is supported for now.')
        self.config = config
        self.model_path = model_path
        ap = config.anyprec if hasattr(config, "anyprec") else None
        if not ap:
            raise ValueError("config has no 'anyprec' section (not an Any-Precision checkpoint)")
        self.supported_bits = list(range(ap['seed_precision'], ap['parent_precision'] + 1))
        if precisions is None:
            self.precisions = self.supported_bits
        else:
            assert len(precisions) == len(set(precisions)), "Precisions must be unique"
            assert all(bit in self.supported_bits for bit in precisions), \
                f"Supported bits {precisions} must be a subset of model supported bits {self.supported_bits}"
            self.precisions = precisions
        self.precision = max(self.precisions)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)

        with torch.device("meta"):
            self.model = AutoModelForCausalLM.from_config(config=config, torch_dtype=torch_dtype, trust_remote_code=trust_remote_code)
        self.ap_linears = []
        self._load_quantized_modules()
        self._native_cache = {}
        self._released = None
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module._restore_module_tree())
        # (also when the HF model inside is asked directly: self.model.state_dict() / save_pretrained.  Through a weak reference -- the
        # wrapper owns captured graphs and must stay collectable by reference counting alone, see _graphs.py)
        self.model.register_state_dict_pre_hook(_WeakRestore(self))
        # new weights: the fused decode model (built from the old tensors) and every captured graph are stale
        self.register_load_state_dict_pre_hook(lambda module, *a, **k: module._drop_native())
        if random_init_seed is not None:
            self._random_weights(random_init_seed)
        else:
            if not os.path.exists(model_path):
                raise FileNotFoundError(f"{model_path}: not a local checkpoint directory (there is no hub download here; "
                                        "the reference calls snapshot_download)")
            self._load_weights(model_path)
        self.tie_weights()
        if self.device.type == "cuda":  # before transformers' generate captures anything under inference mode (generate.py)
            from .generate import prime_graph_rng_state
            prime_graph_rng_state(self.device)
        if fuse_layers:
            self.fuse_layers()
        self.prune_precisions()

    # -- construction ------------------------------------------------------------------------------------------------
    def get_model_layers(self):
        arch = (self.config.anyprec.get('arch_config') or {})
        module = _get_by_path(self.model, arch.get('model_name', 'model'))
        return getattr(module, arch.get('layers_name', 'layers'))

    @property
    def layer_type(self):
        for layer in self.get_model_layers():
            return layer.__class__.__name__
        return None

    def _load_quantized_modules(self):
        arch = (self.config.anyprec.get('arch_config') or {})
        names = arch.get('module_names')
        for layer in self.get_model_layers():
            targets = {}
            if names:
                for n in names:
                    targets[n] = _get_by_path(layer, n)
            else:
                targets = {n: m for n, m in layer.named_modules() if isinstance(m, nn.Linear)}
            for name, module in targets.items():
                lin = AnyPrecisionLinear(module.in_features, module.out_features, self.supported_bits, bias=module.bias is not None,
                                         precisions=self.precisions, device="meta", dtype=torch.float16)
                lin.output = torch.zeros((1, 1, module.out_features), dtype=torch.float16, device=self.device)
                self.ap_linears.append(lin)
                replace_module_by_name(layer, name, lin)

    def _materialise_meta_buffers(self):
        # non-persistent buffers (the rotary embedding's inv_freq) have no data in a checkpoint and are still on the meta
        # device: a fresh instance of the owning module, built from its config on the CPU, supplies them
        for mod in list(self.model.modules()):
            meta = [b for b, buf in mod._buffers.items() if buf is not None and buf.is_meta]
            if not meta:
                continue
            if not hasattr(mod, "config"):
                raise RuntimeError(f"buffers {meta} of {type(mod).__name__} have no data in the checkpoint")
            with torch.device("cpu"):
                fresh = type(mod)(mod.config)
            for b in meta:
                mod._buffers[b] = fresh._buffers[b]

    def _random_weights(self, seed, lut_std=0.02):
        """synthetic weights of the checkpoint format, generated on the device (benchmarks / tests without a checkpoint on disk: the
        `--random_init` of inference/generate.py:222-245 for the HF-path harness)"""
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        sd = {}
        for name, t in list(self.model.state_dict().items()):
            shape = tuple(t.shape)
            if name.endswith("qweight"):
                v = torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=self.device, generator=g)
            elif ".lut" in name:
                v = (torch.randn(shape, device=self.device, generator=g) * lut_std).sort(dim=1).values.half().contiguous()
            elif name.endswith("norm.weight") or "layernorm" in name:
                v = torch.ones(shape, dtype=torch.float16, device=self.device)
            else:
                v = (torch.randn(shape, device=self.device, generator=g) * 0.02).half()
            sd[name] = v
        self.model.load_state_dict(sd, strict=False, assign=True)
        self._materialise_meta_buffers()
        self.model.to(self.device)
        for lin in self.ap_linears:
            lin.output = lin.output.to(self.device)

    def _load_weights(self, path):
        sd = read_hf_state_dict(path)
        if "lm_head.weight" not in sd and getattr(self.config, "tie_word_embeddings", False):
            emb = [k for k in sd if k.endswith("embed_tokens.weight")]
            if emb:
                sd["lm_head.weight"] = sd[emb[0]]
        cast = {k: (v.to(torch.float16) if v.is_floating_point() else v) for k, v in sd.items()}
        missing, unexpected = self.model.load_state_dict(cast, strict=False, assign=True)
        missing = [k for k in missing if "rotary_emb.inv_freq" not in k]
        if missing:
            raise RuntimeError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:4]}")
        self._materialise_meta_buffers()
        self.model.to(self.device)
        for lin in self.ap_linears:
            lin.output = lin.output.to(self.device)
        del sd, cast
        gc.collect()

    # -- the reference's surface ---------------------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        prev_precision = self.precision
        if 'precision' in kwargs:
            self.set_precision(kwargs.pop('precision'))
        results = self.model.forward(*args, **kwargs)
        self.set_precision(prev_precision)
        return results

    # keyword arguments of `generate` the fused routes understand (everything else -> transformers' own generate)
    _ROUTE_KW = {"input_ids", "inputs", "max_new_tokens", "min_new_tokens", "max_length", "do_sample", "temperature", "top_k", "top_p",
                 "pad_token_id", "eos_token_id", "attention_mask", "cache_implementation", "use_cache", "streamer", "num_beams",
                 "num_return_sequences", "repetition_penalty", "return_dict_in_generate"}

    def _route_request(self, args, kwargs):
        """(params, None) when the request is one the fused decode routes serve -- ONE sequence, greedy or top-k (<= 64) sampling with
        or without a nucleus (top_p), no logits processors beyond min_new_tokens on EOS -- else (None, reason).  Missing sampling arguments come from
        the model's generation_config exactly as in transformers (top_k defaults to 50 there)."""
        if len(args) > 1 or any(k not in self._ROUTE_KW for k in kwargs):
            return None, "arguments outside the fused routes: %s" % sorted(k for k in kwargs if k not in self._ROUTE_KW)
        ids = args[0] if args else kwargs.get("input_ids", kwargs.get("inputs"))
        if not torch.is_tensor(ids) or ids.dim() != 2 or ids.shape[0] != 1 or ids.shape[1] < 1 or ids.is_floating_point():
            return None, "input_ids must be one sequence of token ids, [1, T]"
        gc_ = getattr(self.model, "generation_config", None)

        # transformers 5 keeps unset fields of a GenerationConfig as None and fills in its global defaults at generate time (top_k = 50,
        # ...); in transformers 4 (the reference pins 4.52.3) the fields carry the defaults themselves and a None was put there by the
        # user (top_k=None: top-k filtering disabled)
        none_is_unset = gc_ is not None and hasattr(type(gc_), "_get_default_generation_params")

        def g(name, dflt=None):
            # a keyword that is not None wins, else the generation config's value, else the default
            v = kwargs.get(name)
            if v is not None:
                return v
            if gc_ is not None and hasattr(gc_, name):
                v = getattr(gc_, name)
                if v is not None or not none_is_unset:
                    return v
            return dflt
        if (g("num_beams", 1) or 1) != 1 or (g("num_return_sequences", 1) or 1) != 1 or g("return_dict_in_generate", False) or g("use_cache", True) is False:
            return None, "beam search / several return sequences / dict output / use_cache=False"
        rp = g("repetition_penalty", 1.0)
        if rp is not None and float(rp) != 1.0:
            return None, "repetition_penalty"
        for name in ("no_repeat_ngram_size", "encoder_no_repeat_ngram_size", "bad_words_ids", "force_words_ids", "suppress_tokens", "begin_suppress_tokens",
                     "forced_bos_token_id", "forced_eos_token_id", "sequence_bias", "typical_p", "epsilon_cutoff", "eta_cutoff", "min_p", "penalty_alpha",
                     "min_length"):
            v = getattr(gc_, name, None) if gc_ is not None else None
            if v not in (None, 0, 0.0, 1.0, [], False):
                return None, "generation_config.%s" % name
        T = int(ids.shape[1])
        max_new = g("max_new_tokens", None)  # (keyword first, then generation_config.max_new_tokens, then max_length - T: transformers' order)
        if max_new is None:
            ml = g("max_length", None)
            max_new = (int(ml) - T) if ml is not None else None
        if max_new is None or int(max_new) < 1:
            return None, "max_new_tokens"
        am = kwargs.get("attention_mask")
        if am is not None and not (tuple(am.shape) == tuple(ids.shape) and bool((am != 0).all())):
            return None, "attention_mask with masked positions"
        do_sample = bool(g("do_sample", False))
        temperature, top_k, top_p = 0.0, 1, 1.0
        if do_sample:
            t_ = g("temperature", 1.0)
            temperature = 1.0 if t_ is None else float(t_)
            top_p, top_k = g("top_p", 1.0), g("top_k", 50)
            top_p = 1.0 if top_p is None else float(top_p)
            if not 0.0 < top_p <= 1.0:
                return None, "top_p outside (0, 1]"
            # (round 6: top_p < 1 is served -- the fused sampler filters the nucleus of its top-k survivors the way transformers chains
            # TopKLogitsWarper and TopPLogitsWarper; without a top_k <= 64 the nucleus of the FULL distribution would be needed: declined)
            # (top_k None or 0 = no top-k filtering in transformers: the full distribution, which the 64-candidate sampler does not draw from)
            if not top_k or int(top_k) < 1 or int(top_k) > 64 or temperature <= 0.0:
                return None, "top_k outside 1..64 (the fused sampler's candidates)"
        eos = g("eos_token_id", None)
        eos = [] if eos is None else ([int(eos)] if isinstance(eos, int) else [int(e) for e in eos])
        if len(eos) > 4:
            return None, "more than 4 EOS ids"
        return dict(ids=ids, T=T, max_new=int(max_new), min_new=min(int(g("min_new_tokens", 0) or 0), int(max_new)), temperature=temperature,
                    top_k=int(top_k), top_p=float(top_p), eos=eos, streamer=kwargs.get("streamer"), do_sample=do_sample), None

    def generate(self, *args, **kwargs):
        """`generate` of the reference's HF surface (inference_example.py:34-77).  Three routes behind the one call:
          1. the fused decode model of the same checkpoint (`native_decoder`: prompt through the HIP prompt pass, every new token one
             replay of the captured 5-launches-per-layer graph, sampling / EOS suppression / embedding of the next token on the device)
             -- taken AUTOMATICALLY for a request it serves (`_route_request`: one sequence, greedy or top-k <= 64 at top_p = 1 -- the
             reference's own call); `native=False` opts out, `native=True` insists (ValueError when the request is not served);
          2. `capture=True`: the HF module tree with its decode step captured as ONE hipGraph over a transformers StaticCache, the
             fused sampler at its end -- opt-in: measured SLOWER than route 3 on the 8B model (172 vs 249 tokens/s: the module tree's
             ~1,500 small launches per token cost more as graph nodes than as stream launches; the fused model is the fast form);
          3. transformers' own generate on the module tree (anything else: beams, top_p, processors, batches; or `native=False`).
        Returns the [1, prompt + new] token tensor like HF does; stops at EOS (checked every 32 tokens, the tail is cut); honours
        min_new_tokens, streamer, precision=."""
        prev_precision = self.precision
        if 'precision' in kwargs:
            self.set_precision(kwargs.pop('precision'))
        native = kwargs.pop('native', None)
        capture = kwargs.pop('capture', None)
        try:
            req, why = self._route_request(args, kwargs) if (native is not False or capture is True) else (None, "opted out")
            if req is not None and self.device.type != "cuda":
                req, why = None, "the fused routes need the GPU"
            if native is True and req is None:
                raise ValueError("native=True: " + why)
            if req is not None and native is not False:
                # (the q/k/v/gate/up planes of the module tree are released only on the explicit native=True: the automatic route keeps
                # the module tree whole, so that model.state_dict() / save_pretrained / direct buffer access after a plain generate()
                # see every tensor)
                dec = self._native_decoder_or_none(self.precision, release_planes=native is True)
                if dec is not None and req["T"] + req["max_new"] > dec.config.block_size:
                    if native is True:
                        raise ValueError(f"native=True: prompt + max_new_tokens = {req['T'] + req['max_new']} exceeds the fused model's "
                                         f"context ({dec.config.block_size})")
                    dec = None  # (automatic route: transformers' generate decides what a request beyond the context means)
                if dec is not None:
                    return self._generate_native(dec, req)
                if native is True:
                    raise ValueError("native=True: this checkpoint has no fused decode form at %d bits" % self.precision)
            if capture is True:
                if req is None:
                    raise ValueError("capture=True: " + why)
                return self._generate_captured(req)
            with torch.inference_mode():
                return self.model.generate(*args, **kwargs)
        finally:
            self.set_precision(prev_precision)

    # -- route 1: the fused decode model ---------------------------------------------------------------------------------
    _SAMPLER_SEED = 0x2545F491  # (a constant: what varies between sampled calls is the counter word, drawn from torch's generator)

    def _fresh_rng_word(self):
        """the start of the fused sampler's counter stream for ONE sampled call, drawn from torch's generator of this device: as with
        transformers' generate, `torch.manual_seed(s)` before a call makes it reproducible and two unseeded calls differ -- whatever
        graph is cached (the captured launches read the counter from device memory, nothing of the seed is baked in)"""
        return torch.randint(-2**31, 2**31 - 1, (1, ), dtype=torch.int32, device=self.device)

    def _native_decoder_or_none(self, bitwidth, release_planes=False):
        try:
            dec = self.native_decoder(bitwidth, release_planes=release_planes)
        except (ValueError, NotImplementedError):  # "this checkpoint / precision has no fused form"; anything else is a real error
            return None
        dec.setup_caches(1, 8) if not dec.cache_initialized else None
        return dec if dec.native_ready() else None

    def _evict(self, kind):
        """drop the cached entries of one kind ("graph": DecodeGraphs of route 1, "cap": captured steps of route 2), destroying their
        hipGraphs synchronously -- never left to a finaliser that may run inside a later capture (_graphs.py)"""
        keep = {}
        for k, v in self._native_cache.items():
            if k[0] == kind:
                v.close()
            else:
                keep[k] = v
        self._native_cache = keep

    def _emit(self, req, seq_host, lo, hi):
        """tokens [lo, hi) of the host copy of the sequence: to the streamer (up to and including the first EOS that counts -- index
        >= T + min_new_tokens); returns the index behind that EOS, or None"""
        cut = None
        for i in range(max(lo, req["T"] + req["min_new"]), hi if req["eos"] else 0):
            if int(seq_host[i]) in req["eos"]:
                cut = i + 1
                break
        if req["streamer"] is not None and hi > lo:
            req["streamer"].put(seq_host[lo:(cut or hi)])
        return cut

    def _generate_native(self, dec, req, chunk=32):
        from . import generate as gen
        ids, T, max_new = req["ids"].to(self.device), req["T"], req["max_new"]
        total = T + max_new
        if total > dec.config.block_size:
            raise ValueError(f"prompt + max_new_tokens = {total} exceeds the model's context ({dec.config.block_size})")
        # caches (and with them the captured graphs, keyed by the cache length) grow in powers of two from 256 positions: a serving loop
        # with varying lengths re-captures at most log2 times, not per request (the attention launch reads rows up to the position
        # only; a longer cache costs memory, not time).  Beyond 16384 positions the request's own length (the fallback route's
        # causal mask is quadratic in it).
        cap = total if total > 16384 else max(256, 1 << (total - 1).bit_length())
        dec.setup_caches(1, min(cap, dec.config.block_size))
        key = (self.precision, dec.max_seq_length, req["temperature"], req["top_k"], req["top_p"])
        graph = self._native_cache.get(("graph",) + key)
        if graph is None:
            self._evict("graph")
            # (eight token steps per graph replay -- the host looks at the sequence once per `chunk` = 32 tokens anyway; what is left of a
            # chunk runs through the single-step graph over the same state: exactly the steps asked for)
            graph = gen.DecodeGraph(dec, self.device, native_sampling=True, fold_embed=True, seq_capacity=dec.max_seq_length + 1,
                                    seed=self._SAMPLER_SEED, temperature=req["temperature"], top_k=req["top_k"], top_p=req["top_p"], steps_per_replay=8)
            self._native_cache[("graph",) + key] = graph
        if req.get("do_sample"):
            graph.rng_counter.copy_(self._fresh_rng_word())
        ids32 = ids.view(-1).to(torch.int32)
        with torch.inference_mode():
            graph.seq[:T].copy_(ids32)
            # EOS cannot be drawn before min_new_tokens are out: the token of the step at position p is new token number p + 2 - T
            ban = [len(req["eos"]), T - 1 + req["min_new"]] + req["eos"] + [0] * (4 - len(req["eos"]))
            graph.ban.copy_(torch.tensor(ban, dtype=torch.int32), non_blocking=False)
            if T > 1:  # the prompt but for its last token fills the caches (HIP prompt pass; one eager step for a single token)
                pre, pos = ids32[:T - 1], torch.arange(0, T - 1, device=self.device, dtype=torch.int32)
                if dec.prefill_ready(pre):
                    dec.prefill_native(pre, pos, start=0)
                elif T - 1 == 1:
                    dec.decode_native(pre.view(1), pos.view(1))
                else:
                    dec(pre.view(1, -1), pos)
            graph.set_token(ids32[T - 1:T], T - 1)
            if req["streamer"] is not None:
                req["streamer"].put(ids.cpu())
            done, cut = 0, None
            host_checks = bool(req["eos"]) and req["min_new"] < max_new or req["streamer"] is not None
            while done < max_new and cut is None:
                n = min(chunk, max_new - done) if host_checks else max_new - done
                for _ in range(n // graph.steps_per_replay):
                    graph.step()
                for _ in range(n % graph.steps_per_replay):
                    graph.step_one()
                if host_checks:
                    seq_host = graph.seq[:T + done + n].cpu().long()
                    cut = self._emit(req, seq_host, T + done, T + done + n)
                done += n
            end = cut if cut is not None else T + done
            out = graph.seq[:end].to(ids.dtype).view(1, -1).clone()
        if req["streamer"] is not None:
            req["streamer"].end()
        return out

    # -- route 2: the module tree, decode step captured -------------------------------------------------------------------
    def _generate_captured(self, req, chunk=32):
        """bs = 1 decode on the HF module tree with ONE hipGraph per token: the step `logits = model(input_ids [1,1], StaticCache,
        cache_position)` + the fused sampler (gq_sample_topk_ex: the draw, EOS suppression, the sequence store, token feedback) are
        captured once per (precision, cache length, sampling) and replayed; the prompt runs eagerly through the same cache."""
        ids, T, max_new = req["ids"].to(self.device), req["T"], req["max_new"]
        total = T + max_new
        key = ("cap", self.precision, total, req["temperature"], req["top_k"], req["top_p"])
        st = self._native_cache.get(key)
        # (no_grad, not inference_mode: the first graph capture of a process creates the generator's graph-safe state tensors, and
        # inference tensors could not be updated by the captures that follow outside inference mode)
        with torch.no_grad():
            if st is None:
                self._evict("cap")  # (the old entry's graph is destroyed HERE, before the new capture begins)
                st = _CapturedStep(self.model, self.config, self.device, total, req["temperature"], req["top_k"], self._SAMPLER_SEED, req["top_p"])
                self._native_cache[key] = st
            st.cache.reset()
            st.seq[:T].copy_(ids.view(-1).to(torch.int32))
            ban = [len(req["eos"]), T - 1 + req["min_new"]] + req["eos"] + [0] * (4 - len(req["eos"]))
            st.ban.copy_(torch.tensor(ban, dtype=torch.int32))
            if req.get("do_sample"):
                st.ctr.copy_(self._fresh_rng_word())
            if T > 1:
                self.model(input_ids=ids[:, :T - 1], past_key_values=st.cache, use_cache=True)
            st.tok64.copy_(ids[:, T - 1:T])
            st.tok.copy_(ids.view(-1)[T - 1:T].to(torch.int32))
            st.pos.fill_(T - 1)
            if st.graph is None:
                st.capture(T)
            if req["streamer"] is not None:
                req["streamer"].put(ids.cpu())
            done, cut = 0, None
            host_checks = bool(req["eos"]) and req["min_new"] < max_new or req["streamer"] is not None
            while done < max_new and cut is None:
                n = min(chunk, max_new - done) if host_checks else max_new - done
                for _ in range(n):
                    st.graph.replay()
                if host_checks:
                    seq_host = st.seq[:T + done + n].cpu().long()
                    cut = self._emit(req, seq_host, T + done, T + done + n)
                done += n
            end = cut if cut is not None else T + done
            out = st.seq[:end].to(ids.dtype).view(1, -1).clone()
        if req["streamer"] is not None:
            req["streamer"].end()
        return out

    @staticmethod
    def _load_config(model_path, trust_remote_code=True):
        from transformers import AutoConfig
        return AutoConfig.from_pretrained(model_path, trust_remote_code=trust_remote_code)

    @classmethod
    def from_config_random(cls, config, precisions=None, device=None, seed=0, torch_dtype=torch.float16):
        """the module tree of `config` (a transformers config with the `anyprec` section) with synthetic weights, no checkpoint"""
        return cls(model_path=None, config=config, precisions=precisions, torch_dtype=torch_dtype, device=device, random_init_seed=seed)

    @classmethod
    def from_quantized(cls, quant_model_path, trust_remote_code=True, fuse_layers=False, precisions=None, local_dir=None,
                       torch_dtype=torch.float16, device=None):
        config = cls._load_config(quant_model_path, trust_remote_code)
        return cls(model_path=quant_model_path, precisions=precisions, config=config, fuse_layers=fuse_layers,
                   trust_remote_code=trust_remote_code, local_dir=local_dir, torch_dtype=torch_dtype, device=device)

    def prune_precisions(self):
        for ap_linear in self.ap_linears:
            ap_linear.prune_precisions()
        gc.collect()

    def set_precision(self, precision):
        for ap_linear in self.ap_linears:
            ap_linear.set_precision(precision)
        self.precision = precision

    def tie_weights(self):
        if hasattr(self.model, "tie_weights"):
            self.model.tie_weights()

    def fuse_layers(self):
        # the reference leaves this a TODO (AnyPrecisionForCausalLM.py:192-196); the fused decode model is `native_decoder`
        raise NotImplementedError("layer fusion inside the HF module tree is not implemented (as in the reference); use "
                                  "native_decoder(bitwidth) for the fused QKV / Up-Gate decode path")

    def native_decoder(self, bitwidth: Optional[int] = None, release_planes: bool = False):
        """the same checkpoint as the fused gpt-fast `Transformer` (fused QKV / Up-Gate Any-Precision linears at one precision)
        whose bs=1 decode step runs as the captured 5-launches-per-layer HIP graph.  Built from the module tree's tensors BY
        REFERENCE: embedding, lm_head, norms, o_proj and down_proj are the same storage; q/k/v and gate/up are concatenated into the
        fused tensors layer by layer.
        release_planes=True (what `generate(native=True)` asks for), when the checkpoint carries exactly this precision -- the
        GuidedQuant case: the module tree's own q/k/v/gate/up planes are released as the fused tensors are built (one copy of every
        weight; restored from the fused tensors the first time the module tree is used again through this wrapper, a linear's forward,
        `state_dict()` of the wrapper or of `self.model`, or `.to()` -- `_restore_module_tree`).  The default keeps the module tree whole
        (the fused q/k/v/gate/up tensors are then a second copy: 1.1 GB for an 8B 2-bit model)."""
        bitwidth = bitwidth or min(self.precisions)
        dec = self._native_cache.get(("decoder", bitwidth))
        if dec is not None and release_planes and not self._released and all(lin.qweight.shape[0] == bitwidth for lin in self.ap_linears):
            self._drop_native()  # (built without the release: built again, releasing as it goes)
            dec = None
        if dec is None:
            self._restore_module_tree()  # (a decoder of another precision may hold the q/k/v/gate/up planes)
            dec = self._build_native(bitwidth, release_planes)
            self._native_cache[("decoder", bitwidth)] = dec
        return dec

    def _layer_linears(self, layer):
        at, mlp = layer.self_attn, layer.mlp
        return dict(q=at.q_proj, k=at.k_proj, v=at.v_proj, o=at.o_proj, gate=mlp.gate_proj, up=mlp.up_proj, down=mlp.down_proj)

    def _build_native(self, bitwidth, release_planes=False):
        from .APLinear import APLinear
        from .hf_loader import model_args_from_hf_config
        from .model import Transformer
        if bitwidth not in self.precisions:
            raise ValueError(f"bitwidth {bitwidth} not among the loaded precisions {self.precisions}")
        cfg = self.config.to_dict() if hasattr(self.config, "to_dict") else dict(self.config)
        args = model_args_from_hf_config(cfg)
        dev = self.device
        with torch.device("meta"):
            dec = Transformer(torch.float16, args, linear_class=APLinear, linear_kwargs=dict(bitwidth=bitwidth, device="meta"), fuse_linears=True)
        inner = _get_by_path(self.model, (self.config.anyprec.get('arch_config') or {}).get('model_name', 'model'))
        lm_head = self.model.get_output_embeddings() if hasattr(self.model, "get_output_embeddings") else self.model.lm_head
        dec.tok_embeddings.weight = inner.embed_tokens.weight
        dec.output.weight = lm_head.weight
        dec.norm.weight = inner.norm.weight
        # (released only when the planes ARE this precision: nothing is lost by releasing them)
        single = bool(release_planes) and all(lin.qweight.shape[0] == bitwidth for lin in self.ap_linears)

        def put(mod, qweight, lut):
            mod._buffers["qweight"] = qweight
            mod._buffers["lut"] = lut
            mod.output = torch.zeros((1, 1, mod.out_features), dtype=torch.float16, device=dev)

        # what has been released is on record BEFORE the first plane goes (with the decoder that now holds it): a failure at layer k
        # (out of memory in a concatenation, ...) gives layers < k their planes back before the error travels on
        released = []
        if single:
            self._evict("cap")  # (captured steps of the module tree point at the planes about to go)
            self._released = (bitwidth, released, dec)
        restore = _WeakRestore(self)
        try:
            for hf_layer, blk in zip(self.get_model_layers(), dec.layers):
                L = self._layer_linears(hf_layer)
                if any(m.bias is not None for m in L.values()):
                    raise NotImplementedError("fused decode model: biased linears")
                blk.input_layernorm.weight = hf_layer.input_layernorm.weight
                blk.post_attention_layernorm.weight = hf_layer.post_attention_layernorm.weight
                lut = lambda m: m._buffers[f"lut{bitwidth}"].to(torch.float16)  # noqa: E731
                put(blk.attention.wo, L["o"].qweight[:bitwidth], lut(L["o"]))           # (a prefix of the planes: contiguous view)
                put(blk.feed_forward.w2, L["down"].qweight[:bitwidth], lut(L["down"]))
                put(blk.attention.wqkv, torch.cat([L[n].qweight[:bitwidth] for n in "qkv"], dim=1).contiguous(),
                    torch.cat([lut(L[n]) for n in "qkv"], dim=0).contiguous())
                put(blk.feed_forward.w1w3, torch.cat([L["gate"].qweight[:bitwidth], L["up"].qweight[:bitwidth]], dim=1).contiguous(),
                    torch.cat([lut(L["gate"]), lut(L["up"])], dim=0).contiguous())
                if single:
                    released.append(hf_layer)
                    for n in ("q", "k", "v", "gate", "up"):
                        L[n]._buffers["qweight"] = torch.empty((0, ), dtype=torch.int32, device=dev)
                        L[n]._gq_restore = restore
            dec = dec.eval()
            dec._reset_native()
        except BaseException:
            self._restore_module_tree()
            raise
        if not released:
            self._released = None
        return dec

    def _drop_native(self):
        self._restore_module_tree()
        self._evict("graph")
        self._evict("cap")
        self._native_cache = {}

    def _apply(self, fn, *args, **kwargs):
        # .to() / .half() / .cuda(): the module tree gets new tensors -- it takes its planes back first, and the fused decode model (built
        # over the old tensors by reference) and every captured graph are stale
        if getattr(self, "_native_cache", None) is not None:
            self._drop_native()
        return super()._apply(fn, *args, **kwargs)

    def _restore_module_tree(self):
        """give q/k/v and gate/up of the module tree their plane tensors back (slices of the fused decode model's tensors; the gate/up
        rows un-paired), and drop the fused model and its graphs -- one copy of every weight at any time.  Works from its own record
        (`_released` = precision, the layers released so far, the decoder holding their planes), not from the cache."""
        rel = getattr(self, "_released", None)
        if not rel:
            return
        from .model import _pair_perm
        bitwidth, layers, dec = rel
        self._released = None
        done = {id(l) for l in layers}
        for hf_layer, blk in zip(self.get_model_layers(), dec.layers):
            if id(hf_layer) not in done:
                continue
            L = self._layer_linears(hf_layer)
            qkv, gu = blk.attention.wqkv.qweight, blk.feed_forward.w1w3.qweight
            if getattr(blk.feed_forward.w1w3, "gq_row_pairs", False):
                gu = gu[:, torch.argsort(_pair_perm(gu.shape[1] // 2, gu.device)), :]
            r0 = 0
            for n in ("q", "k", "v"):
                L[n]._buffers["qweight"] = qkv[:, r0:r0 + L[n].out_features, :].contiguous()
                r0 += L[n].out_features
            L["gate"]._buffers["qweight"] = gu[:, :L["gate"].out_features, :].contiguous()
            L["up"]._buffers["qweight"] = gu[:, L["gate"].out_features:, :].contiguous()
            for n in ("q", "k", "v", "gate", "up"):
                L[n]._gq_restore = None
            blk.attention.wqkv._buffers["qweight"] = blk.feed_forward.w1w3._buffers["qweight"] = None
        self._evict("graph")
        self._evict("cap")
        self._native_cache = {k: v for k, v in self._native_cache.items() if k[0] != "decoder"}
        gc.collect()
