"""Decode harness mirroring inference/generate.py of the reference: sampling (`logits_to_probs`, `sample`,
`multinomial_sample_one_no_sync`, :53-73), `decode_one_token` (:82-86), `decode_n_tokens` with a captured graph
(:92-139), `generate` (:146-193), `load_model` backend switch with `random_init` (:195-245), the tokens/s and
"Bandwidth achieved" metrics (:247-266, :374-389).

Differences forced by the environment (no network: no tokenizer, no checkpoints): prompts are token-id tensors
(BOS-only by default, as `encode_bos`), weights are synthetic (`random_init=True`, the reference's own switch), and
`torch.compile(mode="max-autotune")` is replaced by the fused HIP decode step + one hipGraph per token step.
"""
import itertools
import time
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import pack
from ._graphs import capture, release
from .APLinear import APLinear
from .LUTGEMMLinear import LUTGEMMLinear
from .model import Transformer


def multinomial_sample_one_no_sync(probs_sort):  # does multinomial sampling without a cuda synchronization
    q = torch.empty_like(probs_sort).exponential_(1)
    return torch.argmax(probs_sort / q, dim=-1, keepdim=True).to(dtype=torch.int)


def logits_to_probs(logits, temperature: float = 1.0, top_k: Optional[int] = None):
    logits = logits / max(temperature, 1e-5)
    if top_k is not None:
        v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
        pivot = v.select(-1, -1).unsqueeze(-1)
        logits = torch.where(logits < pivot, -float("Inf"), logits)
    probs = torch.nn.functional.softmax(logits, dim=-1)
    return probs


def sample(logits, temperature: float = 1.0, top_k: Optional[int] = None):
    logits = logits.float()
    probs = logits_to_probs(logits[:, -1], temperature, top_k)
    idx_next = multinomial_sample_one_no_sync(probs)
    return idx_next, probs


def prefill(model: Transformer, x: torch.Tensor, input_pos: torch.Tensor, **sampling_kwargs) -> torch.Tensor:
    """input_pos = arange(0, T) (generate()).  Any-Precision models on the GPU take the HIP prompt pass (Transformer.prefill_native:
    logits of the last token only -- all that is sampled from); GQ_PREFILL_NATIVE=0 keeps the module forward."""
    import os
    if os.environ.get("GQ_PREFILL_NATIVE", "1") != "0" and model.cache_initialized and model.prefill_ready(x):
        logits = model.prefill_native(x, input_pos, start=0, last_only=True)
    else:
        logits = model(x, input_pos)
    return sample(logits, **sampling_kwargs)[0]


def decode_one_token(model: Transformer, x: torch.Tensor, input_pos: torch.Tensor, **sampling_kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
    """x: [1,1] token ids, input_pos: [1] (int32).  Uses the fused HIP step when the model supports it."""
    assert input_pos.shape[-1] == 1
    if model.native_ready():
        logits = model.decode_native(x.view(1), input_pos.view(1))
    else:
        logits = model(x, input_pos)
    return sample(logits, **sampling_kwargs)


_prime_graphs = {}  # device index -> (graph, tensor): CUDA default generators are per device


def prime_graph_rng_state(device=None):
    """The FIRST graph capture on a device creates that device's default generator's graph-safe state tensors; created under
    torch.inference_mode() (transformers' generate captures there) they are inference tensors, and every later capture outside
    inference mode fails ("Inplace update to inference tensor outside InferenceMode").  One tiny capture outside inference mode, before
    anybody else captures, makes them ordinary tensors, which both kinds of capture may update."""
    if not torch.cuda.is_available() or torch.is_inference_mode_enabled() or torch.cuda.is_current_stream_capturing():
        return
    idx = torch.device(device).index if device is not None and torch.device(device).index is not None else torch.cuda.current_device()
    if idx in _prime_graphs:
        return
    with torch.cuda.device(idx):
        t = torch.zeros(1, device="cuda")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with capture(g, stream=s):
                t.add_(1.0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # the graph is KEPT: the state tensors live as long as a graph is registered with the generator -- with none left, the next
        # capture (possibly under inference mode) would create them anew
        _prime_graphs[idx] = (g, t)


class DecodeGraph:
    """One captured hipGraph of `decode_one_token` with static token / position / output tensors
    (the manual-graph path of generate.py:95-113)."""

    def __init__(self, model: Transformer, device, native_sampling=False, seed=1234, fold_embed=False, seq_capacity=0, steps_per_replay=1,
                 **sampling_kwargs):
        """native_sampling: draw the token with the fused HIP sampler (gq_sample_topk_ex: same distribution as `sample`,
        own counter-based RNG, top_k <= 64) and feed token / position back inside the graph; `next_prob` is then
        not produced.  Default False = the reference's torch sampling ops, captured in the graph.
        fold_embed (native sampling only): the sampler's last block also writes the NEXT step's hidden state (the embedding row of
        the token it drew, model.py:121-130) and its hand-over statistics, so the captured step starts at layer 0's first GEMV -- one
        launch less per token.  The token of the first step must then be set through `set_token` (which runs the lookup once);
        writing `self.tok` directly is only valid without the fold.
        seq_capacity > 0: the sampler also stores every token at `self.seq[pos + 1]` (int32 [seq_capacity]); `self.ban` (int32
        {n, until_pos, id0..id3}) lists tokens that cannot be drawn while pos < until_pos (HF's min_new_tokens on EOS).
        steps_per_replay > 1 (native sampling only): the graph holds that many consecutive token steps -- token, position and RNG
        counter are fed back on the device, so step i + 1 needs nothing from the host; one replay then decodes `steps_per_replay`
        tokens (`next_tok` = the last of them, all of them in `self.seq` when seq_capacity > 0) and the boundary between two graph
        launches is paid once per replay instead of once per token."""
        prime_graph_rng_state(device)
        self.model = model
        self.native_sampling = bool(native_sampling) and model.native_ready() and (sampling_kwargs.get("top_k") or 0) <= 64 \
            and sampling_kwargs.get("top_k") is not None
        self.fold_embed = bool(fold_embed) and self.native_sampling
        self.steps_per_replay = max(1, int(steps_per_replay)) if self.native_sampling else 1
        self.seed = seed
        self.rng_counter = torch.zeros((1, ), dtype=torch.int32, device=device)
        self.work_val = torch.zeros((128 * 64, ), dtype=torch.float32, device=device)
        self.work_idx = torch.zeros((128 * 64, ), dtype=torch.int32, device=device)
        self.tok = torch.zeros((1, 1), dtype=torch.int32, device=device)
        self.pos = torch.zeros((1, ), dtype=torch.int32, device=device)
        self.next_tok = torch.zeros((1, 1), dtype=torch.int32, device=device)
        self.next_prob = torch.zeros((1, model.config.vocab_size), dtype=torch.float32, device=device)
        self.seq = torch.zeros((int(seq_capacity), ), dtype=torch.int32, device=device) if seq_capacity else None
        self.ban = torch.zeros((6, ), dtype=torch.int32, device=device)
        self.sampling_kwargs = sampling_kwargs
        # warm up on a side stream (lazy kernel attributes, allocator), then capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with capture(self.graph):
            for _ in range(self.steps_per_replay):
                self._step()
        torch.cuda.synchronize()
        # (a single-step graph over the same state tensors for what is left of a sequence: `step_one`)
        self.graph1 = self.graph
        if self.steps_per_replay > 1:
            self.graph1 = torch.cuda.CUDAGraph()
            with capture(self.graph1):
                self._step()
            torch.cuda.synchronize()
        self._bound = self._signature()
        self._gen = getattr(model, "_alloc_gen", 0)

    def close(self):
        """destroy the captured graphs now (an owner that drops a DecodeGraph calls this instead of leaving the executable graphs to
        whatever moment the object is finalised -- never inside a later capture, see _graphs.py); the object is unusable afterwards"""
        g, g1 = getattr(self, "graph", None), getattr(self, "graph1", None)
        self.graph = self.graph1 = None
        release(g, g1 if g1 is not g else None)

    def _signature(self):
        """what the captured launches point at: the graph bakes in raw pointers (KV caches, RoPE tables, the native step's
        buffers) and max_seq_length; `setup_caches` with a longer length re-allocates all of them"""
        m = self.model
        sig = [m.max_seq_length, m.max_batch_size, m.rope_cos.data_ptr(), m.rope_sin.data_ptr()]
        sig += [b.attention.kv_cache.k_cache.data_ptr() for b in m.layers]
        # the weight tensors the captured launches read (model.to() / .half() or an in-place re-pair after a sub-module
        # load_state_dict assign new ones; `_alloc_gen` is bumped on those events, this is what the re-validation compares)
        sig += [t.data_ptr() for b in m.layers for t in list(b.buffers(recurse=True)) + list(b.parameters(recurse=True))]
        if m._native is not None:
            sig += [m._native["x"].data_ptr(), m._native["logits"].data_ptr()]
        return tuple(sig)

    def check_bound(self):
        if self._signature() != self._bound:
            raise RuntimeError("the captured decode graph is bound to caches / buffers that have since been re-allocated "
                               "(setup_caches with a longer max_seq_length?): capture a new DecodeGraph")

    def set_token(self, tok, pos):
        """the token and position of the next step (device tensors or ints); with the embedding folded into the sampler, also the
        hidden state of that token"""
        if torch.is_tensor(tok):
            self.tok.copy_(tok.view(1, 1))
        else:
            self.tok.fill_(int(tok))
        if torch.is_tensor(pos):
            self.pos.copy_(pos.view(1))
        else:
            self.pos.fill_(int(pos))
        if self.fold_embed:
            m = self.model
            with torch.cuda.device(m.output.weight.device):
                st = m._native_state()
                m.native_embed(self.tok.view(1), st["x"], st["ssq"] if self._ho0() else None)

    def _ho0(self):
        m = self.model
        return m._native_kind() == "ap" and bool(m._handover_plan(m.layers[0])["qkv_in"])

    def _step(self):
        if self.native_sampling:
            from . import _lib
            m = self.model
            if self.fold_embed:
                with torch.cuda.device(m.output.weight.device):
                    st = m._native_state()
                    ho0 = self._ho0()
                    m.native_layers(st["x"], self.pos.view(1), 0, len(m.layers), ssq_ready=ho0)
                    logits = m.native_head(st["x"])
                emb, xo, so = m.tok_embeddings.weight.data_ptr(), st["x"].data_ptr(), (st["ssq"].data_ptr() if ho0 else None)
            else:
                logits = m.decode_native(self.tok.view(1), self.pos.view(1))
                emb = xo = so = None
            kw = self.sampling_kwargs
            # (top_p: nucleus filter on the top-k survivors, transformers' warper chain -- 1 = off)
            _lib.check(_lib.lib().gq_sample_topk_p(logits.data_ptr(), m.config.vocab_size, int(kw["top_k"]), float(kw.get("top_p") or 1.0),
                                                  float(kw.get("temperature", 1.0)), int(self.seed), self.rng_counter.data_ptr(),
                                                   self.work_val.data_ptr(), self.work_idx.data_ptr(), self.tok.data_ptr(),
                                                   self.pos.data_ptr(), self.next_tok.data_ptr(), self.ban.data_ptr(),
                                                   self.seq.data_ptr() if self.seq is not None else None,
                                                   self.seq.numel() if self.seq is not None else 0, emb, xo, m.config.dim, so,
                                                   _lib.current_stream_ptr()),
                       "gq_sample_topk_p")
            return
        t, p = decode_one_token(self.model, self.tok, self.pos, **{k: v for k, v in self.sampling_kwargs.items() if k != "top_p"})
        self.next_tok.copy_(t)
        self.next_prob.copy_(p)

    def step(self, advance=True):
        """replay one token step; by default feeds the sampled token back and advances the position on device"""
        # one integer compare per token (the model bumps `_alloc_gen` whenever it re-allocates caches / native buffers); the
        # pointer-by-pointer comparison runs only after such an event
        gen = getattr(self.model, "_alloc_gen", 0)
        if gen != self._gen:
            self.check_bound()
            self._gen = gen
        self.graph.replay()
        if advance and not self.native_sampling:
            self.tok.copy_(self.next_tok)
            self.pos.add_(1)

    def step_one(self):
        """one token step whatever `steps_per_replay` is (the tail of a sequence whose length is not a multiple of it)"""
        if self.steps_per_replay == 1:
            return self.step()
        gen = getattr(self.model, "_alloc_gen", 0)
        if gen != self._gen:
            self.check_bound()
            self._gen = gen
        self.graph1.replay()


def decode_n_tokens(model: Transformer, cur_token: torch.Tensor, input_pos: torch.Tensor, num_new_tokens: int,
                    use_graph=True, callback=lambda _: _, graph: Optional[DecodeGraph] = None, **sampling_kwargs):
    new_tokens, new_probs = [], []
    if use_graph:
        g = graph or DecodeGraph(model, cur_token.device, **sampling_kwargs)
        g.set_token(cur_token, input_pos)
        for _ in range(num_new_tokens):
            g.step()
            new_tokens.append(g.next_tok.clone())
            callback(new_tokens[-1])
            new_probs.append(g.next_prob.clone())
    else:
        for _ in range(num_new_tokens):
            next_token, next_prob = decode_one_token(model, cur_token, input_pos, **sampling_kwargs)
            input_pos = input_pos + 1
            new_tokens.append(next_token.clone())
            callback(new_tokens[-1])
            new_probs.append(next_prob.clone())
            cur_token = next_token.clone().view(1, 1)
    if cur_token.is_cuda:
        torch.cuda.synchronize()
    return new_tokens, new_probs


@torch.no_grad()
def generate(model: Transformer, prompt: torch.Tensor, max_new_tokens: int, batch_size: int = 1, callback=lambda x: x,
             use_graph=True, graph: Optional[DecodeGraph] = None, **sampling_kwargs) -> torch.Tensor:
    T = prompt.size(-1)
    T_new = T + max_new_tokens
    max_seq_length = min(T_new, model.config.block_size)
    device, dtype = prompt.device, prompt.dtype
    model.setup_caches(max_batch_size=batch_size, max_seq_length=max_seq_length)
    seq = torch.empty(batch_size, T_new, dtype=dtype, device=device)
    prompt = prompt.view(1, -1).repeat(batch_size, 1)
    seq[:, :T] = prompt
    input_pos = torch.arange(0, T, device=device, dtype=torch.int32)
    if T != 1:
        next_token = prefill(model, prompt.view(batch_size, -1), input_pos, **sampling_kwargs).clone()
        seq[:, T] = next_token.squeeze()
        input_pos = torch.tensor([T], device=device, dtype=torch.int32).view(1)
        generated, _ = decode_n_tokens(model, next_token.view(batch_size, -1), input_pos, max_new_tokens - 1,
                                       use_graph=use_graph, callback=callback, graph=graph, **sampling_kwargs)
        seq[:, T + 1:] = torch.cat(generated, dim=-1)
    else:
        generated, _ = decode_n_tokens(model, prompt.view(batch_size, -1), input_pos, max_new_tokens, use_graph=use_graph,
                                       callback=callback, graph=graph, **sampling_kwargs)
        seq[:, 1:] = torch.cat(generated, dim=-1)
    return seq


def random_init_(model: Transformer, seed: int = 0, lut_std: float = 0.02, cheap: bool = True):
    """`--random_init` equivalent: fill every parameter/buffer with synthetic values of the right format.
    Quantized linears get uniformly random codes (= uniformly random plane words, the packing is a bijection) and
    per-row sorted N(0, lut_std^2) centroids."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    dev = model.output.weight.device
    gd = torch.Generator(device=dev)
    gd.manual_seed(seed)
    for name, mod in model.named_modules():
        if isinstance(mod, APLinear):
            mod.qweight.copy_(torch.randint(-2**31, 2**31 - 1, mod.qweight.shape, dtype=torch.int32, device=dev, generator=gd))
            lut = (torch.randn(mod.lut.shape, device=dev, generator=gd) * lut_std).sort(dim=1).values
            mod.lut.copy_(lut.to(mod.lut.dtype))
        elif isinstance(mod, LUTGEMMLinear):
            mod.qweight.copy_(torch.randint(-2**31, 2**31 - 1, mod.qweight.shape, dtype=torch.int32, device=dev, generator=gd))
            mod.alpha.copy_((torch.rand(mod.alpha.shape, device=dev, generator=gd) * lut_std).to(mod.alpha.dtype))
            mod.q_bias.copy_((torch.randn(mod.q_bias.shape, device=dev, generator=gd) * lut_std).to(mod.q_bias.dtype))
        elif type(mod).__name__ == "QuantizedLinear":
            # QTIP: random trellis words, a smooth symmetric codebook, sign vectors
            mod.trellis.copy_(torch.randint(-2**15, 2**15 - 1, mod.trellis.shape, dtype=torch.int16, device=dev, generator=gd))
            if mod.tlut is not None:
                mod.tlut.data.copy_((torch.randn(mod.tlut.shape, device=dev, generator=gd) * 0.5).to(mod.tlut.dtype))
            mod.SU.copy_((torch.randint(0, 2, mod.SU.shape, device=dev, generator=gd) * 2 - 1).to(mod.SU.dtype))
            mod.SV.copy_((torch.randint(0, 2, mod.SV.shape, device=dev, generator=gd) * 2 - 1).to(mod.SV.dtype) * lut_std)
        elif isinstance(mod, nn.Linear):
            mod.weight.data.copy_((torch.randn(mod.weight.shape, device=dev, generator=gd) * 0.02).to(mod.weight.dtype))
        elif isinstance(mod, nn.Embedding):
            mod.weight.data.copy_((torch.randn(mod.weight.shape, device=dev, generator=gd) * 0.02).to(mod.weight.dtype))
    return model


def load_model(model_name, device, backend, bitwidth, random_init=True, checkpoint_path=None, dtype=torch.float16,
               halve_layers=False):
    linear_kwargs = {}
    if backend == "ap":
        linear_class = APLinear
        linear_kwargs["bitwidth"] = bitwidth
        linear_kwargs["device"] = device
    elif backend == "lutgemm":
        linear_class = LUTGEMMLinear
        linear_kwargs["bitwidth"] = bitwidth
        linear_kwargs["group_size"] = -1
        linear_kwargs["device"] = device
    elif backend == "qtip":
        # generate.py:210-221: the QTIP hyper-parameters come from the checkpoint's config.json ("quip_params"); the
        # linears are not fused (generate.py:232) and the state dict is loaded non-strictly (generate.py:238)
        from .qtip import QuantizedLinear
        import json
        import os
        quip = dict(td_x=16, td_y=16, L=16, K=bitwidth, V=2, tlut_bits=9, decode_mode="quantlut_sym")
        if checkpoint_path is not None and os.path.exists(os.path.join(checkpoint_path, "config.json")):
            with open(os.path.join(checkpoint_path, "config.json"), "r") as f:
                quip.update({k: v for k, v in json.load(f)["quip_params"].items() if k in quip})
        linear_class = QuantizedLinear
        linear_kwargs.update(quip)
        linear_kwargs["device"] = device
    elif backend is None:
        linear_class = nn.Linear
        assert bitwidth == 16
    else:
        raise ValueError(f"unknown backend {backend!r} (ap | lutgemm | qtip | None)")
    model = Transformer.from_name(name=model_name, dtype=dtype, linear_class=linear_class, linear_kwargs=linear_kwargs,
                                  halve_layers=halve_layers, fuse_linears=(backend != "qtip"))
    if not random_init:
        import os
        checkpoint = torch.load(os.path.join(checkpoint_path, "converted_pytorch_model.bin"), mmap=True, weights_only=True)
        model.load_state_dict(checkpoint, assign=True, strict=(backend != "qtip"))
    model = model.to(device=device, dtype=dtype)
    if random_init:
        random_init_(model)
    return model.eval()


def _get_model_size(model):
    """bytes of parameters AND buffers of every non-Embedding child (generate.py:247-266)"""
    model_size = 0
    params = 0
    for name, child in model.named_children():
        if not isinstance(child, torch.nn.Embedding):
            model_size += sum(p.numel() * p.dtype.itemsize for p in itertools.chain(child.parameters(), child.buffers()))
            params += sum(p.numel() for p in itertools.chain(child.parameters(), child.buffers()))
    return model_size, params


def benchmark_decode(model: Transformer, device, num_samples=5, max_new_tokens=100, top_k=32, temperature=0.0, seed=1234,
                     bos_id=128000, native_sampling=False):
    """tokens/s as the reference measures it (generate.py:344-389): BOS-only prompt, one warm-up generate, then
    `num_samples` timed generate() calls; mean/std of tokens/s and model-level bandwidth."""
    torch.manual_seed(seed)
    prompt = torch.tensor([bos_id % model.config.vocab_size], dtype=torch.int32, device=device)
    model.setup_caches(1, 1 + max_new_tokens)
    graph = DecodeGraph(model, device, native_sampling=native_sampling, temperature=temperature, top_k=top_k)
    tps = []
    for i in range(-1, num_samples):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = generate(model, prompt, max_new_tokens, use_graph=True, graph=graph, temperature=temperature, top_k=top_k)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        if i >= 0:
            tps.append((y.size(-1) - prompt.size(-1)) / t)
    model_size, _ = _get_model_size(model)
    return dict(tokens_per_sec=float(np.mean(tps)), std=float(np.std(tps)), bandwidth_GBps=model_size * float(np.mean(tps)) / 1e9,
                model_size=model_size)


def main(prompt=None, num_samples=5, max_new_tokens=100, batch_size=1, top_k=200, temperature=0.8, compile=2,
         compile_prefill=False, profile=None, device="cuda", model_name=None, backend=None, bitwidth=None,
         checkpoint_path=None, config_path=None, dtype=None, print_result=False, random_init=False):
    """generate.py:268-389 -- same arguments and printed report.  `compile` selects the decode driver: 0 = eager per-token
    launches, >= 1 = the captured hipGraph of the decode step (the role torch.compile(mode='max-autotune') plays in the
    reference).  A tokenizer is only needed for a text prompt / --print_result; without one the BOS id of the model
    family is used (the reference's benchmark setting, prompt=None)."""
    print(f"Using device={device}")
    dtype = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32, None: torch.float16}.get(dtype, dtype)
    if batch_size != 1:
        raise ValueError("only batch_size 1 is implemented (generate.py:165)")
    t0 = time.time()
    model = load_model(model_name, device, backend, bitwidth, random_init=random_init, checkpoint_path=checkpoint_path, dtype=dtype)
    tokenizer = None
    if prompt is not None or print_result:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_name)
    if "cuda" in str(device):
        torch.cuda.synchronize()
    print(f"Time to load model: {time.time() - t0:.02f} seconds", flush=True)
    if prompt is not None:
        encoded = torch.tensor(tokenizer.encode(prompt), dtype=torch.int32, device=device)
    else:
        bos = tokenizer.bos_token_id if tokenizer is not None else (128000 if model.config.vocab_size > 100000 else 1)
        encoded = torch.tensor([bos], dtype=torch.int32, device=device)
    prompt_length = encoded.size(-1)
    torch.manual_seed(1234)
    model_size, params = _get_model_size(model)
    model.setup_caches(1, prompt_length + max_new_tokens)
    use_graph = bool(compile) and "cuda" in str(device)
    graph = DecodeGraph(model, device, temperature=temperature, top_k=top_k) if use_graph else None
    tps = []
    for i in range(-1 if use_graph else 0, num_samples):
        if "cuda" in str(device):
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        y = generate(model, encoded, max_new_tokens, use_graph=use_graph, graph=graph, temperature=temperature, top_k=top_k)
        if "cuda" in str(device):
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t1
        if i == -1:
            print(f"Compilation time: {elapsed:.2f} seconds", flush=True)
            continue
        if print_result:
            print(tokenizer.decode(y.reshape(-1).tolist()))
        tokens_sec = (y.size(-1) - prompt_length) / elapsed
        tps.append(tokens_sec)
        if i + 1 == num_samples:
            print(f"Time for inference {i + 1}: {elapsed:.02f} sec total, {tokens_sec:.02f} tokens/sec")
            print(f"Bandwidth achieved: {model_size * tokens_sec / 1e9:.02f} GB/s")
            print(f"FLOPS achieved: {params * (y.numel() / elapsed) * 2 / 1e12:.02f} TF/s")
            print(flush=True)
    print("==========")
    print(f"Batch Size: {batch_size}")
    print(f"Prompt Length: {prompt_length}")
    print(f"Generated tokens: {max_new_tokens}")
    print(f"Average tokens/sec: {float(np.mean(tps)):.2f}")
    print(f"Std of tokens/sec: {float(np.std(tps, ddof=1)) if len(tps) > 1 else 0.0:.2f}")
    return tps


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser(description="Decode benchmark / text generation (flags of the reference's inference/generate.py)")
    parser.add_argument('--prompt', type=str, default=None, help="Input prompt. None: only the bos token (benchmarking)")
    parser.add_argument('--num_samples', type=int, default=5)
    parser.add_argument('--max_new_tokens', type=int, default=100)
    parser.add_argument('--batch_size', type=int, default=1)
    parser.add_argument('--top_k', type=int, default=32)
    parser.add_argument('--temperature', type=float, default=0.0)
    parser.add_argument('--compile', type=int, default=2, help='0: eager launches, otherwise the captured hipGraph decode step')
    parser.add_argument('--compile_prefill', action='store_true', help='accepted for compatibility (prefill is not captured)')
    parser.add_argument('--profile', type=str, default=None, help='accepted for compatibility (use rocprofv3)')
    parser.add_argument('--device', type=str, default="cuda")
    parser.add_argument('--model_name', type=str, default=None)
    parser.add_argument('--bitwidth', type=int, default=None, choices=[2, 3, 4, 16])
    parser.add_argument('--checkpoint_path', type=str, default=None)
    parser.add_argument('--config_path', type=str, default=None, help='QTIP config path')
    parser.add_argument('--dtype', type=str, default="float16", choices=["float16", "float32", "bfloat16"])
    parser.add_argument('--backend', type=str, default=None, choices=["ap", "lutgemm", "qtip", None])
    parser.add_argument('--print_result', action='store_true')
    parser.add_argument('--random_init', action='store_true')
    a = parser.parse_args()
    main(a.prompt, a.num_samples, a.max_new_tokens, a.batch_size, a.top_k, a.temperature, a.compile, a.compile_prefill, a.profile,
         a.device, a.model_name, a.backend, a.bitwidth, a.checkpoint_path, a.config_path, a.dtype, a.print_result, a.random_init)
