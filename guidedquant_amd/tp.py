"""Tensor-parallel (row-split) bs=1 decode of a fused Any-Precision model over W ranks (SURVEY.md section 8e "later"; VERDICT r4
item 8).  The rows of a LUT-GEMV are independent (inference/ap_gemv/anyprec.cu:387: one output row per warp), so every matrix of a
layer can be cut along its OUTPUT rows without touching the arithmetic: rank r holds rows [r N/W, (r+1) N/W) of wo / w2, its H/W query
heads and Hkv/W key / value heads of wqkv (with their KV caches), and its I/W (gate, up) pairs of w1w3.  Each of the four GEMV outputs
of a layer is all-gathered -- 2N bytes per rank and matrix -- and the element-wise pieces (RMSNorm prologue, RoPE, attention of the
rank's heads, silu * up, residual adds) run where their rows live.  This is the only multi-GPU form that cuts the latency of ONE
sequence (the layer pipeline of pipeline.py only raises aggregate throughput).

Transport: the device-to-device hop primitives of csrc/hop.hip.  Every rank owns four fine-grained landing buffers (attention
output, post-attention hidden state, silu * up vector, layer output) and one sequence word per (buffer, peer); a rank writes its slice
into every peer's landing buffer (gq_hop_send: system-scope stores, then the word) and copies the peers' slices out of its own
(gq_hop_wait_copy) -- no host synchronisation, no RCCL call; a decode step is one hipGraph per rank.  Embedding, final norm, lm_head and
the sampler are replicated (same seed, same logits -> the same token on every rank: nothing to exchange).

Status: FUNCTIONAL.  Tested with 2 and 4 ranks sharing one MI355X (tests/test_tp_gpu.py: tokens equal to the single-process decode); no
multi-GPU node was available, so nothing is claimed about xGMI or speed (W - 1 send + W - 1 wait launches per exchange, 8 (W - 1) + 5
launches per layer; an all-gather kernel per exchange is the obvious next step once there is hardware to measure it on)."""
import ctypes
import math
import os

import torch
import torch.distributed as dist

from . import _lib
from ._graphs import capture
from .model import Transformer, _pair_perm


class _Raw:  # a torch view of raw device memory (CUDA array interface)
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(n, ), typestr=typestr, data=(ptr, False), version=2)


class TensorParallelDecoder:

    def __init__(self, model: Transformer, group, rank: int, world: int, max_new_tokens: int, temperature=0.0, top_k=32, seed=1234, bos_id=1):
        c = model.config
        assert model.fuse_linears, "tensor-parallel decode takes the fused (wqkv / w1w3) Any-Precision model"
        assert c.n_head % world == 0 and c.n_local_heads % world == 0 and c.dim % (16 * world) == 0 and c.intermediate_size % (16 * world) == 0, \
            "heads, KV heads, dim / 16 and intermediate / 16 must divide by the world size"
        self.model, self.group, self.rank, self.world = model, group, rank, world
        self.temperature, self.top_k, self.seed, self.bos_id = temperature, top_k, seed, bos_id
        self.max_new_tokens = max_new_tokens
        dev = model.output.weight.device
        self.dev = dev
        model.setup_caches(1, 1 + max_new_tokens)
        model._native_state()  # (buffers of the replicated head; pairs the full model's gate / up rows once, outside any capture)
        self.L = _lib.lib()
        W, r, hd = world, rank, c.head_dim
        self.Hl, self.Kvl = c.n_head // W, c.n_local_heads // W
        self.Dl, self.Il = c.dim // W, c.intermediate_size // W
        f16 = dict(dtype=torch.float16, device=dev)
        # ---- this rank's rows of every matrix (contiguous copies; a deployment would load only these)
        self.layers = []
        qh, kh = c.n_head * hd, c.n_local_heads * hd
        for b in model.layers:
            at, ff = b.attention, b.feed_forward
            if getattr(ff.w1w3, "gq_row_pairs", False):  # back to [w1; w3] first
                inv = torch.argsort(_pair_perm(c.intermediate_size, dev))
                w13q, w13l = ff.w1w3.qweight[:, inv, :], ff.w1w3.lut[inv]
            else:
                w13q, w13l = ff.w1w3.qweight, ff.w1w3.lut
            rows_qkv = torch.cat([torch.arange(r * self.Hl * hd, (r + 1) * self.Hl * hd, device=dev),
                                  qh + torch.arange(r * self.Kvl * hd, (r + 1) * self.Kvl * hd, device=dev),
                                  qh + kh + torch.arange(r * self.Kvl * hd, (r + 1) * self.Kvl * hd, device=dev)])
            gate = torch.arange(r * self.Il, (r + 1) * self.Il, device=dev)
            rows_gu = torch.stack((gate, c.intermediate_size + gate), dim=1).reshape(-1)  # (gate_i, up_i) pairs: the GQ_EPI_SILU_PAIRS order
            rows_o = torch.arange(r * self.Dl, (r + 1) * self.Dl, device=dev)
            self.layers.append(dict(
                bits=at.wqkv.bitwidth,
                qkv=(at.wqkv.qweight[:, rows_qkv, :].contiguous(), at.wqkv.lut[rows_qkv].contiguous()),
                o=(at.wo.qweight[:, rows_o, :].contiguous(), at.wo.lut[rows_o].contiguous()),
                gu=(w13q[:, rows_gu, :].contiguous(), w13l[rows_gu].contiguous()),
                d=(ff.w2.qweight[:, rows_o, :].contiguous(), ff.w2.lut[rows_o].contiguous()),
                n1=b.input_layernorm.weight, n2=b.post_attention_layernorm.weight,
                kc=torch.zeros(self.Kvl, model.max_seq_length, hd, **f16), vc=torch.zeros(self.Kvl, model.max_seq_length, hd, **f16)))
        # ---- working buffers (ordinary memory: only this rank's kernels touch them)
        self.x, self.y, self.h = (torch.zeros(c.dim, **f16) for _ in range(3))
        self.gu = torch.zeros(c.intermediate_size, **f16)
        self.qkv = torch.zeros((self.Hl + 2 * self.Kvl) * hd, **f16)
        self.tok = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.next_tok = torch.zeros(1, dtype=torch.int32, device=dev)
        self.rng_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.work_val = torch.zeros(128 * 64, dtype=torch.float32, device=dev)
        self.work_idx = torch.zeros(128 * 64, dtype=torch.int32, device=dev)
        self.out_buf = torch.zeros(max_new_tokens + 1, dtype=torch.int32, device=dev)
        self.tick = torch.zeros(1, dtype=torch.int32, device=dev)   # exchanges completed before this step: step * 4 * n_layer
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.spins = int(os.environ.get("GQ_HOP_SPINS", str(1 << 21)))
        self._setup_exchange()
        self._capture()

    # ------------------------------------------------------------------------------------------------------------ exchange
    def _setup_exchange(self):
        """fine-grained landing buffers [y | h | gu | x] + sequence words [4][W], shared with every peer through hipIpc handles"""
        c, W, L = self.model.config, self.world, self.L
        sizes = [c.dim * 2, c.dim * 2, c.intermediate_size * 2, c.dim * 2]
        self._off = [sum(sizes[:k]) for k in range(4)]
        self._seq_off = sum(sizes)
        nbytes = self._seq_off + 4 * 4 * W
        with torch.cuda.device(self.dev):
            p = ctypes.c_void_p()
            _lib.check(L.gq_hop_alloc(nbytes, ctypes.byref(p)), "gq_hop_alloc")
            self._fg = int(p.value)
            handle = (ctypes.c_ubyte * 64)()
            _lib.check(L.gq_hop_export(self._fg, handle), "gq_hop_export")
            if L.gq_hop_is_finegrained(self._fg) != 1:  # (peers on other devices write here: see pipeline.py::_setup_ipc)
                raise RuntimeError("tensor-parallel exchange: the landing block is not fine-grained device memory (gq_hop_alloc)")
        self._seq = torch.as_tensor(_Raw(self._fg + self._seq_off, 4 * W, "<i4"), device=self.dev)
        self._seq.zero_()
        torch.cuda.synchronize()
        allh = [None] * W
        dist.all_gather_object(allh, dict(handle=bytes(handle), pid=os.getpid()), group=self.group)
        self._peer, self._maps = {}, []
        for q in range(W):
            if q == self.rank:
                continue
            if allh[q]["pid"] == os.getpid():
                base = self._fg
            else:
                with torch.cuda.device(self.dev):
                    m = ctypes.c_void_p()
                    buf = (ctypes.c_ubyte * 64).from_buffer_copy(allh[q]["handle"])
                    _lib.check(L.gq_hop_import(buf, ctypes.byref(m)), "gq_hop_import")
                base = int(m.value)
                self._maps.append(base)
            self._peer[q] = base
        dist.barrier(group=self.group)

    def close(self):
        for m in self._maps:
            self.L.gq_hop_close(m)
        self._maps = []
        if getattr(self, "_fg", None):
            self._seq = None
            self.L.gq_hop_free(self._fg)
            self._fg = None

    def _allgather(self, kind: int, buf: torch.Tensor, slice_elems: int, add: int):
        """this rank's slice of `buf` (fp16, full vector) -> every peer's landing buffer; the peers' slices -> `buf`"""
        if not self._exchange:
            return
        L, st, W, r = self.L, _lib.current_stream_ptr(), self.world, self.rank
        nb = slice_elems * 2
        for q in range(W):
            if q == r:
                continue
            _lib.check(L.gq_hop_send(buf.data_ptr() + r * nb, self._peer[q] + self._off[kind] + r * nb, nb,
                                     self._peer[q] + self._seq_off + 4 * (kind * W + r), self.tick.data_ptr(), add, st), "gq_hop_send")
        for q in range(W):
            if q == r:
                continue
            _lib.check(L.gq_hop_wait_copy(self._fg + self._seq_off + 4 * (kind * W + q), self.tick.data_ptr(), add, self.err.data_ptr(), self.spins,
                                          self._fg + self._off[kind] + q * nb, buf.data_ptr() + q * nb, nb, st), "gq_hop_wait_copy")

    # ------------------------------------------------------------------------------------------------------------ one step
    def _step(self):
        m, c, L, st, r = self.model, self.model.config, self.L, _lib.current_stream_ptr(), self.rank
        ck = _lib.check
        hd = c.head_dim
        x, y, h, gu, qkv = self.x, self.y, self.h, self.gu, self.qkv
        scale = 1.0 / math.sqrt(hd)
        m.native_embed(self.tok, x)
        nq = (self.Hl + 2 * self.Kvl) * hd
        for li, d in enumerate(self.layers):
            bits = d["bits"]
            add = 4 * li
            if L.gq_anyprec_qkv_rope_supported(nq, c.dim, bits, hd):
                ck(L.gq_anyprec_gemv_qkv_rope(x.data_ptr(), qkv.data_ptr(), d["qkv"][0].data_ptr(), d["qkv"][1].data_ptr(), nq, c.dim, bits,
                                              d["n1"].data_ptr(), c.norm_eps, self.pos.data_ptr(), m.rope_cos.data_ptr(), m.rope_sin.data_ptr(),
                                              d["kc"].data_ptr(), d["vc"].data_ptr(), self.Hl, self.Kvl, hd, m.max_seq_length, st), "wqkv+rope")
                ck(L.gq_attn_decode_roped(qkv.data_ptr(), self.pos.data_ptr(), d["kc"].data_ptr(), d["vc"].data_ptr(),
                                          y.data_ptr() + r * self.Hl * hd * 2, self.Hl, self.Kvl, hd, m.max_seq_length, scale, 1, None, st), "attn")
            else:
                ck(L.gq_anyprec_gemv_fused(x.data_ptr(), qkv.data_ptr(), d["qkv"][0].data_ptr(), d["qkv"][1].data_ptr(), nq, c.dim, bits,
                                           d["n1"].data_ptr(), c.norm_eps, None, 0, st), "wqkv")
                ck(L.gq_attn_decode_split(qkv.data_ptr(), self.pos.data_ptr(), m.rope_cos.data_ptr(), m.rope_sin.data_ptr(), d["kc"].data_ptr(),
                                          d["vc"].data_ptr(), y.data_ptr() + r * self.Hl * hd * 2, self.Hl, self.Kvl, hd, m.max_seq_length, scale, 1,
                                          None, st), "attn")
            self._allgather(0, y, self.Hl * hd, add + 1)
            ck(L.gq_anyprec_gemv_fused(y.data_ptr(), h.data_ptr() + r * self.Dl * 2, d["o"][0].data_ptr(), d["o"][1].data_ptr(), self.Dl, c.dim, bits,
                                       None, 0.0, x.data_ptr() + r * self.Dl * 2, 1, st), "wo")
            self._allgather(1, h, self.Dl, add + 2)
            ck(L.gq_anyprec_gemv_fused(h.data_ptr(), gu.data_ptr() + r * self.Il * 2, d["gu"][0].data_ptr(), d["gu"][1].data_ptr(), 2 * self.Il, c.dim,
                                       bits, d["n2"].data_ptr(), c.norm_eps, None, 4, st), "w1w3")
            self._allgather(2, gu, self.Il, add + 3)
            ck(L.gq_anyprec_gemv_fused(gu.data_ptr(), x.data_ptr() + r * self.Dl * 2, d["d"][0].data_ptr(), d["d"][1].data_ptr(), self.Dl,
                                       c.intermediate_size, bits, None, 0.0, h.data_ptr() + r * self.Dl * 2, 1, st), "w2")
            self._allgather(3, x, self.Dl, add + 4)
        logits = m.native_head(x)
        ck(L.gq_sample_topk_ex(logits.data_ptr(), c.vocab_size, int(self.top_k), float(self.temperature), int(self.seed), self.rng_counter.data_ptr(),
                               self.work_val.data_ptr(), self.work_idx.data_ptr(), self.tok.data_ptr(), self.pos.data_ptr(), self.next_tok.data_ptr(),
                               None, self.out_buf.data_ptr(), self.out_buf.numel(), None, None, 0, None, st), "sample")
        self.tick.add_(4 * len(self.layers))

    def _capture(self):
        from .generate import prime_graph_rng_state
        prime_graph_rng_state(self.dev)
        self._exchange = False  # warm-up without the hops (lazy kernel attributes, allocator): no peer is stepping yet
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._exchange = True
        self.graph = torch.cuda.CUDAGraph()
        with capture(self.graph), torch.no_grad():
            self._step()
        torch.cuda.synchronize()
        self.reset()

    def reset(self):
        self.tok.fill_(self.bos_id)
        self.pos.zero_()
        self.rng_counter.zero_()
        self.tick.zero_()
        self._seq.zero_()
        self.out_buf.zero_()
        self.err.zero_()

    def run(self, n_tokens: int) -> torch.Tensor:
        """decode n_tokens from BOS; returns the tokens (int32 [n_tokens]), the same on every rank"""
        if n_tokens > self.max_new_tokens:
            raise ValueError(f"run({n_tokens}): built for max_new_tokens = {self.max_new_tokens}")
        self.reset()
        torch.cuda.synchronize()
        dist.barrier(group=self.group)  # every rank has cleared its sequence words
        for _ in range(n_tokens):
            self.graph.replay()
        torch.cuda.synchronize()
        if int(self.err.item()):
            raise RuntimeError("tensor-parallel decode: a rank waited for a peer's slice beyond the spin limit (GQ_HOP_SPINS)")
        dist.barrier(group=self.group)  # nobody resets while a peer may still be sending
        return self.out_buf[1:n_tokens + 1].clone()
