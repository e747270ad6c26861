"""Layer-pipelined bs=1 decode across ranks (the reference's precedent is the sequential layer
sharding of qtip/lib/utils/shard_model.py:30-68 and accelerate's device_map="auto",
any_precision/modules/AnyPrecisionForCausalLM.py:76-82 -- both single-process `tensor.to(device)` hops).

MI355X-native form: one process per GPU, stage g owns layers [l0, l1) (embedding on stage 0, final norm + lm_head +
sampling on the last stage) and the KV caches of those layers.  The only data-path exchange is point to point: the fp16
hidden state [dim] (16 KiB for dim = 8192) hops g -> g+1, the sampled token id hops from the last stage back to stage 0
(torch.distributed isend / irecv; backend "nccl" = RCCL over one xGMI link per hop, "gloo" on CPU for the tests).  No
collective is on the data path.

A single bs=1 stream cannot go faster than on one GPU (strict layer dependency); with S = world_size independent
sequences ("slots") in flight every stage is busy every tick and the aggregate rate approaches S x the per-stage rate.

What one stage tick is on the GPU (native path):
  * ONE hipGraph replay per (stage, slot): [embedding lookup] + the stage's layers on that slot's KV caches + the position
    increment [+ final norm, lm_head, top-k sampling], captured once per slot at start-up -- no per-launch host work;
  * token position per slot in DEVICE memory (`pos_dev[slot]`, incremented inside the graph), nothing is filled from the host;
  * every slot has its own hidden-state buffer, so the receive of slot s+1 is POSTED before slot s is computed and lands
    while the stage computes (the hop latency hides under ~0.5 ms of layers); sends are asynchronous; a buffer is reused S
    ticks later, after its send has completed;
  * forward hops and the token feedback use two process groups: with one communicator per rank pair, posted-ahead receives
    and sends of the two directions between the same pair (world = 2) would wait for each other in stream order.
  * transport: device tensors go straight into isend / irecv (RCCL).  With a backend that only moves host memory (gloo; used
    to run the multi-rank native path on ONE GPU in the tests) each hop is staged through a pinned host buffer -- same
    schedule, same graphs.
  * hop="ipc" (GQ_PP_HOP=ipc; round 4): no send / receive calls at all.  Every stage maps the NEXT stage's hidden-state slots
    and sequence words (torch's CUDA IPC: the peer may be another process on the same GPU -- how the tests run it -- or a peer
    GPU over xGMI), its tick graph ENDS with gq_hop_send (system-scope stores of the 8-16 KiB vector, then the slot's sequence
    word = its tick number) and BEGINS with gq_hop_wait (one wave polls its own sequence word, bounded); the sampled token returns
    to stage 0 the same way.  The host replays graphs and never synchronises inside a run.
`stage_ranges(.., head_cost_layers=measure_head_cost(model))` balances the stages with the measured cost of the head.
`PipelinedDecoder.run(n_tokens)` decodes n_tokens for each of the S sequences and returns them (on every rank).
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from ._graphs import capture


def stage_ranges(n_layer: int, world: int, head_cost_layers: float = 0.0) -> List[range]:
    """contiguous layer ranges per stage; `head_cost_layers` = cost of final norm + lm_head + sampling in units of one
    layer, charged to the last stage so that the stages are balanced"""
    total = n_layer + head_cost_layers
    per = total / world
    cuts, acc = [0], 0.0
    for s in range(world - 1):
        acc += per
        cuts.append(min(n_layer, max(cuts[-1] + (1 if n_layer - cuts[-1] > world - 1 - s else 0), int(round(acc)))))
    cuts.append(n_layer)
    return [range(cuts[i], cuts[i + 1]) for i in range(world)]


def measure_head_cost(model, reps: int = 20) -> float:
    """cost of the last stage's extra work (final norm + lm_head GEMV + sampling) in units of one decoder layer, measured
    with events on this GPU (native path); averaged over the ranks when a process group is up, so that every rank derives
    the same stage ranges.  Replaces round 1's guessed constant."""
    assert model.native_ready(), "measure_head_cost needs the native decode step"
    dev = model.output.weight.device
    st = model._native_state()
    x = st["x"]
    pos = torch.zeros(1, dtype=torch.int32, device=dev)
    n = min(4, len(model.layers))
    s = torch.cuda.current_stream()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    with torch.no_grad():
        t_layers = timed(lambda: model.native_layers(x, pos, 0, n)) / n
        t_head = timed(lambda: model.native_head(x)) + 0.02  # + the sampler (~16-20 us)
    x.zero_()
    ratio = torch.tensor([t_head / max(t_layers, 1e-6)], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(ratio)
        ratio /= dist.get_world_size()
    return round(float(ratio.item()), 1)


class PipelinedDecoder:

    def __init__(self, model, rank: int, world: int, layers: range, n_seq: Optional[int] = None, max_new_tokens: int = 100,
                 temperature: float = 0.0, top_k: Optional[int] = 32, bos_id: int = 1, native: Optional[bool] = None, group=None,
                 feedback_group=None, use_graphs: Optional[bool] = None, seed: int = 1234, hop: Optional[str] = None):
        self.model, self.rank, self.world, self.layers = model, rank, world, layers
        self.n_seq = S = n_seq or world
        self.group = group
        # a second communicator for the token feedback (see module docstring); the caller may pass one, else it is created
        # collectively here (every rank constructs the decoder)
        self.fb_group = feedback_group
        if self.fb_group is None and world > 1 and (hop or os.environ.get("GQ_PP_HOP", "p2p")) != "ipc":
            self.fb_group = dist.new_group(ranks=list(range(world)))
        self.hop = (hop or os.environ.get("GQ_PP_HOP", "p2p")) if world > 1 else "p2p"
        assert self.hop in ("p2p", "ipc")
        if self.fb_group is None and world > 1 and self.hop == "ipc":
            self.fb_group = group  # (no feedback communicator needed: the token returns through mapped memory)
        self.first, self.last = rank == 0, rank == world - 1
        self.temperature, self.top_k, self.bos_id, self.seed = temperature, top_k, bos_id, seed
        dev = model.output.weight.device
        self.dev = dev
        model.setup_caches(S, 1 + max_new_tokens)
        self.native = model.native_ready() if native is None else native
        c = model.config
        dt = model.output.weight.dtype
        self.h = torch.zeros(S, c.dim, dtype=dt, device=dev)          # per-slot hidden state (received into / computed in place / sent)
        # (token rows of 16 bytes: the device-to-device hop moves 16-byte units)
        self._tok4 = torch.zeros(S, 4, dtype=torch.int32, device=dev)
        self._tok_out4 = torch.zeros(S, 4, dtype=torch.int32, device=dev)
        self.tok = self._tok4[:, 0]          # per-slot input token (stage 0)
        self.tok_out = self._tok_out4[:, 0]  # per-slot sampled token (last stage)
        self.pos_dev = torch.zeros(S, dtype=torch.int32, device=dev)  # per-slot position, advanced on the device
        self.pos = [0] * S                                            # host mirror (eager path only)
        self.use_graphs = (self.native and dev.type == "cuda") if use_graphs is None else use_graphs
        self.graphs = None
        # host staging: the process group cannot move device memory (gloo) but the model lives on a GPU
        self.staged = world > 1 and dev.type == "cuda" and dist.get_backend(group) == "gloo" and self.hop != "ipc"
        if self.staged:
            self.h_host = torch.zeros(S, c.dim, dtype=dt).pin_memory()
            self.tok_host = torch.zeros(S, dtype=torch.int32).pin_memory()
        if self.native and self.last:
            self.rng_counter = torch.zeros(1, dtype=torch.int32, device=dev)
            self.work_val = torch.zeros(128 * 32, dtype=torch.float32, device=dev)
            self.work_idx = torch.zeros(128 * 32, dtype=torch.int32, device=dev)
        if self.native:
            # the QTIP launch plans are bound to the model's own hidden-state buffer: compute there, copy in / out
            self._x = model._native_state()["x"] if model._native_kind() == "qtip" else None
        if self.hop == "ipc":
            assert self.native and self.use_graphs and dev.type == "cuda", "hop='ipc' is the native graph path on GPUs"
            self._setup_ipc(max_new_tokens)
        if self.use_graphs:
            self._capture()

    # -- device-to-device hops (hop="ipc") ------------------------------------------------------------------------
    def _setup_ipc(self, max_new_tokens):
        """sequence words, tick counters, and the peers' landing slots mapped into this process.  Everything a PEER writes -- the
        landing slots of the hidden states / tokens and the sequence words -- is FINE-GRAINED memory of this stage's device
        (gq_hop_alloc: ordinary allocations are coarse-grained, a peer GPU's writes to them are not guaranteed visible to the kernels
        running here), shared through hipIpc handles; gq_hop_wait_copy moves a landed payload into the ordinary buffers the stage's
        kernels read (`self.h`, `self._tok4`)."""
        import ctypes
        from . import _lib
        L = _lib.lib()
        S, dev, c = self.n_seq, self.dev, self.model.config
        self.max_new_tokens = max_new_tokens
        self.tick = torch.zeros(S, dtype=torch.int32, device=dev)      # ticks of slot s completed on this stage
        self.tick64 = torch.zeros(S, dtype=torch.int64, device=dev)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.out_buf = torch.zeros(S, max_new_tokens + 1, dtype=torch.int32, device=dev)
        self.spins = int(os.environ.get("GQ_HOP_SPINS", str(1 << 21)))
        # one fine-grained block: [landing slots of h: S x dim fp16][landing slots of tokens: S x 16 B][seq_h: S words][seq_tok: S words]
        hb, tb = S * c.dim * 2, S * 16
        self._fg_layout = dict(h=0, tok=hb, seq_h=hb + tb, seq_tok=hb + tb + 4 * S)
        nbytes = hb + tb + 8 * S
        with torch.cuda.device(dev):
            p = ctypes.c_void_p()
            _lib.check(L.gq_hop_alloc(nbytes, ctypes.byref(p)), "gq_hop_alloc")
            self._fg = int(p.value)
            handle = (ctypes.c_ubyte * 64)()
            _lib.check(L.gq_hop_export(self._fg, handle), "gq_hop_export")
            # what a peer on ANOTHER device writes must be fine-grained memory (a coarse-grained landing slot would work between ranks that
            # share one GPU -- the tests -- and lose writes across xGMI): checked here, where the handle leaves the process
            if L.gq_hop_is_finegrained(self._fg) != 1:
                raise RuntimeError("pipeline hop: the landing block is not fine-grained device memory (gq_hop_alloc)")

        class _Raw:  # a torch view of raw device memory (CUDA array interface): reset() writes the words with tensor ops
            def __init__(r, ptr, n):
                r.__cuda_array_interface__ = dict(shape=(n, ), typestr="<i4", data=(ptr, False), version=2)
        self.seq_h = torch.as_tensor(_Raw(self._fg + self._fg_layout["seq_h"], S), device=dev)      # incoming hidden state of slot s carries tick seq_h[s]
        self.seq_tok = torch.as_tensor(_Raw(self._fg + self._fg_layout["seq_tok"], S), device=dev)  # (stage 0) incoming token ...
        self.seq_tok.fill_(1)                                                                        # 1 = the BOS token of tick 1
        torch.cuda.synchronize()
        allh = [None] * self.world
        dist.all_gather_object(allh, dict(handle=bytes(handle), pid=os.getpid(), dev=torch.cuda.get_device_properties(dev).name), group=self.group)
        self._peer_map = None
        peer = self.rank + 1 if not self.last else 0
        if allh[peer]["pid"] == os.getpid():
            base = self._fg  # (a one-stage pipeline sends to itself)
        else:
            with torch.cuda.device(dev):
                q = ctypes.c_void_p()
                buf = (ctypes.c_ubyte * 64).from_buffer_copy(allh[peer]["handle"])
                _lib.check(L.gq_hop_import(buf, ctypes.byref(q)), "gq_hop_import")
            base = self._peer_map = int(q.value)
        self._peer = dict(h=base + self._fg_layout["h"], tok=base + self._fg_layout["tok"], seq_h=base + self._fg_layout["seq_h"],
                          seq_tok=base + self._fg_layout["seq_tok"])
        dist.barrier(group=self.group)  # every mapping is open before anybody may free or reuse

    def close_ipc(self):
        """unmap the peer's slots and free this stage's (after a barrier of the caller's: nobody may still be sending)"""
        from . import _lib
        if getattr(self, "_peer_map", None):
            _lib.lib().gq_hop_close(self._peer_map)
            self._peer_map = None
        if getattr(self, "_fg", None):
            self.seq_h = self.seq_tok = None
            _lib.lib().gq_hop_free(self._fg)
            self._fg = None

    def _tick_ipc(self, slot: int):
        from . import _lib
        L, st = _lib.lib(), _lib.current_stream_ptr()
        tick = self.tick[slot:slot + 1]
        c = self.model.config
        lay, hb = self._fg_layout, c.dim * 2
        if self.first:  # the token of the tick lands in the token slot; copied into the row the embedding lookup reads
            _lib.check(L.gq_hop_wait_copy(self._fg + lay["seq_tok"] + 4 * slot, tick.data_ptr(), 1, self.err.data_ptr(), self.spins,
                                          self._fg + lay["tok"] + 16 * slot, self._tok4[slot].data_ptr(), 16, st), "gq_hop_wait")
        else:
            _lib.check(L.gq_hop_wait_copy(self._fg + lay["seq_h"] + 4 * slot, tick.data_ptr(), 1, self.err.data_ptr(), self.spins,
                                          self._fg + lay["h"] + hb * slot, self.h[slot].data_ptr(), hb, st), "gq_hop_wait")
        self._tick_native(slot)
        if not self.last:
            _lib.check(L.gq_hop_send(self.h[slot].data_ptr(), self._peer["h"] + hb * slot, hb, self._peer["seq_h"] + 4 * slot,
                                     tick.data_ptr(), 1, st), "gq_hop_send")
        else:
            self.out_buf[slot].index_copy_(0, self.tick64[slot:slot + 1], self.tok_out[slot:slot + 1])
            _lib.check(L.gq_hop_send(self._tok_out4[slot].data_ptr(), self._peer["tok"] + 16 * slot, 16, self._peer["seq_tok"] + 4 * slot,
                                     tick.data_ptr(), 2, st), "gq_hop_send")
        tick.add_(1)
        self.tick64[slot:slot + 1].add_(1)

    def _run_ipc(self, n_tokens: int) -> torch.Tensor:
        if n_tokens > self.max_new_tokens:
            raise ValueError(f"run({n_tokens}): the pipeline was built for max_new_tokens = {self.max_new_tokens}")
        self.reset()  # ticks, sequence words and the output index start over: a run never continues the previous one's counters
        torch.cuda.synchronize()
        dist.barrier(group=self.group)  # every stage has reset its sequence words
        for _ in range(n_tokens):
            for slot in range(self.n_seq):
                self.graphs[slot].replay()
        torch.cuda.synchronize()
        if int(self.err.item()):
            raise RuntimeError("device-to-device hop: a stage waited for its input beyond the spin limit (GQ_HOP_SPINS)")
        out = self.out_buf[:, :n_tokens].contiguous()
        dist.broadcast(out, src=self.world - 1, group=self.group)
        return out

    # -- stage compute -------------------------------------------------------------------------------------------
    def _tick_native(self, slot: int):
        """everything stage `rank` does for one token of sequence `slot`, on the current stream, no host work besides the
        launches (captured once per slot)"""
        m = self.model
        h = self.h[slot]
        pos = self.pos_dev[slot:slot + 1]
        x = h if self._x is None else self._x
        if self.first:
            m.native_embed(self.tok[slot:slot + 1], x)
        elif self._x is not None:
            x.copy_(h)
        m.native_layers(x, pos, self.layers.start, self.layers.stop, slot)
        pos.add_(1)
        if self.last:
            self._sample_native(m.native_head(x), slot)
        elif self._x is not None:
            h.copy_(x)

    def _sample_native(self, logits, slot):
        from . import _lib
        if self.top_k is not None and self.top_k <= 32:
            _lib.check(_lib.lib().gq_sample_topk(logits.data_ptr(), self.model.config.vocab_size, int(self.top_k), float(self.temperature),
                                                int(self.seed), self.rng_counter.data_ptr(), self.work_val.data_ptr(),
                                                self.work_idx.data_ptr(), None, None, self.tok_out[slot:slot + 1].data_ptr(),
                                                _lib.current_stream_ptr()), "gq_sample_topk")
        else:
            from .generate import sample
            idx, _ = sample(logits.view(1, 1, -1), temperature=self.temperature, top_k=self.top_k)
            self.tok_out[slot:slot + 1].copy_(idx.view(1))

    def _tick_eager(self, slot: int):
        """the same stage tick through the module-by-module forward (any linear class, CPU): reference semantics"""
        from .generate import sample
        m = self.model
        if self.first:
            self.h[slot].copy_(m.tok_embeddings(self.tok[slot:slot + 1].long()).view(-1))
        x = self.h[slot].view(1, 1, -1)
        ip = torch.tensor([self.pos[slot]], dtype=torch.long, device=self.dev)
        mask = m.causal_mask[None, None, ip]
        for li in self.layers:
            blk = m.layers[li]
            kv = blk.attention.kv_cache
            full_k, full_v = kv.k_cache, kv.v_cache
            kv.k_cache, kv.v_cache = full_k[slot:slot + 1], full_v[slot:slot + 1]  # views: this sequence's cache
            try:
                x = blk(x, ip, mask, m.rope_cos, m.rope_sin)
            finally:
                kv.k_cache, kv.v_cache = full_k, full_v
        self.h[slot].copy_(x.reshape(-1))
        self.pos[slot] += 1
        if self.last:
            logits = m.output(m.norm(self.h[slot].view(1, 1, -1)))
            idx, _ = sample(logits.view(1, 1, -1), temperature=self.temperature, top_k=self.top_k)
            self.tok_out[slot:slot + 1].copy_(idx.view(1).to(torch.int32))

    def _capture(self):
        """one hipGraph per slot (the slot selects the KV caches, buffers and position word the launches point at)"""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for slot in range(self.n_seq):  # warm-up: lazy kernel attributes, allocator (without the hops: no peer is ticking yet)
                self._tick_native(slot)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graphs = []
        for slot in range(self.n_seq):
            g = torch.cuda.CUDAGraph()
            with capture(g), torch.no_grad():
                if self.hop == "ipc":
                    self._tick_ipc(slot)
                else:
                    self._tick_native(slot)
            self.graphs.append(g)
        torch.cuda.synchronize()
        self.reset()

    def reset(self):
        """start new sequences: positions to 0 (the caches are overwritten as the positions advance)"""
        self.pos_dev.zero_()
        self.pos = [0] * self.n_seq
        if self.native and self.last:
            self.rng_counter.zero_()
        if self.hop == "ipc":
            self.tick.zero_()
            self.tick64.zero_()
            self.seq_h.zero_()
            self.seq_tok.fill_(1)
            self._tok4.zero_()
            self._tok4[:, 0] = self.bos_id
            if self.first:  # the landing slot of the token of tick 1 (its word already says "arrived"): BOS
                class _Raw:
                    def __init__(r, ptr, n):
                        r.__cuda_array_interface__ = dict(shape=(n, ), typestr="<i4", data=(ptr, False), version=2)
                land = torch.as_tensor(_Raw(self._fg + self._fg_layout["tok"], 4 * self.n_seq), device=self.dev).view(self.n_seq, 4)
                land.zero_()
                land[:, 0] = self.bos_id

    def _tick(self, slot: int):
        if self.graphs is not None:
            self.graphs[slot].replay()
        elif self.native:
            self._tick_native(slot)
            self.pos[slot] += 1
        else:
            self._tick_eager(slot)

    # -- schedule ------------------------------------------------------------------------------------------------
    def run(self, n_tokens: int) -> torch.Tensor:
        """returns int32 [n_seq, n_tokens] (valid on every rank: broadcast from the last stage at the end)"""
        if self.hop == "ipc":
            return self._run_ipc(n_tokens)
        w, r, S = self.world, self.rank, self.n_seq
        out = torch.zeros(S, n_tokens, dtype=torch.int32, device=self.dev)
        recv_req = [None] * S  # the posted receive of each slot's next input
        send_req = [None] * S  # the last send out of each slot's buffer

        class _StagedRecv:  # irecv into the pinned host row, copy to the device buffer on wait
            def __init__(s_, req, host, devt):
                s_.req, s_.host, s_.devt = req, host, devt

            def wait(s_):
                s_.req.wait()
                s_.devt.copy_(s_.host, non_blocking=True)

        def irecv(devt, host, src, group):
            if not self.staged:
                return dist.irecv(devt, src=src, group=group)
            return _StagedRecv(dist.irecv(host, src=src, group=group), host, devt)

        def isend(devt, host, dst, group):
            if not self.staged:
                return dist.isend(devt, dst=dst, group=group)
            host.copy_(devt, non_blocking=True)
            torch.cuda.current_stream().synchronize()  # the device-to-host copy has landed before the host send reads it
            return dist.isend(host, dst=dst, group=group)

        tok_stage = self.tok_host.clone().pin_memory() if self.staged else None  # send side of the token feedback

        def post_recv(step, slot):
            """post the receive of tick (step, slot)'s input, if it comes from another rank"""
            if step >= n_tokens or recv_req[slot] is not None:
                return
            if self.first:
                if step > 0 and w > 1:
                    recv_req[slot] = irecv(self.tok[slot:slot + 1], self.tok_host[slot:slot + 1] if self.staged else None, w - 1, self.fb_group)
            else:
                if send_req[slot] is not None:  # the buffer's previous content must have left before it is overwritten
                    send_req[slot].wait()
                    send_req[slot] = None
                recv_req[slot] = irecv(self.h[slot], self.h_host[slot] if self.staged else None, r - 1, self.group)

        def nxt(step, slot):
            return (step, slot + 1) if slot + 1 < S else (step + 1, 0)

        post_recv(0, 0)
        for step in range(n_tokens):
            for slot in range(S):
                # the next tick's input is received while this one computes -- unless it is this slot's own next step (one
                # sequence in flight: n_seq = 1): its buffer is in use until this tick has been computed and sent on
                ahead = nxt(step, slot)
                if ahead[1] != slot:
                    post_recv(*ahead)
                if self.first and step == 0:
                    self.tok[slot:slot + 1].fill_(self.bos_id)
                elif self.first and w == 1:
                    self.tok[slot:slot + 1].copy_(self.tok_out[slot:slot + 1])
                if recv_req[slot] is not None:
                    recv_req[slot].wait()
                    recv_req[slot] = None
                if self.first and send_req[slot] is not None:  # stage 0 computes in place in the buffer it sent from
                    send_req[slot].wait()
                    send_req[slot] = None
                self._tick(slot)
                if not self.last:
                    send_req[slot] = isend(self.h[slot], self.h_host[slot] if self.staged else None, r + 1, self.group)
                else:
                    out[slot, step:step + 1].copy_(self.tok_out[slot:slot + 1])
                    if w > 1 and step + 1 < n_tokens:
                        if send_req[slot] is not None:
                            send_req[slot].wait()
                        # the token leaves from the output row (stable storage: tok_out[slot] is rewritten next step)
                        send_req[slot] = isend(out[slot, step:step + 1], tok_stage[slot:slot + 1] if self.staged else None, 0, self.fb_group)
                if ahead[1] == slot:
                    post_recv(*ahead)
        for req in send_req:
            if req is not None:
                req.wait()
        if w > 1:
            dist.broadcast(out, src=w - 1, group=self.group)
        return out
