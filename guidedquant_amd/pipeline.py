"""Layer-pipelined bs=1 decode across ranks (the reference's precedent is the sequential layer
sharding of qtip/lib/utils/shard_model.py:30-68 and accelerate's device_map="auto",
any_precision/modules/AnyPrecisionForCausalLM.py:76-82 -- both single-process `tensor.to(device)` hops).

MI355X-native form: one process per GPU, stage g owns layers [l0, l1) (embedding on stage 0, final norm + lm_head on the
last stage) and the KV caches of those layers; the only data-path exchange is the point-to-point hop of the fp16
hidden state [dim] (16 KiB for dim = 8192) to the next stage and of the sampled token id back to stage 0, issued with
torch.distributed send/recv (backend "nccl" = RCCL over xGMI; "gloo" on CPU for the tests).  No collective is needed.

A single bs=1 stream cannot go faster than on one GPU (strict layer dependency); with S = world_size independent
sequences in flight every stage is busy every tick and the aggregate rate approaches S x the per-stage rate.
`PipelinedDecoder.run(n_tokens)` decodes n_tokens for each of the S sequences and returns them (on every rank).
"""
from typing import List, Optional

import torch
import torch.distributed as dist


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


class PipelinedDecoder:

    def __init__(self, model, rank: int, world: int, layers: range, n_seq: Optional[int] = None, max_new_tokens: int = 100,
                 temperature: float = 0.0, top_k: Optional[int] = 32, bos_id: int = 1, native: Optional[bool] = None, group=None):
        self.model, self.rank, self.world, self.layers = model, rank, world, layers
        self.n_seq = n_seq or world
        self.group = group
        self.first, self.last = rank == 0, rank == world - 1
        self.temperature, self.top_k, self.bos_id = temperature, top_k, bos_id
        dev = model.output.weight.device
        self.dev = dev
        model.setup_caches(self.n_seq, 1 + max_new_tokens)
        self.native = model.native_ready() if native is None else native
        c = model.config
        self.hidden = torch.zeros(c.dim, dtype=model.output.weight.dtype, device=dev)
        self.tok = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pos = [0] * self.n_seq
        self.pos_t = torch.zeros(1, dtype=torch.int32, device=dev)

    # -- stage compute -------------------------------------------------------------------------------------------
    def _embed(self):
        if self.native:
            self.model.native_embed(self.tok, self.hidden)
        else:
            self.hidden.copy_(self.model.tok_embeddings(self.tok.long()).view(-1))

    def _layers(self, slot: int):
        m = self.model
        self.pos_t.fill_(self.pos[slot])
        if self.native:
            m.native_layers(self.hidden, self.pos_t, self.layers.start, self.layers.stop, slot)
            return
        x = self.hidden.view(1, 1, -1)
        ip = self.pos_t.long()
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
        self.hidden.copy_(x.reshape(-1))

    def _head_and_sample(self) -> torch.Tensor:
        from .generate import sample
        m = self.model
        if self.native:
            logits = m.native_head(self.hidden)
        else:
            logits = m.output(m.norm(self.hidden.view(1, 1, -1)))
        idx, _ = sample(logits.view(1, 1, -1), temperature=self.temperature, top_k=self.top_k)
        return idx.view(1).to(torch.int32)

    # -- schedule ------------------------------------------------------------------------------------------------
    def run(self, n_tokens: int) -> torch.Tensor:
        """returns int32 [n_seq, n_tokens] (valid on every rank: broadcast from the last stage at the end)"""
        w, r, S = self.world, self.rank, self.n_seq
        out = torch.zeros(S, n_tokens, dtype=torch.int32, device=self.dev)
        pending = []  # outstanding isend requests (with their buffers): sends never block the stage loop

        def post(t, dst):
            buf = t.clone()
            pending.append((dist.isend(buf, dst=dst, group=self.group), buf))
            while len(pending) > 4 * S:
                pending.pop(0)[0].wait()

        for step in range(n_tokens):
            for slot in range(S):
                if self.first:
                    if step == 0:
                        self.tok.fill_(self.bos_id)
                    elif w > 1:
                        dist.recv(self.tok, src=w - 1, group=self.group)
                    else:
                        self.tok.copy_(out[slot, step - 1:step])
                    self._embed()
                else:
                    dist.recv(self.hidden, src=r - 1, group=self.group)
                self._layers(slot)
                self.pos[slot] += 1
                if not self.last:
                    post(self.hidden, r + 1)
                else:
                    t = self._head_and_sample()
                    out[slot, step] = t[0]
                    if w > 1 and step + 1 < n_tokens:
                        post(t, 0)
        for req, _ in pending:
            req.wait()
        if w > 1:
            dist.broadcast(out, src=w - 1, group=self.group)
        return out
