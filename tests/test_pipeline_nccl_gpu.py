"""GPU, world_size 2, nccl (= RCCL): the layer-pipelined decode on the native path (one hipGraph per stage tick, posted
receives, device-side positions) produces exactly the tokens of the single-GPU graph decode.  Skips below 2 GPUs (the
gpurun boxes have one); the driver's multi-GPU node runs it."""
import os
import socket
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import ROOT  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev):
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.generate import random_init_
    from guidedquant_amd.model import ModelArgs, Transformer
    cfg = ModelArgs(block_size=256, vocab_size=1024, n_layer=4, n_head=8, dim=512, intermediate_size=1024, n_local_heads=2,
                    rope_base=500000, model_name="llama-test")
    m = Transformer(torch.float16, cfg, linear_class=APLinear, linear_kwargs=dict(bitwidth=2, device=dev)).to(device=dev, dtype=torch.float16)
    random_init_(m, seed=7, lut_std=0.05)
    return m.eval()


def _worker(rank, world, port, ntok, q, backend="nccl", one_gpu=False, hop="p2p", n_seq=None):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0 if one_gpu else rank)
    dev = torch.device("cuda", 0 if one_gpu else rank)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from guidedquant_amd.pipeline import PipelinedDecoder, stage_ranges
        model = _model(dev)
        rng = stage_ranges(model.config.n_layer, world, head_cost_layers=1.0)[rank]
        dec = PipelinedDecoder(model, rank, world, rng, n_seq=n_seq or world, max_new_tokens=ntok, temperature=0.0, top_k=32, bos_id=1, hop=hop)
        assert dec.native and dec.graphs is not None and dec.staged == (backend == "gloo" and hop == "p2p")
        with torch.no_grad():
            out = dec.run(ntok)
            dec.reset()
            out2 = dec.run(ntok)  # a second run on the same graphs / caches
        torch.cuda.synchronize()
        if rank == 0:
            q.put((out.tolist(), out2.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_pipelined_native_decode_two_gpus_equals_single_gpu():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    from guidedquant_amd.generate import generate
    ntok = 12
    d0 = torch.device("cuda", 0)
    ref = generate(_model(d0), torch.tensor([1], dtype=torch.int32, device=d0), ntok, use_graph=False, temperature=0.0, top_k=32)[0, 1:].tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ntok, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got, got2 = q.get(timeout=300)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert got[0] == ref and got[1] == ref and got2 == got


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_native_decode_ranks_sharing_one_gpu(world):
    """the multi-rank NATIVE path on the one GPU a gpurun box has: `world` processes on cuda:0, gloo process group (hops staged
    through pinned host buffers), one hipGraph per (stage, slot), device-side positions, posted receives, separate feedback
    communicator -- tokens of every sequence equal the single-process decode, also on a second run over the same graphs"""
    import torch.multiprocessing as mp
    from guidedquant_amd.generate import generate
    ntok = 12
    d0 = torch.device("cuda", 0)
    ref = generate(_model(d0), torch.tensor([1], dtype=torch.int32, device=d0), ntok, use_graph=False, temperature=0.0, top_k=32)[0, 1:].tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ntok, q, "gloo", True)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got, got2 = q.get(timeout=400)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(row == ref for row in got) and got2 == got


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,n_seq", [(2, 2), (3, 3), (2, 1)])
def test_pipelined_decode_device_to_device_hops_ranks_sharing_one_gpu(world, n_seq):
    """hop="ipc": the stages hand the hidden state and the sampled token over through IPC-mapped device memory (gq_hop_send /
    gq_hop_wait at the ends of every tick graph; no send / receive call, no host synchronisation inside a run) -- `world`
    processes on cuda:0, tokens of every sequence equal the single-process decode, also on a second run, also with ONE sequence
    in flight (the reference's use case)"""
    import torch.multiprocessing as mp
    from guidedquant_amd.generate import generate
    ntok = 12
    d0 = torch.device("cuda", 0)
    ref = generate(_model(d0), torch.tensor([1], dtype=torch.int32, device=d0), ntok, use_graph=False, temperature=0.0, top_k=32)[0, 1:].tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ntok, q, "gloo", True, "ipc", n_seq)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got, got2 = q.get(timeout=600)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert len(got) == n_seq and all(row == ref for row in got) and got2 == got
