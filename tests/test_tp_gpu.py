"""GPU: tensor-parallel (row-split) decode, guidedquant_amd/tp.py -- W processes sharing the one MI355X of a gpurun box, every GEMV
output all-gathered through the device-to-device hop primitives (fine-grained landing buffers, hipIpc): the tokens of every rank
equal the single-process decode (a row's arithmetic does not depend on which rank owns it: anyprec.cu:387), also on a second run."""
import os
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import ROOT  # noqa: E402
from test_pipeline_nccl_gpu import _free_port  # noqa: E402


def _model(dev, kv):
    from guidedquant_amd.APLinear import APLinear
    from guidedquant_amd.generate import random_init_
    from guidedquant_amd.model import ModelArgs, Transformer
    cfg = ModelArgs(block_size=256, vocab_size=1024, n_layer=3, n_head=8, dim=512, intermediate_size=1024, n_local_heads=kv,
                    rope_base=500000, model_name="llama-test")
    m = Transformer(torch.float16, cfg, linear_class=APLinear, linear_kwargs=dict(bitwidth=2, device=dev)).to(device=dev, dtype=torch.float16)
    random_init_(m, seed=11, lut_std=0.05)
    return m.eval()


def _worker(rank, world, port, ntok, q, kv):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from guidedquant_amd.tp import TensorParallelDecoder
        dec = TensorParallelDecoder(_model(dev, kv), dist.group.WORLD, rank, world, max_new_tokens=ntok, temperature=0.0, top_k=32, bos_id=1)
        with torch.no_grad():
            out = dec.run(ntok)
            out2 = dec.run(ntok)
        torch.cuda.synchronize()
        q.put((rank, out.tolist(), out2.tolist()))
        dist.barrier()
        dec.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,kv", [(2, 2), (4, 4)])
def test_row_split_decode_ranks_sharing_one_gpu(world, kv):
    import torch.multiprocessing as mp
    from guidedquant_amd.generate import generate
    ntok = 12
    d0 = torch.device("cuda", 0)
    ref = generate(_model(d0, kv), torch.tensor([1], dtype=torch.int32, device=d0), ntok, use_graph=False, temperature=0.0, top_k=32)[0, 1:].tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ntok, q, kv)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=600) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert sorted(g[0] for g in got) == list(range(world))
    for _, out, out2 in got:
        assert out == ref and out2 == ref
