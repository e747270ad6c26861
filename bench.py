#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its config: bs=1 decode tokens/sec of Llama-3.1-8B-Instruct, 2-bit
Any-Precision weights, fused QKV / Up-Gate, on MI355X -- plus the AP-GEMV roofline figure and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one decoded token (bs = 1): embedding, 32 x (RMSNorm->wqkv GEMV, RoPE+KV+attention, wo GEMV+residual,
RMSNorm->w1w3 GEMV, SiLU*up->w2 GEMV+residual), final norm + fp16 lm_head GEMV, top-k sampling -- the reference's
`decode_one_token` (inference/generate.py:82-86), replayed as one hipGraph.  Like the reference's harness the
tokens are decoded from a BOS-only prompt in sequences of 100 new tokens (generate.py:395-401), so the KV length
seen by the attention kernel cycles through 1..100.  Weights are synthetic (`--random_init` of the reference):
there is no network for checkpoints; every quantized tensor has the real shape, format and size, all resident in
HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU): the decode path of one sequence does not shard
without changing the reference's data path, so ranks run independent replicas (one sequence per GPU, no data-path
collective) and the job value is the sum -- "replicas only", weak scaling.

Output: ONE JSON line on rank 0 (schema in the task contract) with `roofline` (dominant quantized kernel, the
w1w3 AP-GEMV, algorithmic bytes B_ap / average launch duration measured with HIP events on the launch stream over
all 32 layers' distinct weights) and `cpu_baseline` (oracle port timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = "meta-llama/Meta-Llama-3.1-8B-Instruct"
SEQ_NEW_TOKENS = 100  # reference default max_new_tokens (generate.py:398)
HBM_PEAK_GBPS = 8000.0


def b_ap(bits, N, K):
    """algorithmic bytes of one AP-GEMV launch: planes + LUT + x + y"""
    return bits * N * K // 8 + 2 * N * (1 << bits) + 2 * K + 2 * N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--mode", choices=["default", "exact", "fast"], default="default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallel", choices=["replicas", "pp"], default="replicas",
                    help="N>1: independent replicas (default) or the layer pipeline with point-to-point hops (pp)")
    ap.add_argument("--model", default=MODEL, help="model name from guidedquant_amd.model.transformer_configs (e.g. "
                    "meta-llama/Llama-3.3-70B-Instruct with --parallel pp on 8 GPUs); the headline metric is quoted on the default")
    ap.add_argument("--torch-sampling", action="store_true", help="sample with the reference's torch ops instead of the fused HIP sampler")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from guidedquant_amd import _lib
    from guidedquant_amd.generate import DecodeGraph, load_model, _get_model_size

    L = _lib.lib()
    if args.mode != "default":
        _lib.check(L.gq_set_ap_mode(1 if args.mode == "exact" else 0), "gq_set_ap_mode")

    torch.manual_seed(1234)
    model = load_model(args.model, dev, "ap", args.bits, random_init=True)
    cfg = model.config
    model.setup_caches(1, SEQ_NEW_TOKENS + 1)
    assert model.native_ready()
    graph = DecodeGraph(model, dev, native_sampling=not args.torch_sampling, temperature=0.0, top_k=32)
    bos = torch.tensor([[128000 % cfg.vocab_size]], dtype=torch.int32, device=dev)
    zero = torch.zeros((1, ), dtype=torch.int32, device=dev)

    def run_steps(n):
        done = 0
        while done < n:
            graph.tok.copy_(bos)
            graph.pos.copy_(zero)
            for _ in range(min(SEQ_NEW_TOKENS, n - done)):
                graph.step()
            done += min(SEQ_NEW_TOKENS, n - done)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pp = world > 1 and args.parallel == "pp"
    if pp:
        # layer pipeline: stage r owns a contiguous layer range, `world` sequences in flight, hidden state hops r -> r+1
        # and token ids hop back to stage 0 over RCCL send/recv; a "step" is still one decoded token (summed over
        # the sequences), so the job decodes args.steps tokens per rank-equivalent = world * args.steps in total
        from guidedquant_amd.pipeline import PipelinedDecoder, stage_ranges
        rng = stage_ranges(cfg.n_layer, world, head_cost_layers=6.0)[rank]
        dec = PipelinedDecoder(model, rank, world, rng, n_seq=world, max_new_tokens=max(args.steps, args.warmup),
                               temperature=0.0, top_k=32, bos_id=128000 % cfg.vocab_size)

        def run_steps(n):  # noqa: F811
            dec.pos = [0] * dec.n_seq
            with torch.no_grad():
                dec.run(n)

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    tok_s = world * args.steps / dt

    # ------------------------------------------------------------------ roofline of the dominant quantized kernel
    s = torch.cuda.current_stream()
    I, D = cfg.intermediate_size, cfg.dim
    x = torch.randn(D, device=dev).half()
    gu = torch.empty(2 * I, dtype=torch.float16, device=dev)
    nw = model.layers[0].post_attention_layernorm.weight

    def w1w3_pass():
        for blk in model.layers:
            m = blk.feed_forward.w1w3
            rc = L.gq_anyprec_gemv_fused(x.data_ptr(), gu.data_ptr(), m.qweight.data_ptr(), m.lut.data_ptr(), 2 * I, D,
                                         m.bitwidth, nw.data_ptr(), cfg.norm_eps, None, 0, _lib.current_stream_ptr())
            assert rc == 0, L.gq_last_error()

    w1w3_pass()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record(s)
    for _ in range(reps):
        w1w3_pass()
    e1.record(s)
    e1.synchronize()
    t_kernel_us = e0.elapsed_time(e1) * 1e3 / (reps * cfg.n_layer)
    bytes_launch = b_ap(args.bits, 2 * I, D)
    achieved = bytes_launch / t_kernel_us / 1e3  # GB/s
    # HBM bytes per launch from the PMC counters: measured offline (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # passes, tools/prof_bench.sh) for exactly this kernel and shape, committed under profiles/ with its correction
    traffic = None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_w1w3_traffic.json")
    if args.bits == 2 and cfg.dim == 4096 and I == 14336 and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "kernel": "AP-GEMV w1w3 %dx%d" % (2 * I, D),
                "avg_launch_us": round(t_kernel_us, 3), "algorithmic_bytes_per_launch": bytes_launch}

    # ------------------------------------------------------------------ CPU baseline (rank 0 only, bounded sample)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_sample(cfg, args.bits)

    model_size, _ = _get_model_size(model)
    mode = {"default": "exact" if os.environ.get("GQ_AP_EXACT", "0") != "0" else "default"}.get(args.mode, args.mode)
    if rank == 0:
        line = {
            "metric": "decode tokens/sec (bs=1)", "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "%s %d-bit Any-Precision (LNQ format), fused QKV/UpGate, bs=1 decode, "
                                   "BOS prompt, 100 new tokens per sequence, top_k=32, temperature=0" % (cfg.model_name, args.bits),
                       "bits": args.bits, "parallelism": ("pp%d (layer pipeline, p2p hops, %d sequences in flight)" % (world, world)) if pp else ("replicas" if world > 1 else "single"), "ap_mode": mode, "sampling": "torch ops" if args.torch_sampling else "fused HIP top-k sampler",
                       "model_bytes": model_size, "model_bandwidth_GBps": round(model_size * tok_s / world / 1e9, 1)},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_sample(cfg, bits):
    """Reference-order AP-GEMV port (oracle/, C + OpenMP) on the host cores: one transformer layer's four GEMVs at
    full size, extrapolated to tokens/s = 1 / (n_layer * t_layer + t_lm_head).  The reference itself has no CPU
    kernel for this path."""
    import numpy as np
    import torch
    from oracle import oracle
    from guidedquant_amd import pack
    oracle.build()
    cores = os.cpu_count() or 1
    oracle.set_threads(cores)
    D, I = cfg.dim, cfg.intermediate_size
    kvd = (cfg.n_head + 2 * cfg.n_local_heads) * cfg.head_dim
    shapes = [(kvd, D), (D, D), (2 * I, D), (D, I)]
    rng = np.random.default_rng(0)
    t_layer = 0.0
    for N, K in shapes:
        q = pack.random_planes(N, K, bits, seed=N + K)
        lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
        xv = rng.normal(0, 1, K).astype(np.float16)
        t0 = time.perf_counter()
        oracle.ap_gemv_f16(xv, q, lut, bits)
        t_layer += time.perf_counter() - t0
    # dense lm_head on the CPU: fp32 matvec of a 1/8 row sample, scaled
    rows = cfg.vocab_size // 8
    W = torch.randn(rows, D, dtype=torch.float32)
    xv = torch.randn(D, dtype=torch.float32)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    for _ in range(3):
        W @ xv
    t_lm = (time.perf_counter() - t0) / 3 * 8
    tok_s = 1.0 / (cfg.n_layer * t_layer + t_lm)
    return {"value": round(tok_s, 4), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"one full-size layer (4 AP-GEMVs, fp16-order oracle, {cores} OpenMP threads: {t_layer:.3f} s) x {cfg.n_layer} "
                      f"+ fp32 lm_head matvec ({t_lm * 1e3:.1f} ms, measured on 1/8 of the rows)"}


if __name__ == "__main__":
    main()
