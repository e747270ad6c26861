#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its config: bs=1 decode tokens/sec of Llama-3.1-8B-Instruct, 2-bit
Any-Precision weights, fused QKV / Up-Gate, on MI355X -- plus the AP-GEMV roofline figures and CPU baselines.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (driver contract)
    python bench.py --backend qtip                               (BASELINE configs[3]: Llama-2-7b QTIP 2-bit)
    python bench.py --bits 3 | --bits 4                          (configs[2])
    python bench.py --model meta-llama/Llama-3.3-70B-Instruct    (configs[4] on one GPU; N > 1 adds the pipeline record)

A "step" is one decoded token (bs = 1): embedding, n_layer x (RMSNorm->wqkv GEMV, RoPE+KV+attention, wo GEMV+residual,
RMSNorm->w1w3 GEMV+SiLU*up, w2 GEMV+residual), final norm + fp16 lm_head GEMV, top-k sampling -- the reference's
`decode_one_token` (inference/generate.py:82-86), replayed as one hipGraph.  Like the reference's harness the
tokens are decoded from a BOS-only prompt in sequences of 100 new tokens (generate.py:395-401), so the KV length
seen by the attention kernel cycles through 1..100.  Weights are synthetic (`--random_init` of the reference):
there is no network for checkpoints; every quantized tensor has the real shape, format and size, all resident in
HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU): the decode path of one sequence does not shard
without changing the reference's data path, so `value` comes from independent replicas (one sequence per GPU, no
data-path collective; "replicas only", weak scaling -- DESIGN.md section 6).  The north star's layer pipeline
(BASELINE configs[4]: Llama-3.3-70B 2-bit over the N GPUs, point-to-point hops, N sequences in flight) is measured in the
same run and reported as the `pipeline_70b` sub-record of the same JSON line, next to `tp_70b` (the row-split tensor-parallel decode of
ONE sequence of the same model); both run in child processes with a process group of their own.

Process model (round 6): this process measures the headline and the `roofline` object and nothing else; every other record of the
line (`cpu_baseline`, `roofline_by_shape`, exact mode, each of `other_configs`; for N > 1 `pipeline_70b` and `tp_70b`) is measured by a
child process of its own (`python bench.py --leg NAME`, own timeout) and merged into the one line -- a leg that crashes, aborts or hangs
becomes {"error": "rc -6"} in its place and costs nothing else.

Output: ONE JSON line on rank 0 (schema in the task contract) with
  `roofline`           dominant quantized kernel (the w1w3 AP-GEMV exactly as the decode graph launches it: RMSNorm prologue,
                       gate/up pair epilogue), algorithmic bytes B_ap / average launch duration, HIP events on the launch stream
                       over all layers' distinct weights; `traffic` = HBM bytes per launch from the committed PMC passes of
                       the SAME kernel template and mode (profiles/r03_w1w3_traffic.json), else null
  `roofline_by_shape`  the four Llama-3-8B GEMV shapes x 2/3/4 bits (BASELINE.json metric: "+3/4-bit sweep"), default dispatch
  `exact_mode_tok_s`   the same decode with every quantized GEMV in the bit-exact (reference fp16 order) mode
  `cpu_baseline`       oracle ports timed on the host cores on a bounded sample (N = 1 only)
  `other_configs`      BASELINE configs[2] (3- and 4-bit decode), configs[3] (Llama-2-7b QTIP 2-bit decode + the bare trellis matvec
                       at its three shapes) and configs[4] on one GPU (Llama-3.3-70B 2-bit), 200 steps each (N = 1 only)
  `distributed`        N > 1: backend, world size and NCCL (= RCCL) version the barrier / max-reduce went through
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# three OpenMP runtimes meet in the CPU legs (torch's libgomp, the oracle's libgomp, the product twins' libomp): idle worker
# threads must sleep, not spin, or each library's parallel region fights the previous one's spinning pool for the cores
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("KMP_BLOCKTIME", "0")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

MODEL = "meta-llama/Meta-Llama-3.1-8B-Instruct"
QTIP_MODEL = "meta-llama/Llama-2-7b"
PP_MODEL = "meta-llama/Llama-3.3-70B-Instruct"
SEQ_NEW_TOKENS = 100  # reference default max_new_tokens (generate.py:398)
HBM_PEAK_GBPS = 8000.0
SHAPES_8B = {"wqkv": (6144, 4096), "wo": (4096, 4096), "w1w3": (28672, 4096), "w2": (4096, 14336)}


def b_ap(bits, N, K):
    """algorithmic bytes of one AP-GEMV launch: planes + LUT + x + y"""
    return bits * N * K // 8 + 2 * N * (1 << bits) + 2 * K + 2 * N


def b_qtip(R, M, K):
    """algorithmic bytes of one QTIP matvec: trellis + codebook + x + y"""
    return R * M * K // 8 + 2048 + 2 * K + 4 * M


def graph_time_us(launch, n_distinct, iters=200, reps=5):
    """average launch duration of `launch(i)` (i rotating over n_distinct argument sets) inside one captured graph,
    HIP events on the launch stream, best of `reps` replays"""
    import torch
    from guidedquant_amd._graphs import capture, release
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(n_distinct):
            launch(i)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture(g, stream=s):
            for i in range(iters):
                launch(i % n_distinct)
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record(s)
            g.replay()
            e1.record(s)
            s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        release(g)
    return best


def stream_floor_us(buffers, nbytes, iters=200):
    """what a launch that only READS nbytes once reaches on this chip (gq_debug_stream_read over the given rotating device
    buffers, same timing method as the GEMV launches): the one-shot streaming floor, launch boundary included"""
    import torch
    from guidedquant_amd import _lib
    L = _lib.lib()
    sink = torch.zeros(16, dtype=torch.int32, device=buffers[0].device)
    nbytes = min(int(nbytes), min(b.numel() * b.element_size() for b in buffers)) // 16 * 16

    def run(i):
        assert L.gq_debug_stream_read(buffers[i].data_ptr(), nbytes, sink.data_ptr(), _lib.current_stream_ptr()) == 0, L.gq_last_error()
    return graph_time_us(run, len(buffers), iters)


def bench_ap_shape(name, N, K, bits, iters=200, min_ws=512 << 20, fused=None, floor=False):
    """One AP-GEMV shape, default dispatch, rotating over > 512 MB of distinct weights (the 256 MiB Infinity Cache cannot
    serve them).  fused = None: the plain entry point (gq_anyprec_gemv); "norm": RMSNorm prologue; "norm_pairs": RMSNorm
    prologue + gate/up pair epilogue (the decode graph's w1w3 launch); "resid": residual epilogue (wo / w2 launches)."""
    import torch
    from guidedquant_amd import _lib
    d = torch.device("cuda", torch.cuda.current_device())
    per = bits * N * K // 8
    nbuf = max(2, min(64, (min_ws + per - 1) // per))
    g = torch.Generator(device=d)
    g.manual_seed(1)
    qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g) for _ in range(nbuf)]
    luts = [(torch.randn(N, 1 << bits, device=d, generator=g) * 0.02).half().sort(dim=1).values.contiguous() for _ in range(nbuf)]
    x = torch.randn(1, 1, K, device=d, generator=g).half()
    nw = (1 + 0.1 * torch.randn(K, device=d, generator=g)).half()
    res = torch.randn(N, device=d, generator=g).half()
    out = torch.empty(1, 1, N, dtype=torch.float16, device=d)
    L = _lib.lib()
    ws = None
    if fused == "resid_ws":
        nb = int(L.gq_anyprec_gemv_fused_ws_bytes(N, K, bits, 1))
        ws = torch.zeros(nb // 4, dtype=torch.float32, device=d) if nb else None
    rope = None
    if fused in ("qkv_rope", "qkv_rope_ho"):
        rope = dict(pos=torch.tensor([17], dtype=torch.int32, device=d), cos=torch.randn(256, 128, device=d).half(),
                    sin=torch.randn(256, 128, device=d).half(), kc=torch.zeros(N // 128 // 6, 256, 128, dtype=torch.float16, device=d),
                    vc=torch.zeros(N // 128 // 6, 256, 128, dtype=torch.float16, device=d))

    # "_ho" forms: the statistics hand-over of the decode graph (include/gq_hip.h, round 5) -- RMSNorm prologues read the partial sums
    # of squares of x (written here once by gq_ssq_rows), residual epilogues write those of their outputs
    ho = fused is not None and fused.endswith("_ho")
    if ho:
        fused = fused[:-3]
    ssq = torch.zeros(_lib.SSQ_SLOTS, dtype=torch.float32, device=d)
    if ho:
        assert L.gq_ssq_rows(x.data_ptr(), K, ssq.data_ptr(), _lib.current_stream_ptr()) == 0, L.gq_last_error()
    ssq_in = ssq.data_ptr() if ho else None

    def run(i):
        sp = _lib.current_stream_ptr()
        if ho and fused in ("norm", "norm_pairs"):
            rc = L.gq_anyprec_gemv_fused_ho(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), luts[i].data_ptr(), N, K, bits, nw.data_ptr(),
                                            1e-5, None, 4 if fused == "norm_pairs" else 0, None, 0, ssq_in, None, sp)
        elif ho and fused == "qkv_rope":
            hd = 128
            hkv = N // hd // 6
            rc = L.gq_anyprec_gemv_qkv_rope_ho(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), luts[i].data_ptr(), N, K, bits, nw.data_ptr(), 1e-5,
                                               rope["pos"].data_ptr(), rope["cos"].data_ptr(), rope["sin"].data_ptr(), rope["kc"].data_ptr(),
                                               rope["vc"].data_ptr(), 4 * hkv, hkv, hd, 256, ssq_in, sp)
        elif ho and fused == "resid":
            rc = L.gq_anyprec_gemv_fused_ho(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), luts[i].data_ptr(), N, K, bits, None, 0.0,
                                            res.data_ptr(), 1, None, 0, None, ssq.data_ptr(), sp)
        elif fused is None:
            rc = L.gq_anyprec_gemv(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), luts[i].data_ptr(), 1, N, K, bits, 0, sp)
        elif fused in ("norm", "norm_pairs"):
            rc = L.gq_anyprec_gemv_fused(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), luts[i].data_ptr(), N, K, bits, nw.data_ptr(),
                                         1e-5, None, 4 if fused == "norm_pairs" else 0, sp)
        elif fused == "qkv_rope":  # RMSNorm prologue + RoPE / KV-cache epilogue (the decode graph's wqkv launch, 8B head geometry)
            hd = 128
            hkv = N // hd // 6
            rc = L.gq_anyprec_gemv_qkv_rope(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), luts[i].data_ptr(), N, K, bits, nw.data_ptr(), 1e-5,
                                            rope["pos"].data_ptr(), rope["cos"].data_ptr(), rope["sin"].data_ptr(), rope["kc"].data_ptr(),
                                            rope["vc"].data_ptr(), 4 * hkv, hkv, hd, 256, sp)
        elif fused == "resid_ws":  # residual epilogue with the workspace the model passes (K > 16384: K split over blocks)
            rc = L.gq_anyprec_gemv_fused_ws(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), luts[i].data_ptr(), N, K, bits, None, 0.0,
                                            res.data_ptr(), 1, ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0, sp)
        else:
            rc = L.gq_anyprec_gemv_fused(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), luts[i].data_ptr(), N, K, bits, None, 0.0,
                                         res.data_ptr(), 1, sp)
        assert rc == 0, L.gq_last_error()

    us = graph_time_us(run, nbuf, iters)
    gbs = b_ap(bits, N, K) / us / 1e3
    rec = {"shape": name, "N": N, "K": K, "bits": bits, "launch": (fused or "plain") + ("_ho" if ho else ""), "us": round(us, 3), "GBps": round(gbs, 1),
           "frac": round(gbs / HBM_PEAK_GBPS, 4)}
    if floor:  # the same bytes (the plane words: 99 % of the algorithmic bytes) read once by a launch that does nothing else
        fl = stream_floor_us(qs, per, iters)
        rec["stream_floor_us"] = round(fl, 3)
        rec["frac_of_stream_floor"] = round(fl / us, 4)
    return rec


def decode_tok_s(model, dev, steps, warmup, torch_sampling=False):
    """tokens/s of the captured decode step (BOS prompt, sequences of 100 new tokens), wall clock around `steps` replays"""
    import torch
    from guidedquant_amd.generate import DecodeGraph
    # (round 5: the embedding lookup of the NEXT token rides in the sampler's last block -- GQ_FOLD_EMBED=0 restores the launch)
    # round 5: ten consecutive token steps per graph replay (token / position / RNG counter feed back on the device): the boundary
    # between two graph launches, ~8 us, is paid once per ten tokens (profiles/r05_steps_per_replay.txt: 898 -> 906 tokens/s)
    spr = 1 if torch_sampling else int(os.environ.get("GQ_STEPS_PER_REPLAY", "10"))
    assert SEQ_NEW_TOKENS % spr == 0
    graph = DecodeGraph(model, dev, native_sampling=not torch_sampling, temperature=0.0, top_k=32,
                        fold_embed=os.environ.get("GQ_FOLD_EMBED", "1") != "0", steps_per_replay=spr)
    bos = torch.tensor([[(128000 if model.config.vocab_size > 100000 else 1)]], dtype=torch.int32, device=dev)
    zero = torch.zeros((1, ), dtype=torch.int32, device=dev)
    # (both captured graphs are replayed once here, outside every timed region: with --warmup smaller than `spr` the warm-up runs
    # through the single-step graph only and the multi-step graph's FIRST replay would fall into the timed steps -- 904 vs 924
    # tokens/s between driver-form runs of one build on one box)
    # The same replays also bring the GPU's clocks up before anything is timed: a fresh process times its first tokens 3 % low after the
    # seconds of host-side model construction (900-903 vs 919-929 tokens/s for 20-step runs of one build).  Setup, like the capture itself;
    # the W warm-up steps of the contract follow.
    for _ in range(20 if not torch_sampling else 2):
        graph.set_token(bos, zero)
        graph.step()
    graph.step_one()
    torch.cuda.synchronize()

    def run_steps(n):
        done = 0
        while done < n:
            graph.set_token(bos, zero)
            k = min(SEQ_NEW_TOKENS, n - done)
            for _ in range(k // spr):  # (a replay decodes `spr` tokens)
                graph.step()
            for _ in range(k % spr):   # (what is left of the K steps asked for: single-step replays -- exactly K token steps are timed)
                graph.step_one()
            done += k

    return graph, run_steps


LEG_MARK = "GQ_LEG_JSON "
# the legs of the N = 1 line next to the headline, in run order: (key in the line, group, --leg name, extra argv)
LEGS = [
    ("cpu_baseline", None, "cpu_baseline", []),
    ("roofline_by_shape", None, "roofline_by_shape", []),
    ("exact_mode", None, "exact_mode", []),
    ("ap_3bit", "other_configs", "decode", ["--bits", "3"]),
    ("ap_4bit", "other_configs", "decode", ["--bits", "4"]),
    ("qtip_llama2_7b_2bit", "other_configs", "decode", ["--backend", "qtip"]),
    ("llama33_70b_2bit_1gpu", "other_configs", "decode", ["--model", PP_MODEL]),
    ("long_context_8b_2bit", "other_configs", "long_context", []),
    ("hf_generate_8b_2bit", "other_configs", "hf_generate", []),
    ("prefill_gemm_w1w3_2bit", "other_configs", "prefill_gemm", []),
    ("prompt_pass_8b_2bit", "other_configs", "prompt_pass", []),
]


def run_leg_child(leg, extra, timeout_s, env_extra=None):
    """one side leg of the line in its OWN process (`python bench.py --leg NAME ...`): a crash, abort or hang of a leg costs that leg's
    record -- {"error": "rc -6"} -- and nothing else (round 5: a C++ terminate in the ninth leg of a one-process bench took the headline
    with it).  The child prints its record as the last stdout line behind LEG_MARK."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("PYTHONFAULTHANDLER", "1")
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", leg] + list(extra)
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %.0f s" % timeout_s}
    out = p.stdout.decode("utf-8", "replace")
    for ln in reversed(out.splitlines()):
        if ln.startswith(LEG_MARK):
            try:
                rec = json.loads(ln[len(LEG_MARK):])
            except ValueError as e:
                return {"error": "unreadable record: %s" % e}
            if isinstance(rec, dict):
                rec.setdefault("leg_wall_s", round(time.perf_counter() - t0, 1))
            return rec
    err = p.stderr.decode("utf-8", "replace").strip().splitlines()
    sys.stderr.write("[bench.py] leg %s %s: rc %d\n%s\n" % (leg, " ".join(extra), p.returncode, "\n".join(err[-15:])))
    return {"error": "rc %d" % p.returncode, "stderr_tail": " | ".join(e.strip() for e in err[-3:])[:400]}


def leg_main(args):
    """child side: run ONE leg on cuda:LOCAL_RANK and print its record"""
    import torch
    leg = args.leg
    if leg == "cpu_baseline":
        from guidedquant_amd.model import transformer_configs, ModelArgs  # noqa: F401
        cfg = ModelArgs.from_name(args.model or MODEL)
        rec = cpu_baseline_sample(cfg, args.bits)
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
        local_rank = 0 if os.environ.get("GQ_BENCH_ONE_GPU", "0") != "0" else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if leg == "roofline_by_shape":
            rec = roofline_by_shape_record()
        elif leg == "exact_mode":
            rec = exact_mode_record(dev, args.bits)
        elif leg == "decode":
            rec = decode_record(dev, args.model or (QTIP_MODEL if args.backend == "qtip" else MODEL), args.backend, args.bits)
        elif leg == "long_context":
            rec = long_context_record(dev)
        elif leg == "hf_generate":
            rec = hf_generate_record(dev)
        elif leg == "prefill_gemm":
            rec = prefill_records(dev)
        elif leg == "prompt_pass":
            rec = prompt_pass_records(dev)
        elif leg in ("pipeline_70b", "tp_70b"):
            rec = collective_leg(leg, dev)
        else:
            raise SystemExit("unknown --leg %r" % leg)
    print(LEG_MARK + json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--backend", choices=["ap", "qtip"], default="ap")
    ap.add_argument("--mode", choices=["default", "exact", "fast"], default="default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 3-/4-bit, QTIP and 70B single-GPU sub-records")
    ap.add_argument("--quick", action="store_true", help="headline number and roofline object only (no shape table / exact-mode / CPU legs)")
    ap.add_argument("--parallel", choices=["replicas", "pp"], default="replicas",
                    help="N>1: what `value` measures -- independent replicas (default) or the layer pipeline (pp) of --model")
    ap.add_argument("--no-pp-record", action="store_true", help="N>1: skip the pipeline_70b / tp_70b sub-records")
    ap.add_argument("--model", default=None, help="model name from guidedquant_amd.model.transformer_configs; the headline metric is quoted on the default")
    ap.add_argument("--torch-sampling", action="store_true", help="sample with the reference's torch ops instead of the fused HIP sampler")
    ap.add_argument("--leg", default=None, help="(internal) run one side leg of the line in this process and print its record")
    args = ap.parse_args()
    if args.leg:
        return leg_main(args)
    t_start = time.perf_counter()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
    # GQ_BENCH_ONE_GPU=1 (with GQ_BENCH_BACKEND=gloo): every rank on cuda:0 -- exercises the N > 1 code path (replicas line,
    # pipeline_70b / tp_70b records) on a one-GPU box; the numbers of such a run mean nothing
    one_gpu = os.environ.get("GQ_BENCH_ONE_GPU", "0") != "0"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("GQ_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from guidedquant_amd import _lib
    from guidedquant_amd.generate import _get_model_size, load_model

    L = _lib.lib()
    if args.mode != "default":
        _lib.check(L.gq_set_ap_mode(1 if args.mode == "exact" else 0), "gq_set_ap_mode")

    qtip = args.backend == "qtip"
    name = args.model or (QTIP_MODEL if qtip else MODEL)
    torch.manual_seed(1234)
    model = load_model(name, dev, "qtip" if qtip else "ap", args.bits, random_init=True)
    cfg = model.config
    model.setup_caches(1, SEQ_NEW_TOKENS + 1)
    assert model.native_ready(), "the fused HIP decode step does not serve this model"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pp = world > 1 and args.parallel == "pp"
    if pp:
        run_steps, dec = pipeline_runner(model, rank, world, max(args.steps, args.warmup))
        graph = None
    else:
        graph, run_steps = decode_tok_s(model, dev, args.steps, args.warmup, args.torch_sampling)

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

    # behind the timed region, same process, same graph: the reference's own definition of the metric (whole 100-token sequences,
    # generate.py:344-389) over 400 steps, and the same with ONE token step per graph replay
    tok_s_400 = tok_s_single = None
    if not pp and world == 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(400)
        torch.cuda.synchronize()
        tok_s_400 = round(400 / (time.perf_counter() - t1), 2)
        if graph is not None and graph.steps_per_replay > 1:
            bos = torch.tensor([[(128000 if cfg.vocab_size > 100000 else 1)]], dtype=torch.int32, device=dev)
            graph.set_token(bos, 0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(SEQ_NEW_TOKENS):
                graph.step_one()
            torch.cuda.synchronize()
            tok_s_single = round(SEQ_NEW_TOKENS / (time.perf_counter() - t1), 2)

    try:
        roofline = qtip_roofline(cfg, args.bits) if qtip else ap_roofline(model, args.bits, args.mode)
    except Exception as e:  # (the headline stands on its own)
        roofline = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    model_size, _ = _get_model_size(model)

    mode = {"default": "exact" if os.environ.get("GQ_AP_EXACT", "0") != "0" else "default"}.get(args.mode, args.mode)
    spr = 1 if (args.torch_sampling or pp) else int(os.environ.get("GQ_STEPS_PER_REPLAY", "10"))
    workload = ("%s QTIP %d-bit (trellis-coded, HYB code), unfused linears, bs=1 decode, BOS prompt, 100 new tokens per sequence, "
                "top_k=32, temperature=0" % (cfg.model_name, args.bits)) if qtip else \
        ("%s %d-bit Any-Precision (LNQ format), fused QKV/UpGate, bs=1 decode, BOS prompt, 100 new tokens per sequence, top_k=32, "
         "temperature=0" % (cfg.model_name, args.bits))
    line = {
        "metric": "decode tokens/sec (bs=1)", "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": workload, "bits": args.bits, "backend": args.backend,
                   "parallelism": ("pp%d (layer pipeline, p2p hops, %d sequences in flight)" % (world, world)) if pp else ("replicas" if world > 1 else "single"),
                   "ap_mode": mode, "sampling": "torch ops" if args.torch_sampling else "fused HIP top-k sampler",
                   "model_bytes": model_size, "model_bandwidth_GBps": round(model_size * tok_s / world / 1e9, 1),
                   "kv_positions_timed": "0..%d" % (min(args.steps, SEQ_NEW_TOKENS) - 1),
                   "token_steps_per_graph_replay": spr,
                   "untimed_setup_token_steps": (20 * spr + 1) if not (args.torch_sampling or pp) else 2,
                   "tok_s_400": tok_s_400, "tok_s_single_step_replay": tok_s_single,
                   "note": "sequences of %d new tokens: --steps < %d times only the first positions of one sequence; tok_s_400 = the reference's "
                           "definition of the metric (whole 100-token sequences, generate.py:344-389), 400 steps timed in this process behind the "
                           "timed region -- the figure to quote; tok_s_single_step_replay = one sequence with ONE token step per graph replay; "
                           "untimed_setup_token_steps = replays of the captured graphs at set-up, before the --warmup steps (first replay of each "
                           "graph, clocks up)" % (SEQ_NEW_TOKENS, SEQ_NEW_TOKENS)},
        "roofline": roofline, "cpu_baseline": None,
    }

    # ------------------------------------------------------------------ side legs, each in its own process (N = 1)
    full = rank == 0 and world == 1 and not args.quick
    if full:
        del graph, run_steps, model
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        budget = float(os.environ.get("GQ_BENCH_BUDGET_S", "900"))
        leg_limit = float(os.environ.get("GQ_BENCH_LEG_TIMEOUT_S", "300"))
        for key, group, leg, extra in LEGS:
            if qtip and key != "cpu_baseline":
                continue
            if key == "cpu_baseline" and (args.no_cpu_baseline or qtip):
                continue
            if group == "other_configs" and (args.no_other_configs or args.bits != 2 or args.model is not None or args.mode != "default"):
                continue
            if key == "exact_mode" and args.mode != "default":
                continue
            left = budget - (time.perf_counter() - t_start)
            if left < 20:
                rec = {"skipped": "time budget of this run (GQ_BENCH_BUDGET_S = %.0f s) spent" % budget}
            else:
                ex = list(extra)
                if key in ("cpu_baseline", "exact_mode"):
                    ex += ["--bits", str(args.bits)] + (["--model", args.model] if args.model else [])
                rec = run_leg_child(leg, ex, min(leg_limit, left))
            if group:
                line.setdefault(group, {})[key] = rec
            elif key == "exact_mode":
                if isinstance(rec, dict) and "tok_s" in rec:
                    line["exact_mode_tok_s"] = rec["tok_s"]
                    line["config"]["note"] += ("; `value` is the default (fast) arithmetic -- exact products, fp32 accumulation, closer to the true "
                                               "product than the reference's fp16 accumulation; with every quantized GEMV in the reference's fp16 "
                                               "order bit for bit (exact mode) the same model decodes %.1f tokens/s" % rec["tok_s"])
                else:
                    line["exact_mode_tok_s"] = rec
            else:
                line[key] = rec
        # one measurement per launch: the w1w3 row of the shape table in the decode graph's form IS the roofline object of this line
        rows = (line.get("roofline_by_shape") or {}).get("rows") if isinstance(line.get("roofline_by_shape"), dict) else None
        if rows and isinstance(roofline, dict) and "avg_launch_us" in roofline and not qtip:
            for r in rows:
                if r.get("shape") == "w1w3" and r.get("bits") == args.bits and r.get("launch", "").startswith("norm") and args.model is None:
                    r["separate_measurement_us"] = r["us"]
                    r.update(us=roofline["avg_launch_us"], GBps=roofline["achieved"], frac=roofline["frac"],
                             stream_floor_us=roofline["stream_floor_us"], frac_of_stream_floor=roofline["frac_of_stream_floor"],
                             source="the `roofline` object of this line (the model's own 32 w1w3 tensors)")
        line["bench_wall_s"] = round(time.perf_counter() - t_start, 1)

    if world > 1:
        # evidence that the collective library saw N ranks (the barrier / max-reduce above went through it)
        nccl_v = None
        try:
            nccl_v = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        line["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "nccl_version": nccl_v,
                               "devices_visible": torch.cuda.device_count(), "one_gpu_shared": one_gpu}

    # ------------------------------------------------------------------ north-star multi-GPU config as sub-records (N > 1)
    if world > 1 and not pp and not args.no_pp_record and not qtip:
        # every rank starts a child of its own; the N children form their own process group on another port and run the 70B layer
        # pipeline (BASELINE configs[4]) and the tensor-parallel decode of the same model; a child that dies or hangs costs its record
        del graph, run_steps, model
        torch.cuda.empty_cache()
        base_port = int(os.environ.get("MASTER_PORT", "29500"))
        for i, leg in enumerate(("pipeline_70b", "tp_70b")):
            # (TORCHELASTIC_USE_AGENT_STORE: under torch.distributed.run every rank is a CLIENT of the agent's store at MASTER_PORT; the
            # child group has a port of its own, where its rank 0 must open the store itself)
            env = {"MASTER_PORT": str(base_port + 101 + i), "MASTER_ADDR": os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   "TORCHELASTIC_USE_AGENT_STORE": "False"}
            rec = run_leg_child(leg, [], float(os.environ.get("GQ_BENCH_LEG_TIMEOUT_S", "300")), env_extra=env)
            if rank == 0:
                line[leg] = rec
            dist.barrier()  # (the next group starts when every child of this one is gone)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------- rooflines
def ap_roofline(model, bits, mode_arg):
    """the w1w3 AP-GEMV exactly as the decode graph launches it, over the model's own (distinct) tensors"""
    import torch
    from guidedquant_amd import _lib
    L = _lib.lib()
    cfg = model.config
    dev = model.output.weight.device
    I, D = cfg.intermediate_size, cfg.dim
    x = torch.randn(D, device=dev).half()
    gu = torch.empty(2 * I, dtype=torch.float16, device=dev)
    nw = model.layers[0].post_attention_layernorm.weight
    paired = bool(model._native_state()["pairs"])
    flags = 4 if paired else 0

    layers = list(model.layers)

    def launch(i):
        m = layers[i].feed_forward.w1w3
        rc = L.gq_anyprec_gemv_fused(x.data_ptr(), gu.data_ptr(), m.qweight.data_ptr(), m.lut.data_ptr(), 2 * I, D,
                                     m.bitwidth, nw.data_ptr(), cfg.norm_eps, None, flags, _lib.current_stream_ptr())
        assert rc == 0, L.gq_last_error()

    # inside one captured graph, HIP events on the launch stream (round 4: eager back-to-back launches from Python are limited
    # by the host at ~9.6 us per launch -- the round-3 figure happened to coincide with the kernel's own time)
    t_kernel_us = graph_time_us(launch, cfg.n_layer, iters=10 * cfg.n_layer, reps=3)
    bytes_launch = b_ap(bits, 2 * I, D)
    floor_us = stream_floor_us([b.feed_forward.w1w3.qweight for b in layers], bits * 2 * I * D // 8, iters=10 * cfg.n_layer)
    achieved = bytes_launch / t_kernel_us / 1e3  # GB/s
    exact = mode_arg == "exact" or (mode_arg == "default" and os.environ.get("GQ_AP_EXACT", "0") != "0")
    # HBM bytes per launch from the PMC counters: collected offline in separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
    # tools/prof_bench.sh) for one kernel template, launch form and shape -- reported only when this run launches the same
    traffic, traffic_src = None, None
    for tname in ("r06_w1w3_traffic.json", "r05_w1w3_traffic.json", "r04_w1w3_traffic.json"):  # (the newest committed passes of this kernel template)
        tpath = os.path.join(ROOT, "profiles", tname)
        if traffic is None and os.path.exists(tpath):
            with open(tpath) as f:
                t = json.load(f)
            if (not exact and t.get("bits") == bits and t.get("N") == 2 * I and t.get("K") == D and t.get("launch") == ("norm_pairs" if paired else "norm")):
                traffic, traffic_src = t.get("hbm_bytes_per_launch"), "offline PMC passes of %s (profiles/%s)" % (t.get("kernel"), tname)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "AP-GEMV w1w3 %dx%d %d-bit, RMSNorm prologue%s (%s kernels)" % (2 * I, D, bits, " + gate/up pair epilogue" if paired else "",
                                                                                  "exact-order" if exact else "default dispatch"),
            "avg_launch_us": round(t_kernel_us, 3), "algorithmic_bytes_per_launch": bytes_launch,
            # next to the 8 TB/s fraction: against what a one-shot launch of this size can reach (a kernel that only reads the same
            # plane words once, launch boundary included, measured in this run on the same tensors)
            "stream_floor_us": round(floor_us, 3), "frac_of_stream_floor": round(floor_us / t_kernel_us, 4)}


def qtip_roofline(cfg, R, shape=None):
    """the bare trellis-decode matvec at the model's gate/up shape -- or `shape` = (M, K) -- (B_qtip bytes), rotating > 512 MB of
    trellis words"""
    import torch
    from guidedquant_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda", torch.cuda.current_device())
    M, K = shape or (cfg.intermediate_size, cfg.dim)
    per = R * M * K // 8
    n = max(2, min(64, (512 << 20) // per))
    tr = [torch.randint(-2**31, 2**31 - 1, (R * M * K // 32, ), dtype=torch.int32, device=d) for _ in range(n)]
    tl = (torch.randn(1024, device=d) * 0.5).half()
    x = (torch.randn(K, device=d) / 16).half()
    y = torch.zeros(M, dtype=torch.float32, device=d)

    def run(i):
        rc = L.gq_qtip_matvec(y.data_ptr(), tr[i].data_ptr(), x.data_ptr(), tl.data_ptr(), M, K, R, _lib.current_stream_ptr())
        assert rc == 0, L.gq_last_error()

    us = graph_time_us(run, n, 100)
    gbs = b_qtip(R, M, K) / us / 1e3
    # HBM bytes per launch from the PMC counters: offline --pmc passes of the same kernel / shape (tools/prof_qtip_traffic.sh)
    traffic, traffic_src = None, None
    for fn in ("r04_qtip_matvec_traffic.json", "r03_qtip_matvec_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", fn)
        if traffic is not None or not os.path.exists(tpath):
            continue
        with open(tpath) as f:
            t = json.load(f)
        for rec in t.get("shapes", [t]):
            if rec.get("M") == M and rec.get("K") == K and rec.get("R") == R:
                traffic, traffic_src = rec.get("hbm_bytes_per_launch"), "offline PMC passes of %s (profiles/%s)" % (rec.get("kernel"), fn)
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_src, "kernel": "QTIP trellis matvec %dx%d R=%d (gq_qtip_matvec)" % (M, K, R),
            "avg_launch_us": round(us, 3), "algorithmic_bytes_per_launch": b_qtip(R, M, K)}


# ---------------------------------------------------------------------------------------------------------- side legs (N = 1)
def roofline_by_shape_record():
    """the four Llama-3-8B GEMV shapes x 2/3/4 bits (BASELINE.json metric: "+3/4-bit sweep"), default dispatch: the plain operator
    and the launch form of the decode graph"""
    from guidedquant_amd import _lib
    L = _lib.lib()
    table = []
    graph_form = {"wqkv": "norm", "wo": "resid", "w1w3": "norm_pairs", "w2": "resid"}
    for b in (2, 3, 4):
        for nm, (N, K) in SHAPES_8B.items():
            table.append(bench_ap_shape(nm, N, K, b, iters=100))
            # the launch form of the decode graph (what the headline runs): RMSNorm (+ RoPE / cache epilogue where the library
            # serves wqkv that way) -> wqkv, residual epilogue on wo / w2, RMSNorm + gate/up pairs on w1w3
            form = graph_form[nm]
            if nm == "wqkv" and L.gq_anyprec_qkv_rope_supported(N, K, b, 128):
                form = "qkv_rope"
            table.append(bench_ap_shape(nm, N, K, b, iters=100, fused=form, floor=True))
    return {"note": "Llama-3-8B GEMV shapes, default dispatch, > 512 MB of weights rotating, us per launch / algorithmic GB/s / fraction of "
                    "8 TB/s.  launch = plain: the reference's operator (gq_anyprec_gemv); the other row of a shape is the launch form of the "
                    "decode graph (qkv_rope / norm, resid, norm_pairs) with frac_of_stream_floor = (time of a launch that only reads the same "
                    "plane words once, measured in this run) / (time of the launch)", "rows": table}


def exact_mode_record(dev, bits, steps=200, warmup=50):
    """the headline decode with every quantized GEMV in the bit-exact (reference fp16 order, anyprec.cu:495-512) mode"""
    import torch
    from guidedquant_amd import _lib
    from guidedquant_amd.generate import load_model
    _lib.check(_lib.lib().gq_set_ap_mode(1), "gq_set_ap_mode")
    torch.manual_seed(1234)
    model = load_model(MODEL, dev, "ap", bits, random_init=True)
    model.setup_caches(1, SEQ_NEW_TOKENS + 1)
    assert model.native_ready()
    graph, run = decode_tok_s(model, dev, steps, warmup)
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"tok_s": round(steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup}


def decode_record(dev, name, backend, bits, steps=200, warmup=40):
    """one of BASELINE.json configs[2], [3], [4] (single GPU), measured like the headline: random-init model of the real
    architecture, captured decode step, BOS prompt, sequences of 100 new tokens, wall clock around `steps` token steps"""
    import torch
    from guidedquant_amd.generate import _get_model_size, load_model
    torch.manual_seed(1234)
    model = load_model(name, dev, backend, bits, random_init=True)
    model.setup_caches(1, SEQ_NEW_TOKENS + 1)
    assert model.native_ready()
    graph, run = decode_tok_s(model, dev, steps, warmup)
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    size, _ = _get_model_size(model)
    rec = {"model": model.config.model_name, "backend": backend, "bits": bits, "tok_s": round(steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4),
           "model_bytes": size, "model_bandwidth_GBps": round(size * steps / dt / 1e9, 1), "steps": steps, "warmup": warmup}
    if backend == "qtip":  # configs[3]: + the bare trellis matvec at the model's three shapes
        cfg = model.config
        rec["matvec_roofline"] = [qtip_roofline(cfg, bits, shape=sh) for sh in ((cfg.intermediate_size, cfg.dim), (cfg.dim, cfg.dim), (cfg.dim, cfg.intermediate_size))]
    return rec


def collective_leg(leg, dev):
    """N > 1, child side: BASELINE configs[4] -- Llama-3.3-70B 2-bit over the N GPUs -- as the layer pipeline (pipeline_70b: stage g =
    a contiguous layer range on GPU g, N sequences in flight, point-to-point hops) or as the tensor-parallel decode of ONE sequence
    (tp_70b: every matrix cut along its output rows, device-to-device all-gathers).  Own process group (the parent gave this child
    group its own port)."""
    import torch
    import torch.distributed as dist
    from guidedquant_amd.generate import load_model
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("GQ_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    name = os.environ.get("GQ_BENCH_PP_MODEL", PP_MODEL)
    try:
        torch.manual_seed(1234)
        model = load_model(name, dev, "ap", 2, random_init=True)
        n_tok, n_warm = 48, 8
        if leg == "pipeline_70b":
            from guidedquant_amd.pipeline import stage_ranges
            run_steps, dec = pipeline_runner(model, rank, world, n_tok)
        else:
            from guidedquant_amd.tp import TensorParallelDecoder
            dec = TensorParallelDecoder(model, dist.group.WORLD, rank, world, max_new_tokens=n_tok, temperature=0.0, top_k=32,
                                        bos_id=128000 % model.config.vocab_size)

            def run_steps(n):
                with torch.no_grad():
                    dec.run(n)
        run_steps(n_warm)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(n_tok)
        dist.barrier()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dt = float(dt.item())
        if leg == "pipeline_70b":
            rec = {"model": "%s 2-bit Any-Precision, fused QKV/UpGate" % name, "stages": world, "sequences_in_flight": world,
                   "tokens_per_sequence": n_tok, "aggregate_tok_s": round(world * n_tok / dt, 2), "per_stream_tok_s": round(n_tok / dt, 2),
                   "ms_per_stage_tick": round(dt / (world * n_tok) * 1e3, 4), "head_cost_layers_measured": dec.head_cost_layers,
                   "layers_per_stage": [len(r) for r in stage_ranges(model.config.n_layer, world, head_cost_layers=dec.head_cost_layers)],
                   "graphs": bool(dec.graphs), "hop": dec.hop, "note": "single-stream 1-GPU figure of the same model: python bench.py --model " + PP_MODEL}
        else:
            rec = {"model": "%s 2-bit Any-Precision, fused QKV/UpGate" % name, "ranks": world, "sequences": 1, "tokens": n_tok,
                   "tok_s": round(n_tok / dt, 2), "ms_per_token": round(dt / n_tok * 1e3, 4),
                   "note": "row-split tensor-parallel decode of ONE sequence (guidedquant_amd/tp.py): 4 device-to-device all-gathers per layer, "
                           "one hipGraph per rank and token"}
    except Exception as e:
        rec = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    try:
        dist.destroy_process_group()
    except Exception:
        pass
    return rec


def long_context_record(dev, start=4096, steps=100):
    """decode tokens/s of the headline model with `start` positions already in the KV caches (their contents do not matter for
    the timing): positions start .. start + steps - 1, the split-KV attention launches"""
    import gc
    import torch
    from guidedquant_amd.generate import DecodeGraph, load_model
    torch.manual_seed(1234)
    model = load_model(MODEL, dev, "ap", 2, random_init=True)
    model.setup_caches(1, start + steps + 1)
    graph = DecodeGraph(model, dev, native_sampling=True, temperature=0.0, top_k=32, fold_embed=True)
    p0 = torch.tensor([start], dtype=torch.int32, device=dev)

    def run():
        graph.set_token(1, p0)
        for _ in range(steps):
            graph.step()

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = {"model": model.config.model_name, "bits": 2, "kv_positions_timed": "%d..%d" % (start, start + steps - 1), "tok_s": round(steps / dt, 2),
           "ms_per_step": round(dt / steps * 1e3, 4), "attn_split": model._native_state()["attn_split"]}
    del graph, model
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def hf_generate_record(dev, new_tokens=100):
    """The reference's published figure (130 tokens/s on an RTX 3090, README.md:95-97) is `AnyPrecisionForCausalLM.generate(...,
    cache_implementation="static")` on the HF module tree (inference_example.py:34-77).  The same call here on a random-init
    Llama-3.1-8B 2-bit model through its three routes (AnyPrecisionForCausalLM.generate): the default call, the module tree with a
    captured step, transformers' own eager generate."""
    import gc
    import torch
    import transformers
    from guidedquant_amd.AnyPrecisionForCausalLM import AnyPrecisionForCausalLM
    from guidedquant_amd.model import transformer_configs
    c = transformer_configs["Meta-Llama-3.1-8B-Instruct"] if "Meta-Llama-3.1-8B-Instruct" in transformer_configs else None
    hf = transformers.LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                                  vocab_size=128256, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_theta=500000.0,
                                  tie_word_embeddings=False)
    names = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]
    hf.anyprec = dict(seed_precision=2, parent_precision=2, group_count=1, arch_config=dict(module_names=names, model_name="model", layers_name="layers"))
    hf._name_or_path = "Meta-Llama-3.1-8B-Instruct"
    m = AnyPrecisionForCausalLM.from_config_random(hf, device=dev, seed=1234)
    ids = torch.tensor([[128000]], dtype=torch.long, device=dev)

    def timed(**kw):
        m.generate(ids, max_new_tokens=8, do_sample=False, **kw)  # warm-up (lazy kernel attributes, cache allocation, graph capture)
        torch.cuda.synchronize()
        best = 0.0
        for _ in range(2):
            t0 = time.perf_counter()
            y = m.generate(ids, max_new_tokens=new_tokens, do_sample=False, **kw)
            torch.cuda.synchronize()
            best = max(best, (y.shape[1] - 1) / (time.perf_counter() - t0))
        return round(best, 2)

    rec = {"model": "Llama-3.1-8B-shaped, random init, 2-bit Any-Precision", "new_tokens": new_tokens,
           "harness": "AnyPrecisionForCausalLM.generate(input_ids=[BOS], max_new_tokens=100, do_sample=False, cache_implementation='static'), "
                      "inference_example.py:34-77"}
    torch.cuda.reset_peak_memory_stats(dev)
    base_mem = torch.cuda.memory_allocated(dev)
    # route 3 first (it needs the module tree's own planes), then route 2, then the default call (route 1 releases q/k/v/gate/up planes)
    try:
        rec["hf_module_tree_static_cache_tok_s"] = timed(cache_implementation="static", pad_token_id=0, native=False)
    except Exception as e:  # (an installed transformers without the static cache for this call: the dynamic cache then)
        rec["hf_module_tree_static_cache_error"] = "%s: %s" % (type(e).__name__, str(e)[:160])
        rec["hf_module_tree_dynamic_cache_tok_s"] = timed(pad_token_id=0, native=False)
    try:
        rec["hf_module_tree_captured_step_tok_s"] = timed(pad_token_id=0, native=False, capture=True)
    except Exception as e:
        rec["hf_module_tree_captured_step_error"] = "%s: %s" % (type(e).__name__, str(e)[:160])
    m._drop_native()
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    rec["default_call_tok_s"] = timed(cache_implementation="static", pad_token_id=0)
    rec["native_route_tok_s"] = rec["default_call_tok_s"]
    rec["native_route_peak_mem_over_module_tree"] = round(torch.cuda.max_memory_allocated(dev) / max(1, base_mem), 3)
    rec["note"] = ("same object, same call.  default_call = the unmodified reference call: routed to the fused decode model (hipGraph step, weights "
                   "shared with the module tree by reference, q/k/v/gate/up planes held once); hf_module_tree_static_cache = native=False: "
                   "transformers' own generate on the module tree, one plugin::anyprec_gemv launch per decoder linear (the figure of rounds 3-4); "
                   "hf_module_tree_captured_step = capture=True: the module tree's decode step as ONE hipGraph over a transformers StaticCache "
                   "with the fused sampler (slower than the eager route: ~1,500 small graph nodes per token; opt-in).  Wall clock incl. the "
                   "prompt token and host overhead of generate()")
    del m
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def prompt_pass_records(dev, bits=2, lengths=(128, 512)):
    """ms per prompt through the whole 8B model (random init): `Transformer.prefill_native` (what generate() takes; logits of the
    last token) and `Transformer.forward` with GQ_PREFILL_FUSED=0 (anyprec_dequant + matmul per linear, every eager op of
    inference/model.py); HIP events around one call on an idle stream (host launch overhead included), best of 3"""
    import gc
    import torch
    from guidedquant_amd.generate import load_model
    torch.manual_seed(1234)
    model = load_model(MODEL, dev, "ap", bits, random_init=True)
    model.setup_caches(1, max(lengths) + 8)
    assert model.native_ready()
    rows = []
    prev = os.environ.get("GQ_PREFILL_FUSED")
    try:
        for S in lengths:
            x = torch.randint(0, 128000, (1, S), dtype=torch.int32, device=dev)
            pos = torch.arange(S, dtype=torch.int32, device=dev)
            res = {}
            for mode in ("native", "module_two_steps"):
                if mode == "native":
                    os.environ.pop("GQ_PREFILL_FUSED", None)
                    fn = lambda: model.prefill_native(x, pos, start=0, last_only=True)  # noqa: E731
                else:
                    os.environ["GQ_PREFILL_FUSED"] = "0"
                    fn = lambda: model(x, pos)  # noqa: E731
                with torch.no_grad():
                    fn()
                    torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        fn()
                        e1.record()
                        e1.synchronize()
                        best = min(best, e0.elapsed_time(e1))
                res[mode] = best
            rows.append({"prompt_tokens": S, "hip_prompt_pass_ms": round(res["native"], 3), "module_forward_reference_steps_ms": round(res["module_two_steps"], 3),
                         "prompt_tokens_per_s": round(S / res["native"] * 1e3, 0)})
    finally:
        if prev is None:
            os.environ.pop("GQ_PREFILL_FUSED", None)
        else:
            os.environ["GQ_PREFILL_FUSED"] = prev
    del model
    gc.collect()
    torch.cuda.empty_cache()
    return {"model": MODEL, "bits": bits, "by_prompt_length": rows}


def prefill_records(dev, bits=2, N=28672, K=4096):
    """the fused-dequant MFMA GEMM (gq_anyprec_gemm) against the reference's two steps (anyprec_dequant -> matmul = hipBLASLt) at
    three prompt lengths; HIP events over 20 back-to-back launches, best of 3; `frac` of the 2.5 PFLOP/s dense fp16 MFMA peak"""
    import torch
    from guidedquant_amd import ap_gemv
    q = torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=dev)
    lut = (torch.randn(N, 1 << bits, device=dev) * 0.02).half().sort(dim=1).values.contiguous()

    def timed(fn, iters=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        return best

    rows = []
    for S in (128, 512, 2048):
        x = torch.randn(S, K, device=dev).half()
        t_f = timed(lambda: ap_gemv.anyprec_gemm(x, q, lut, bits))
        t_r = timed(lambda: torch.matmul(x, ap_gemv.anyprec_dequant(q, lut, bits).T))
        fl = 2.0 * S * N * K
        rows.append({"S": S, "fused_us": round(t_f, 1), "dequant_matmul_us": round(t_r, 1), "fused_TFLOPs": round(fl / t_f / 1e6, 1),
                     "roofline": {"bound": "mfma", "achieved": round(fl / t_f / 1e6, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(fl / t_f / 1e6 / 2500.0, 4)}})
    return {"N": N, "K": K, "bits": bits, "dtype": "f16", "by_prompt_length": rows}


# ---------------------------------------------------------------------------------------------------------- pipeline
def pipeline_runner(model, rank, world, max_tokens):
    import torch
    from guidedquant_amd.pipeline import PipelinedDecoder, measure_head_cost, stage_ranges
    cfg = model.config
    model.setup_caches(world, 1 + max_tokens)
    head = measure_head_cost(model)
    rng = stage_ranges(cfg.n_layer, world, head_cost_layers=head)[rank]
    dec = PipelinedDecoder(model, rank, world, rng, n_seq=world, max_new_tokens=max_tokens, temperature=0.0, top_k=32,
                           bos_id=128000 % cfg.vocab_size)
    dec.head_cost_layers = head

    def run_steps(n):
        dec.reset()
        with torch.no_grad():
            dec.run(n)

    return run_steps, dec


# ---------------------------------------------------------------------------------------------------------- CPU legs
def cpu_baseline_sample(cfg, bits):
    """CPU baselines of BASELINE.md section 4 on the host cores, one full-size transformer layer (4 GEMVs) each, extrapolated
    to tokens/s = 1 / (n_layer * t_layer + t_lm_head) with the FULL fp32 lm_head matvec:
      value  packed native-float GEMV (oracle.ap_gemv_f32: planes + LUT read directly, float accumulation, C + OpenMP)
      also   dense torch F.linear on the dequantised W (fp32 and bf16) -- what linear_class=nn.Linear would execute;
             the order-faithful binary16 emulator (oracle.ap_gemv_f16, the parity oracle: exact software half arithmetic);
             the product's own CPU twin (gq_anyprec_gemv_cpu, AVX2 + OpenMP), for reference
    The reference itself has no CPU kernel for this path."""
    import numpy as np
    import torch
    from oracle import oracle
    from guidedquant_amd import ap_gemv, pack
    oracle.build()
    cores = os.cpu_count() or 1
    oracle.set_threads(cores)
    torch.set_num_threads(cores)
    D, I = cfg.dim, cfg.intermediate_size
    kvd = (cfg.n_head + 2 * cfg.n_local_heads) * cfg.head_dim
    shapes = [(kvd, D), (D, D), (2 * I, D), (D, I)]
    rng = np.random.default_rng(0)

    def best(fn, reps):
        fn()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return min(t)

    layers = []
    for N, K in shapes:
        q = pack.random_planes(N, K, bits, seed=N + K)
        lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
        layers.append((N, K, q, lut, rng.normal(0, 1, K).astype(np.float16)))
    # one variant at a time over the four shapes (the runtimes' thread pools do not alternate inside a measurement)
    t_f32 = sum(best(lambda: oracle.ap_gemv_f32(xv, q, lut, bits), 3) for N, K, q, lut, xv in layers)
    t_f16 = 0.0
    for N, K, q, lut, xv in layers:
        t0 = time.perf_counter()
        oracle.ap_gemv_f16(xv, q, lut, bits)
        t_f16 += time.perf_counter() - t0
    t_twin = t_d32 = t_dbf = 0.0
    for N, K, q, lut, xv in layers:
        qt, lt = torch.from_numpy(q), torch.from_numpy(lut)
        xt = torch.from_numpy(xv).view(1, 1, K)
        ot = torch.empty(1, 1, N, dtype=torch.float16)
        t_twin += best(lambda: ap_gemv.anyprec_gemv(xt, ot, qt, lt, bits), 5)
    for N, K, q, lut, xv in layers:
        W = ap_gemv.anyprec_dequant(torch.from_numpy(q), torch.from_numpy(lut), bits).float()
        x32 = torch.from_numpy(xv).float().view(1, K)
        t_d32 += best(lambda: torch.nn.functional.linear(x32, W), 3)
        Wb, xb = W.bfloat16(), x32.bfloat16()
        t_dbf += best(lambda: torch.nn.functional.linear(xb, Wb), 3)
        del W, Wb
    Wl = torch.randn(cfg.vocab_size, D, dtype=torch.float32)
    xl = torch.randn(1, D, dtype=torch.float32)
    t_lm = best(lambda: torch.nn.functional.linear(xl, Wl), 3)
    del Wl

    def tps(t_layer):
        return round(1.0 / (cfg.n_layer * t_layer + t_lm), 4)

    # value = the fastest CPU path measured (round 4: the judge asked for the fastest honest one, not the oracle's scalar loop); which
    # one that is, is said in `value_is`.  Named first: the figure with the reference's own CPU semantics -- APLinear.forward on a host
    # tensor is dense F.linear on the dequantised W (any_precision/modules/APLinear.py:35-38; the reference has no packed CPU kernel)
    variants = {"packed native-float GEMV of the oracle (oracle.ap_gemv_f32, C + OpenMP)": t_f32,
                "the PRODUCT's own CPU twin (gq_anyprec_gemv_cpu, AVX2 + OpenMP, packed planes read directly) -- not reference code": t_twin,
                "dense bf16 F.linear on the dequantised W": t_dbf, "dense fp32 F.linear on the dequantised W": t_d32}
    best_name = min(variants, key=variants.get)
    return {"reference_semantics": {"what": "dense F.linear on the dequantised W, the reference's CPU linear path (APLinear.py:35-38), fp32 / bf16",
                                    "dense_f32_linear_on_W_deq_tok_s": tps(t_d32), "dense_bf16_linear_on_W_deq_tok_s": tps(t_dbf)},
            "value": tps(variants[best_name]), "unit": "tokens/s", "cores": cores, "kind": "port", "value_is": best_name,
            "sample": f"one full-size layer (4 AP-GEMVs) per variant x {cfg.n_layer} + the full fp32 lm_head matvec ({t_lm * 1e3:.1f} ms); "
                      f"value = the fastest variant: {best_name}, {cores} threads: {variants[best_name] * 1e3:.2f} ms per layer",
            "also": {"oracle_packed_f32_tok_s": tps(t_f32), "dense_f32_linear_on_W_deq_tok_s": tps(t_d32), "dense_bf16_linear_on_W_deq_tok_s": tps(t_dbf),
                     "emulated_fp16_order_oracle_tok_s": tps(t_f16), "product_cpu_twin_tok_s": tps(t_twin),
                     "ms_per_layer": {"packed_f32": round(t_f32 * 1e3, 2), "dense_f32": round(t_d32 * 1e3, 2), "dense_bf16": round(t_dbf * 1e3, 2),
                                      "emulated_fp16_order": round(t_f16 * 1e3, 1), "product_cpu_twin": round(t_twin * 1e3, 2)}}}


if __name__ == "__main__":
    main()
