"""Kernel-level AP-GEMV micro-benchmark (GPU box): achieved algorithmic GB/s per Llama shape and bit-width.

Rotates over enough distinct weight tensors that the working set exceeds the 256 MiB Infinity Cache
(SURVEY.md section 8d), times with HIP events on the launch stream inside one hipGraph replay.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib  # noqa: E402

SHAPES = {"wqkv": (6144, 4096), "wo": (4096, 4096), "w1w3": (28672, 4096), "w2": (4096, 14336),
          # Llama-3.3-70B layer shapes (BASELINE config 5)
          "70b_wqkv": (10240, 8192), "70b_wo": (8192, 8192), "70b_w1w3": (57344, 8192), "70b_w2": (8192, 28672)}


def b_ap(bits, N, K):
    return bits * N * K // 8 + 2 * N * (1 << bits) + 2 * K + 2 * N


def bench_shape(name, N, K, bits, iters=200, min_ws=512 << 20):
    d = torch.device("cuda:0")
    per = bits * N * K // 8
    nbuf = max(2, min(64, (min_ws + per - 1) // per))
    g = torch.Generator(device=d)
    g.manual_seed(1)
    qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g) for _ in range(nbuf)]
    luts = [(torch.randn(N, 1 << bits, device=d, generator=g) * 0.02).half().sort(dim=1).values.contiguous() for _ in range(nbuf)]
    x = torch.randn(1, 1, K, device=d, generator=g).half()
    out = torch.empty(1, 1, N, dtype=torch.float16, device=d)
    L = _lib.lib()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def run(i):
            rc = L.gq_anyprec_gemv(x.data_ptr(), out.data_ptr(), qs[i % nbuf].data_ptr(), luts[i % nbuf].data_ptr(), 1, N, K,
                                   bits, 0, _lib.current_stream_ptr())
            assert rc == 0, L.gq_last_error()
        for i in range(nbuf):
            run(i)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for i in range(iters):
                run(i)
        graph.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(s)
            graph.replay()
            e1.record(s)
            s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    gbs = b_ap(bits, N, K) / best / 1e3
    return {"shape": name, "N": N, "K": K, "bits": bits, "us": round(best, 3), "GBps": round(gbs, 1),
            "frac_of_8TBps": round(gbs / 8000, 4), "nbuf": nbuf}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", type=int, nargs="*", default=[2, 3, 4])
    ap.add_argument("--shapes", nargs="*", default=["wqkv", "wo", "w1w3", "w2"])
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    for b in a.bits:
        for n in a.shapes:
            N, K = SHAPES[n]
            print(json.dumps(bench_shape(n, N, K, b, a.iters)), flush=True)
