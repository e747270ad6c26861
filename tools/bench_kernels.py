"""Kernel-level micro-benchmark of the non-AP kernels of the path (GPU box): QTIP trellis matvec, LUT-GEMM GEMV,
AP dequant, Hadamard.  Same method as tools/bench_ap.py: rotate over enough distinct weight buffers that the working set
exceeds the 256 MiB Infinity Cache, time one hipGraph of back-to-back launches with HIP events on the launch stream.
Algorithmic bytes per launch as defined in DESIGN.md section 4."""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib  # noqa: E402


def timed(launch, nbuf, iters=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(nbuf):
            launch(i)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(iters):
                launch(i % nbuf)
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(s)
            g.replay()
            e1.record(s)
            s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-ws", type=int, default=512 << 20)
    args = ap.parse_args()
    L = _lib.lib()
    d = torch.device("cuda:0")
    st = _lib.current_stream_ptr
    out = []

    def nb(per):
        return max(2, min(64, (args.min_ws + per - 1) // per))

    # QTIP matvec: Llama-2-7b shapes (wrapper.cpp:557-645), R = 2
    for M, K in ((4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (22016, 4096)):
        R = 2
        per = R * M * K // 8
        n = nb(per)
        comp = [torch.randint(-2**31, 2**31 - 1, (per // 4,), dtype=torch.int32, device=d) for _ in range(n)]
        cb = (torch.randn(1024, device=d) * 0.5).half()
        x = torch.randn(K, device=d).half()
        y = torch.empty(M, dtype=torch.float32, device=d)
        us = timed(lambda i: _lib.check(L.gq_qtip_matvec(y.data_ptr(), comp[i].data_ptr(), x.data_ptr(), cb.data_ptr(), M, K, R, st()), "qtip"), n)
        b = per + 2048 + 2 * K + 4 * M
        out.append({"kernel": "qtip_matvec", "M": M, "K": K, "R": R, "us": round(us, 3), "GBps": round(b / us / 1e3, 1), "frac_of_8TBps": round(b / us / 8e6, 4)})
    # LUT-GEMM: 3-bit, group 128 (lutgemm.cu:24-149)
    for N, K in ((4096, 4096), (28672, 4096), (4096, 14336)):
        bits, gs = 3, 128
        per = K // 32 * bits * N * 4
        n = nb(per)
        qw = [torch.randint(-2**31, 2**31 - 1, (K // 32, bits, N), dtype=torch.int32, device=d) for _ in range(n)]
        alpha = (torch.randn(K // gs, bits, N, device=d) * 0.01).half()
        qb = (torch.randn(K // gs, N, device=d) * 0.01).half()
        x = torch.randn(K, device=d).half()
        y = torch.zeros(N, dtype=torch.float16, device=d)
        us = timed(lambda i: _lib.check(L.gq_lutgemm_gemv(x.data_ptr(), y.data_ptr(), qw[i].data_ptr(), alpha.data_ptr(), qb.data_ptr(), N, K, bits, gs, st()), "lutgemm"), n)
        b = per + alpha.numel() * 2 + qb.numel() * 2 + 2 * K + 2 * N
        out.append({"kernel": "lutgemm_gemv", "N": N, "K": K, "bits": bits, "us": round(us, 3), "GBps": round(b / us / 1e3, 1), "frac_of_8TBps": round(b / us / 8e6, 4)})
        ws = torch.empty(K * 64, dtype=torch.uint8, device=d)
        us = timed(lambda i: _lib.check(L.gq_lutgemm_gemv_ws(x.data_ptr(), y.data_ptr(), qw[i].data_ptr(), alpha.data_ptr(), qb.data_ptr(), N, K, bits, gs,
                                                            ws.data_ptr(), ws.numel(), st()), "lutgemm ws"), n)
        out.append({"kernel": "lutgemm_gemv_ws (tables once + GEMV)", "N": N, "K": K, "bits": bits, "us": round(us, 3), "GBps": round(b / us / 1e3, 1),
                    "frac_of_8TBps": round(b / us / 8e6, 4)})
    # AP dequant (write-bound: 2 bytes per weight out)
    for N, K in ((4096, 4096), (28672, 4096)):
        bits = 2
        per = bits * N * K // 8
        n = nb(per)
        qw = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d) for _ in range(n)]
        lut = (torch.randn(N, 1 << bits, device=d) * 0.02).half()
        W = torch.empty(N, K, dtype=torch.float16, device=d)
        us = timed(lambda i: _lib.check(L.gq_anyprec_dequant(qw[i].data_ptr(), lut.data_ptr(), W.data_ptr(), N, K, bits, st()), "dequant"), n, iters=50)
        b = per + 2 * N * K + lut.numel() * 2
        out.append({"kernel": "anyprec_dequant", "N": N, "K": K, "bits": bits, "us": round(us, 3), "GBps": round(b / us / 1e3, 1), "frac_of_8TBps": round(b / us / 8e6, 4)})
    # Hadamard: rows x n fp32 in place
    for rows, n_ in ((1, 4096), (1, 16384), (64, 8192)):
        x = torch.randn(rows, n_, device=d)
        y = torch.empty_like(x)
        us = timed(lambda i: _lib.check(L.gq_hadamard(x.data_ptr(), y.data_ptr(), rows, n_, 1.0, st()), "had"), 2)
        b = 8 * rows * n_
        out.append({"kernel": "hadamard", "rows": rows, "n": n_, "us": round(us, 3), "GBps": round(b / us / 1e3, 1)})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
