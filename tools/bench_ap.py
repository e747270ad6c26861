"""Kernel-level AP-GEMV micro-benchmark (GPU box): achieved algorithmic GB/s per Llama shape and bit-width.

Rotates over enough distinct weight tensors that the working set exceeds the 256 MiB Infinity Cache
(SURVEY.md section 8d), times with HIP events on the launch stream inside one hipGraph replay (bench.bench_ap_shape).
--launch selects the entry point / fusion: plain (gq_anyprec_gemv), norm (RMSNorm prologue), norm_pairs (RMSNorm +
gate/up pair epilogue: the decode graph's w1w3 launch), resid (residual epilogue: the wo / w2 launches).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bench_ap_shape  # noqa: E402

SHAPES = {"wqkv": (6144, 4096), "wo": (4096, 4096), "w1w3": (28672, 4096), "w2": (4096, 14336),
          # Llama-3.3-70B layer shapes (BASELINE config 5)
          "70b_wqkv": (10240, 8192), "70b_wo": (8192, 8192), "70b_w1w3": (57344, 8192), "70b_w2": (8192, 28672)}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", type=int, nargs="*", default=[2, 3, 4])
    ap.add_argument("--shapes", nargs="*", default=["wqkv", "wo", "w1w3", "w2"])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--launch", choices=["plain", "norm", "norm_pairs", "resid", "resid_ws", "qkv_rope", "norm_ho", "norm_pairs_ho", "resid_ho", "qkv_rope_ho"], default="plain")
    a = ap.parse_args()
    for b in a.bits:
        for n in a.shapes:
            N, K = SHAPES[n]
            print(json.dumps(bench_ap_shape(n, N, K, b, a.iters, fused=None if a.launch == "plain" else a.launch)), flush=True)
