"""Prefill micro-benchmark (GPU box): the fused-dequant MFMA GEMM (gq_anyprec_gemm) against the reference's two steps
(anyprec_dequant -> torch.matmul = hipBLASLt) on the Llama-3-8B layer shapes; us per call, TFLOP/s (2 S N K) and the
fraction of the dense fp16 MFMA peak (2500 TFLOP/s)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import ap_gemv  # noqa: E402

SHAPES = {"wqkv": (6144, 4096), "wo": (4096, 4096), "w1w3": (28672, 4096), "w2": (4096, 14336)}


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


if __name__ == "__main__":
    d = torch.device("cuda:0")
    bits_list = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4]
    for bits in bits_list:
        for name, (N, K) in SHAPES.items():
            q = torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d)
            lut = (torch.randn(N, 1 << bits, device=d) * 0.02).half().sort(dim=1).values.contiguous()
            for S in (128, 512, 2048):
                x = torch.randn(S, K, device=d).half()
                t_f = timed(lambda: ap_gemv.anyprec_gemm(x, q, lut, bits))
                t_r = timed(lambda: torch.matmul(x, ap_gemv.anyprec_dequant(q, lut, bits).T))
                fl = 2.0 * S * N * K
                print(json.dumps({"shape": name, "N": N, "K": K, "S": S, "bits": bits, "fused_us": round(t_f, 1), "dequant_matmul_us": round(t_r, 1),
                                  "fused_TFLOPs": round(fl / t_f / 1e6, 1), "frac_of_2500_TF": round(fl / t_f / 1e6 / 2500, 4),
                                  "speedup_vs_reference_steps": round(t_r / t_f, 2)}), flush=True)
