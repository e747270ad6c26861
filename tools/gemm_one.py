"""one prefill GEMM shape, a few launches (profiling target): python tools/gemm_one.py N K S bits [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import ap_gemv
N, K, S, bits = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
d = torch.device("cuda:0")
q = torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d)
lut = (torch.randn(N, 1 << bits, device=d) * 0.02).half().sort(dim=1).values.contiguous()
x = torch.randn(S, K, device=d).half()
for _ in range(iters):
    ap_gemv.anyprec_gemm(x, q, lut, bits)
torch.cuda.synchronize()
