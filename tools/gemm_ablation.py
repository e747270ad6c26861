import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from guidedquant_amd import ap_gemv, _lib
sys.path.insert(0, "tools")
from bench_prefill import timed
d = torch.device("cuda:0")
for dbg in (0, 1, 2, 3):
    os.environ["GQ_GEMM_DBG"] = str(dbg); _lib.lib().gq_reset_env_cache()
    for (N, K, S) in ((28672, 4096, 2048), (28672, 4096, 512), (4096, 4096, 2048)):
        bits = 2
        q = torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d)
        lut = (torch.randn(N, 4, device=d) * 0.02).half().contiguous()
        x = torch.randn(S, K, device=d).half()
        t = timed(lambda: ap_gemv.anyprec_gemm(x, q, lut, bits))
        print(dbg, N, K, S, round(t, 1), "us", round(2.0 * S * N * K / t / 1e6, 1), "TF", flush=True)
