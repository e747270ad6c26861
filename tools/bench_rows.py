"""One-pass multi-row AP GEMV: M batch rows through gq_anyprec_gemv (default dispatch), us per launch vs M x the one-row time and
vs one block row per batch row (GQ_PL_ONEPASS=0); > 512 MB of weights rotating, HIP events around a captured graph."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import graph_time_us
from guidedquant_amd import _lib

L = _lib.lib()
d = torch.device("cuda:0")
for name, N, K in (("w1w3", 28672, 4096), ("wqkv", 6144, 4096), ("wo", 4096, 4096)):
    for bits in (2, 4):
        per = bits * N * K // 8
        n = max(2, min(64, (512 << 20) // per))
        qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d) for _ in range(n)]
        lut = (torch.randn(N, 1 << bits, device=d) * 0.02).half().sort(dim=1).values.contiguous()
        res = {"shape": name, "N": N, "K": K, "bits": bits}
        for M in (1, 2, 4, 8):
            x = torch.randn(M, 1, K, device=d).half()
            out = torch.empty(M, 1, N, dtype=torch.float16, device=d)
            for onepass in ("1", "0"):
                if M == 1 and onepass == "0":
                    continue
                os.environ["GQ_PL_ONEPASS"] = onepass
                L.gq_reset_env_cache()
                us = graph_time_us(lambda i: _lib.check(L.gq_anyprec_gemv(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), lut.data_ptr(), M, N, K, bits, 0,
                                                                           _lib.current_stream_ptr()), "gemv"), n, 100)
                res["M%d_%s_us" % (M, "onepass" if onepass == "1" else "rowblocks")] = round(us, 2)
        os.environ.pop("GQ_PL_ONEPASS", None)
        L.gq_reset_env_cache()
        print(json.dumps(res), flush=True)
