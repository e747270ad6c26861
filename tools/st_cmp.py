"""stream kernel (ap_stream.hip) against the round-3 plane kernel on the same inputs: where do they differ?"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib
os.environ.setdefault('GQ_PL_MIN_MWEIGHTS', '0')
L = _lib.lib(); L.gq_set_ap_mode(0)
d = torch.device("cuda:0")
def run(st, x, q, lut, nw, N, K, bits, epi):
    os.environ["GQ_ST"] = str(st); L.gq_reset_env_cache()
    out = torch.zeros(N // (2 if epi & 4 else 1), dtype=torch.float16, device=d)
    rc = L.gq_anyprec_gemv_fused(x.data_ptr(), out.data_ptr(), q.data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr() if nw is not None else None, 1e-5, None, epi, None)
    assert rc == 0, L.gq_last_error()
    torch.cuda.synchronize()
    return out.float().cpu().numpy()
for (N, K, epi) in [(6144, 4096, 0), (28672, 4096, 0), (28672, 4096, 4), (3072, 2048, 0), (10240, 8192, 0), (57344, 8192, 4)]:
    torch.manual_seed(N + K)
    bits = 2
    q = torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d)
    lut = (torch.randn(N, 1 << bits, device=d) * 0.02).half()
    x = torch.randn(K, device=d).half(); nw = (1 + 0.1 * torch.randn(K, device=d)).half()
    a = run(0, x, q, lut, nw, N, K, bits, epi); b = run(1, x, q, lut, nw, N, K, bits, epi)
    diff = np.abs(a - b); bad = np.nonzero(diff > 2e-3 * np.abs(a).max())[0]
    print(N, K, epi, "max|a|", np.abs(a).max(), "maxdiff", diff.max(), "nbad", len(bad), "first bad", bad[:12], "bad%16", sorted(set((bad % 16).tolist()))[:16], "bad//16 %8", sorted(set(((bad // 16) % 8).tolist())))
