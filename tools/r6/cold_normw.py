"""does it matter that the RMSNorm weights of a launch come from HBM?  The decode step reads every layer's norm weights once per token (cold:
2.8 GB of planes have passed through the 256 MB memory-side cache since); bench.py's per-shape table feeds every launch the SAME 8 KB (hot).
us per launch of the 2-bit wqkv / w1w3 forms of the decode graph with one norm-weight tensor vs one per rotating weight buffer."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from guidedquant_amd import _lib
L = _lib.lib(); L.gq_set_ap_mode(0)
d = torch.device("cuda:0")
bits = int(os.environ.get("BITS", "2"))
for nm, (N, K), pairs in (("wqkv", (6144, 4096), 0), ("w1w3", (28672, 4096), 4)):
    g = torch.Generator(device=d); g.manual_seed(1)
    per = bits * N * K // 8
    nbuf = max(2, min(64, ((512 << 20) + per - 1) // per))
    qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g) for _ in range(nbuf)]
    lut = (torch.randn(N, 1 << bits, device=d, generator=g) * 0.02).half().sort(dim=1).values.contiguous()
    x = torch.randn(K, device=d, generator=g).half()
    nws = [(1 + 0.1 * torch.randn(K, device=d, generator=g)).half() for _ in range(nbuf)]
    out = torch.empty(N, dtype=torch.float16, device=d)
    for rep in range(2):
        for mode in ("one tensor (hot)", "one per buffer (cold)"):
            def run(i):
                nw = nws[0] if mode.startswith("one tensor") else nws[i]
                rc = L.gq_anyprec_gemv_fused(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5, None, pairs,
                                             _lib.current_stream_ptr())
                assert rc == 0
            print(nm, "bits", bits, "norm weights:", mode, "%.3f us" % bench.graph_time_us(run, nbuf, 200), flush=True)
    del qs
