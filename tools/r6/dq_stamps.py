"""phase stamps of the dq kernel (GQ_STAMPS build of ap_gemv.hip): per wave of the middle block, s_memrealtime (10 ns) at kernel start of
the main loop and per item {top, plane words in hand, next item requested, decoded}"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from guidedquant_amd import _lib
L = _lib.lib(); L.gq_set_ap_mode(0)
os.environ["GQ_DQ"] = "7"; os.environ["GQ_DQ_MIN_MWEIGHTS"] = "0"; L.gq_reset_env_cache()
d = torch.device("cuda:0")
N, K = 28672, 4096
for bits in (2, 3, 4):
    g = torch.Generator(device=d); g.manual_seed(1)
    nbuf = 24
    qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g) for _ in range(nbuf)]
    lut = (torch.randn(N, 1 << bits, device=d, generator=g) * 0.02).half().sort(dim=1).values.contiguous()
    x = torch.randn(1, 1, K, device=d, generator=g).half()
    out = torch.empty(1, 1, N, dtype=torch.float16, device=d)
    dbg = torch.zeros(16 * 32, dtype=torch.int64, device=d)
    L.gq_debug_set_timing_buffer(dbg.data_ptr())
    rows = []
    for i in range(nbuf):
        dbg.zero_()
        assert L.gq_anyprec_gemv(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), lut.data_ptr(), 1, N, K, bits, 0, _lib.current_stream_ptr()) == 0
        torch.cuda.synchronize()
        t = dbg.cpu().numpy().reshape(16, 32).astype(np.float64)
        if i < 4:
            continue
        t0 = t[:, 0].min()
        rows.append((t - t0) / 100.0)
    r = np.median(np.array(rows), axis=0)   # [wave][stamp] us
    print("bits", bits, "(us from the first wave's loop start; per item: top, words in hand, next requested, decoded)")
    for w in (0, 7, 8, 15):
        vals = [v for v in r[w][:17] if v >= 0]
        print("  wave %2d: start %.2f | " % (w, vals[0]) + " | ".join(" ".join("%.2f" % v for v in vals[1 + 4 * k:5 + 4 * k]) for k in range(4)))
    L.gq_debug_set_timing_buffer(None)
    del qs
