"""dispatch ramp of an exact-order launch (GQ_STAMPS=2 build of ap_gemv.hip: tools/build_variant.sh stamps2 ap_gemv.hip -DGQ_STAMPS=2; run with
GQ_LIB_PATH=guidedquant_amd/abl_stamps2/libgq_hip.so GQ_AP_PT=0): s_memrealtime (10 ns, one clock for the chip) at the start and the end of
EVERY block; printed: when blocks start and end relative to the first start, the HIP-event time of the same launches"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from guidedquant_amd import _lib
L = _lib.lib(); L.gq_set_ap_mode(1)
d = torch.device("cuda:0")
bits = int(os.environ.get("BITS", "2"))
for nm, (N, K) in {"w1w3": (28672, 4096), "w2": (4096, 14336), "wqkv": (6144, 4096), "wo": (4096, 4096)}.items():
    g = torch.Generator(device=d); g.manual_seed(1)
    nbuf = 24
    qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g) for _ in range(nbuf)]
    lut = (torch.randn(N, 1 << bits, device=d, generator=g) * 0.02).half().sort(dim=1).values.contiguous()
    x = torch.randn(1, 1, K, device=d, generator=g).half()
    out = torch.empty(1, 1, N, dtype=torch.float16, device=d)
    dbg = torch.zeros(2 * 2048, dtype=torch.int64, device=d)
    L.gq_debug_set_timing_buffer(dbg.data_ptr())
    rows, evs = [], []
    for i in range(nbuf):
        dbg.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert L.gq_anyprec_gemv(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), lut.data_ptr(), 1, N, K, bits, 0, _lib.current_stream_ptr()) == 0
        e1.record()
        torch.cuda.synchronize()
        t = dbg.cpu().numpy().reshape(-1, 2).astype(np.float64)
        t = t[t[:, 0] > 0]
        if i < 4:
            continue
        t0 = t[:, 0].min()
        rows.append(((t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0))
        evs.append(e0.elapsed_time(e1) * 1e3)
    st = np.median(np.array([r[0] for r in rows]), axis=0); en = np.median(np.array([r[1] for r in rows]), axis=0)
    pct = lambda a: " ".join("%.2f" % np.percentile(a, p) for p in (0, 10, 50, 90, 99, 100))
    print("%s bits %d: %d blocks | start (min p10 p50 p90 p99 max) %s | end %s | block time %s | HIP events %.2f us (median)" %
          (nm, bits, len(st), pct(st), pct(en), pct(en - st), float(np.median(evs))), flush=True)
    L.gq_debug_set_timing_buffer(None)
    del qs
