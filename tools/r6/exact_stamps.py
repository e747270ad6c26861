"""phase stamps of the exact-order kernel (ap_gemv_quad_kernel, GQ_STAMPS build of ap_gemv.hip: tools/build_variant.sh stamps ap_gemv.hip
-DGQ_STAMPS=1; run with GQ_LIB_PATH=guidedquant_amd/abl_stamps/libgq_hip.so GQ_AP_PT=0): per wave of the middle block, s_memrealtime
(10 ns) at kernel start, activations in registers, per row step {plane words in hand, done}, behind the loop's barrier, at the end"""
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
    nbuf = 20
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
        t0 = t[:, 0][t[:, 0] > 0].min()
        rows.append(np.where(t > 0, (t - t0) / 100.0, -1.0))
    r = np.median(np.array(rows), axis=0)   # [wave][stamp] us
    print(nm, "bits", bits, "(us from the block's first wave start: start | x in registers | per row step: words in hand, done ... | loop barrier | end)")
    for w in range(16):
        vals = [v for v in r[w] if v >= 0]
        if len(vals) > 2:
            print("  wave %2d: " % w + " ".join("%.2f" % v for v in vals))
    L.gq_debug_set_timing_buffer(None)
    del qs
