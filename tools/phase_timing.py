"""(Needs a library built WITH the stamp sites -- they are compiled out of the shipped one: csrc/gq_internal.h GQ_STAMPS; e.g.
tools/build_variant.sh stamps ap_stream.hip -DGQ_STAMPS=1 and GQ_LIB_PATH=guidedquant_amd/abl_stamps/libgq_hip.so.)
Per-wave phase timestamps (shader clock) of the middle block of the plane-MFMA kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib
os.environ.setdefault('GQ_PL_MIN_MWEIGHTS', '0')
L = _lib.lib(); L.gq_set_ap_mode(0)
L.gq_debug_set_timing_buffer.argtypes = [ctypes.c_void_p]
d = torch.device("cuda:0")
SH = {"wqkv": (6144, 4096), "wo": (4096, 4096), "w1w3": (28672, 4096), "w2": (4096, 14336)}
FUSED = os.environ.get("PT_FUSED", "0") != "0"
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for name in sys.argv[2:] or list(SH):
    N, K = SH[name]
    qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d) for _ in range(max(2, (600 << 20) // (bits * N * K // 8)))]
    lut = (torch.randn(N, 1 << bits, device=d) * 0.02).half()
    x = torch.randn(1, 1, K, device=d).half(); out = torch.empty(1, 1, N, dtype=torch.float16, device=d)
    nw = (1 + 0.1 * torch.randn(K, device=d)).half()
    dbg = torch.zeros(128 + 256, dtype=torch.int64, device=d)
    for i, q in enumerate(qs):
        if i == len(qs) - 1:
            L.gq_debug_set_timing_buffer(dbg.data_ptr())
        if os.environ.get("PT_FUSED") == "2":  # ... with the statistics hand-over (round 5)
            ssq = torch.zeros(_lib.SSQ_SLOTS, dtype=torch.float32, device=d)
            L.gq_ssq_rows(x.data_ptr(), K, ssq.data_ptr(), None)
            L.gq_anyprec_gemv_fused_ho(x.data_ptr(), out.data_ptr(), q.data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5, None,
                                       4 if name == "w1w3" else 0, None, 0, ssq.data_ptr(), None, None)
        elif FUSED:  # the decode step's own launch: RMSNorm prologue (+ gate/up pair epilogue for w1w3)
            L.gq_anyprec_gemv_fused(x.data_ptr(), out.data_ptr(), q.data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5, None,
                                    4 if name == "w1w3" else 0, None)
        else:
            L.gq_anyprec_gemv(x.data_ptr(), out.data_ptr(), q.data_ptr(), lut.data_ptr(), 1, N, K, bits, 0, None)
    torch.cuda.synchronize(); L.gq_debug_set_timing_buffer(None)
    raw = dbg.cpu().numpy()
    t = raw[:128].reshape(16, 8)[:, [0, 6, 7, 1, 2, 3, 4, 5]]
    t0 = t[t > 0].min()
    print(name, "cycles: [start, x landed (planes issued), stats done, image built, after image barrier, main loop done, after barrier, end]")
    for w in range(16):
        if t[w].max() > 0:
            print("  wave", w, [int(v - t0) if v > 0 else None for v in t[w]])
    t2 = raw[128:].reshape(16, 16)
    print("  prologue stamps [before x wait, x landed, statistics done, after statistics barrier, scale known]")
    for w in range(16):
        if t2[w].max() > 0:
            print("  wave", w, [int(v - t0) for v in t2[w] if v > 0])
