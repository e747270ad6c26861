"""(Needs a library built WITH the stamp sites -- they are compiled out of the shipped one: csrc/gq_internal.h GQ_STAMPS; e.g.
tools/build_variant.sh stamps ap_stream.hip -DGQ_STAMPS=1 and GQ_LIB_PATH=guidedquant_amd/abl_stamps/libgq_hip.so.)
Per-wave phase stamps (shader clock) of the middle block of gq_qtip_matvec:
[start, first tile blocks requested, prologue done (codebook + activations in LDS), item 0 main loop done, item 0 flushed, item 1 ...]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib
L = _lib.lib()
L.gq_debug_set_qtip_timing_buffer.argtypes = [ctypes.c_void_p]
d = torch.device("cuda:0")
R = 2
for M, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
    per = R * M * K // 8
    qs = [torch.randint(-2**31, 2**31 - 1, (per // 4,), dtype=torch.int32, device=d) for _ in range(max(2, (600 << 20) // per))]
    tl = (torch.randn(1024, device=d) * 0.5).half()
    x = torch.randn(K, device=d).half(); y = torch.empty(M, dtype=torch.float32, device=d)
    dbg = torch.zeros(128, dtype=torch.int64, device=d)
    for i, q in enumerate(qs):
        if i == len(qs) - 1:
            L.gq_debug_set_qtip_timing_buffer(dbg.data_ptr())
        L.gq_qtip_matvec(y.data_ptr(), q.data_ptr(), x.data_ptr(), tl.data_ptr(), M, K, R, None)
    torch.cuda.synchronize(); L.gq_debug_set_qtip_timing_buffer(None)
    t = dbg.cpu().numpy().reshape(16, 8)
    t0 = t[t > 0].min()
    print("%dx%d" % (M, K))
    for w in range(16):
        if t[w].max() > 0:
            print("  wave", w, [int(v - t0) for v in t[w] if v > 0])

# the fused transform-in + matvec launch (gq_qtip_linear_in) of the decode groups
for name, Ms, K, pro, ks in (("qkv", [4096, 4096, 4096], 4096, 1, 1), ("qkv", [4096, 4096, 4096], 4096, 1, 3), ("o", [4096], 4096, 0, 2),
                             ("gate_up", [11008, 11008], 4096, 1, 1), ("down_pre", [4096], 11008, 3, 2)):
    per = sum(R * M * K // 8 for M in Ms)
    n = max(2, (600 << 20) // per)
    tr = [[torch.randint(-2**31, 2**31 - 1, (R * M * K // 32,), dtype=torch.int32, device=d) for M in Ms] for _ in range(n)]
    tl = (torch.randn(1024, device=d) * 0.5).half()
    su = torch.ones(K, device=d); x = torch.randn(K, device=d).half(); x2 = torch.randn(K, device=d).half(); nw = torch.ones(K, device=d).half()
    y32 = [torch.zeros(4 * M, device=d) for M in Ms]
    dbg = torch.zeros(256, dtype=torch.int64, device=d)
    for i in range(n):
        desc = (_lib.GqQtipIn * len(Ms))(*[_lib.GqQtipIn(tr[i][j].data_ptr(), su.data_ptr(), tl.data_ptr(), y32[j].data_ptr(), M) for j, M in enumerate(Ms)])
        if i == n - 1:
            L.gq_debug_set_qtip_timing_buffer(dbg.data_ptr())
        _lib.check(L.gq_qtip_linear_in(x.data_ptr(), x2.data_ptr(), nw.data_ptr(), 1e-5, pro, K, R, len(Ms), desc, 0, None, ks, None), "A")
    torch.cuda.synchronize(); L.gq_debug_set_qtip_timing_buffer(None)
    t = dbg.cpu().numpy()[:128].reshape(16, 8)
    t2 = dbg.cpu().numpy()[128:].reshape(16, 8)
    t0 = t[t > 0].min()
    print(name, Ms, K, "ksplit", ks, " engine: [start, tiles requested, prologue done, (main loop done, flushed) per item]; prologue: [entered, table stored, norm done, v written, fwht first pass, fwht done]")
    for w in (0, 5, 10, 15):
        print("  wave", w, [int(v - t0) for v in t[w] if v > 0], " prologue", [int(v - t0) for v in t2[w] if v > 0])
    del tr
