"""(Needs a library built WITH the stamp sites -- they are compiled out of the shipped one: csrc/gq_internal.h GQ_STAMPS; e.g.
tools/build_variant.sh stamps ap_stream.hip -DGQ_STAMPS=1 and GQ_LIB_PATH=guidedquant_amd/abl_stamps/libgq_hip.so.)
Phase stamps (shader clock, lane 0 of every wave of block 32) of gq_qtip_mlp_mid at Llama-2-7b's MLP width, the sums written
by a launch right in front of it: [start, rows done (sums landed + column sums), tables in LDS, out product, middle, in product, end]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib
L = _lib.lib()
d = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "had_n11008.npz"))
n, Kf = 11008, 172
hk = torch.from_numpy(g["hadK"].astype(np.float32)).to(d)
hkT16, hk16 = hk.t().contiguous().half(), hk.half().contiguous()
yg = torch.randn(n, device=d); yu = torch.randn(n, device=d)
sv = torch.ones(n, device=d) * 0.5; su = torch.ones(n, device=d)
z = torch.zeros(n, device=d)
mid = _lib.GqQtipMid(yg.data_ptr(), yu.data_ptr(), sv.data_ptr(), sv.data_ptr(), hkT16.data_ptr(), su.data_ptr(), hk16.data_ptr(), z.data_ptr(), None, None)
dbg = torch.zeros(128, dtype=torch.int64, device=d)
for i in range(20):
    yg.normal_(); yu.normal_()
    if i == 19:
        L.gq_debug_set_qtip_timing_buffer(dbg.data_ptr())
    _lib.check(L.gq_qtip_mlp_mid(ctypes.pointer(mid), 1, n, Kf, None), "mid")
torch.cuda.synchronize(); L.gq_debug_set_qtip_timing_buffer(None)
t = dbg.cpu().numpy().reshape(16, 8)
t0 = t[t > 0].min()
for w in (0, 2, 5, 10, 15):
    print("wave", w, [int(v - t0) for v in t[w] if v > 0])
# in a graph: 50 launches behind a writer of the sums each
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gr = torch.cuda.CUDAGraph()
    yg.normal_(); s.synchronize()
    with torch.cuda.graph(gr, stream=s):
        for i in range(50):
            yg.mul_(1.0001)
            _lib.check(L.gq_qtip_mlp_mid(ctypes.pointer(mid), 1, n, Kf, _lib.current_stream_ptr()), "mid")
    gr2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr2, stream=s):
        for i in range(50):
            yg.mul_(1.0001)
    for gx, name in ((gr, "writer + mid"), (gr2, "writer alone")):
        gx.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(s); gx.replay(); e1.record(s); s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
        print(name, round(best, 2), "us per iteration")
