"""lm_head (dense fp16 GEMV, 128256 x 4096) against the one-shot streaming floor of the same bytes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from guidedquant_amd import _lib
L = _lib.lib(); d = torch.device("cuda:0")
V, D = 128256, 4096
Ws = [(torch.randn(V, D, device=d) * 0.02).half() for _ in range(2)]
x = torch.randn(D, device=d).half(); nw = torch.ones(D, device=d).half(); out = torch.zeros(V, dtype=torch.float16, device=d)
def run(i):
    assert L.gq_dense_gemv_f16(x.data_ptr(), Ws[i].data_ptr(), out.data_ptr(), V, D, nw.data_ptr(), 1e-5, _lib.current_stream_ptr()) == 0
us = bench.graph_time_us(run, 2, 20)
fl = bench.stream_floor_us(Ws, V * D * 2, 20)
print("lm_head %.1f us (%.0f GB/s)  stream floor %.1f us (%.0f GB/s)  frac_of_floor %.3f" % (us, V * D * 2 / us / 1e3, fl, V * D * 2 / fl / 1e3, fl / us))
