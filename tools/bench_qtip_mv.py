"""QTIP trellis matvec micro-benchmark (Llama-2-7b shapes, R = 2): the bare gq_qtip_matvec and the fused gq_qtip_linear_in per
decode group and K split, rotating over > 512 MB of trellis data, HIP events around one captured graph of 100 launches.
Algorithmic bytes of a launch: R M K / 8 + 2048 (codebook) + 2 K + 4 M."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib

L = _lib.lib()
d = torch.device("cuda:0")
R = int(os.environ.get("R", "2"))


def timed(fn, n, iters=100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(n):
            fn(i)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(iters):
                fn(i % n)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(s); g.replay(); e1.record(s); s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    st = _lib.current_stream_ptr
    tl = (torch.randn(1024, device=d) * 0.5).half()
    shapes = ((4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (22016, 4096))
    if os.environ.get("MV_SHAPE"):  # one shape only, "MxK" (PMC passes: every dispatch of the kernel is the same launch)
        shapes = (tuple(int(v) for v in os.environ["MV_SHAPE"].split("x")),)
    for M, K in (shapes[:3] if os.environ.get("MV_ONLY") else shapes):
        per = R * M * K // 8
        n = max(2, min(64, (512 << 20) // per))
        comp = [torch.randint(-2**31, 2**31 - 1, (per // 4,), dtype=torch.int32, device=d) for _ in range(n)]
        x = torch.randn(K, device=d).half()
        y = torch.empty(M, dtype=torch.float32, device=d)
        us = timed(lambda i: _lib.check(L.gq_qtip_matvec(y.data_ptr(), comp[i].data_ptr(), x.data_ptr(), tl.data_ptr(), M, K, R, st()), "qtip"), n)
        b = per + 2048 + 2 * K + 4 * M
        print(json.dumps({"kernel": "qtip_matvec", "M": M, "K": K, "R": R, "us": round(us, 3), "GBps": round(b / us / 1e3, 1),
                          "frac_of_8TBps": round(b / us / 8e6, 4)}), flush=True)
        del comp
    if os.environ.get("MV_ONLY"):
        return
    for name, Ms, K, pro, kss in (("qkv", [4096, 4096, 4096], 4096, 1, (1, 2, 3, 4)), ("o", [4096], 4096, 0, (1, 2, 3, 4)),
                                  ("gate_up", [11008, 11008], 4096, 1, (1, 2, 4)), ("down_pre", [4096], 11008, 3, (1, 2, 3, 4)),
                                  ("gate_up_p2", [8192, 8192], 4096, 1, (1, 2)), ("down_p2", [4096], 8192, 2, (1, 2, 4))):
        per = sum(R * M * K // 8 for M in Ms)
        n = max(2, min(48, (512 << 20) // per))
        tr = [[torch.randint(-2**31, 2**31 - 1, (R * M * K // 32,), dtype=torch.int32, device=d) for M in Ms] for _ in range(n)]
        su = torch.ones(K, device=d); x = torch.randn(K, device=d).half(); x2 = torch.randn(K, device=d).half(); nw = torch.ones(K, device=d).half()
        y32 = [torch.zeros(4 * M, device=d) for M in Ms]
        descs = [(_lib.GqQtipIn * len(Ms))(*[_lib.GqQtipIn(tr[i][j].data_ptr(), su.data_ptr(), tl.data_ptr(), y32[j].data_ptr(), M)
                                             for j, M in enumerate(Ms)]) for i in range(n)]
        res = {"linear": name, "M": Ms, "K": K, "MB": round(per / 1e6, 1)}
        plan = L.gq_qtip_plan_ksplit(len(Ms), (_lib.ctypes.c_uint32 * len(Ms))(*Ms), K, 4) if hasattr(L, "gq_qtip_plan_ksplit") else None
        res["planned_ks"] = plan
        for ks in kss:
            res["A_ks%d_us" % ks] = round(timed(lambda i: _lib.check(L.gq_qtip_linear_in(x.data_ptr(), x2.data_ptr(), nw.data_ptr(), 1e-5, pro, K, R, len(Ms),
                                                                                           descs[i], 0, None, ks, st()), "A"), n), 2)
        print(json.dumps(res), flush=True)
        del tr


if __name__ == "__main__":
    main()
