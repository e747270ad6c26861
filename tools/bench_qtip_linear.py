"""Fused QTIP linear micro-benchmark: transform-in + matvec (A), transform-out (B) and the bare matvec per decode shape of a
7B-like model, rotating over > 512 MB of trellis data, HIP events around one captured graph."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib

L = _lib.lib()
d = torch.device("cuda:0")
R = 2


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
    out = []
    for name, Ms, K, pro in (("qkv", [4096, 4096, 4096], 4096, 1), ("o", [4096], 4096, 0), ("gate_up", [8192, 8192], 4096, 1),
                             ("gate_up_7b", [11008, 11008], 4096, 1), ("down", [4096], 8192, 2)):
        per = sum(R * M * K // 8 for M in Ms)
        n = max(2, min(48, (512 << 20) // per))
        tr = [[torch.randint(-2**31, 2**31 - 1, (R * M * K // 32,), dtype=torch.int32, device=d) for M in Ms] for _ in range(n)]
        tl = (torch.randn(1024, device=d) * 0.5).half()
        su = torch.ones(K, device=d); x = torch.randn(K, device=d).half(); x2 = torch.randn(K, device=d).half(); nw = torch.ones(K, device=d).half()
        y32 = [torch.zeros(4 * M, device=d) for M in Ms]  # (room for 4 split-K parts)
        sv = [torch.ones(M, device=d) for M in Ms]
        o16 = [torch.zeros(M, device=d, dtype=torch.float16) for M in Ms]
        descs = [(_lib.GqQtipIn * len(Ms))(*[_lib.GqQtipIn(tr[i][j].data_ptr(), su.data_ptr(), tl.data_ptr(), y32[j].data_ptr(), M)
                                             for j, M in enumerate(Ms)]) for i in range(n)]
        dout = (_lib.GqQtipOut * len(Ms))(*[_lib.GqQtipOut(y32[j].data_ptr(), sv[j].data_ptr(), None, o16[j].data_ptr(), M) for j, M in enumerate(Ms)])
        st = _lib.current_stream_ptr
        res = {"linear": name, "M": Ms, "K": K}
        for p, label in ((0, "A_plain"), (pro, "A_pro")):
            if p == pro and pro == 0:
                continue
            res[label + "_us"] = round(timed(lambda i: _lib.check(L.gq_qtip_linear_in(x.data_ptr(), x2.data_ptr(), nw.data_ptr(), 1e-5, p, K, R, len(Ms),
                                                                                       descs[i], 0, None, 1, st()), "A"), n), 2)
        for ks in [int(v) for v in os.environ.get("GQ_BENCH_KS", "").split(",") if v]:  # K ranges per band (GQ_BENCH_KS=2,3,4)
            res["A_pro_ks%d_us" % ks] = round(timed(lambda i: _lib.check(L.gq_qtip_linear_in(x.data_ptr(), x2.data_ptr(), nw.data_ptr(), 1e-5, pro, K, R, len(Ms),
                                                                                                descs[i], 0, None, ks, st()), "A"), n), 2)
        res["planned_ks"] = int(L.gq_qtip_plan_ksplit(len(Ms), (__import__("ctypes").c_uint32 * len(Ms))(*Ms), K, 4))
        if all(M & (M - 1) == 0 for M in Ms):
            res["B_us"] = round(timed(lambda i: _lib.check(L.gq_qtip_linear_out(len(Ms), dout, st()), "B"), 1), 2)
        res["matvec_sum_us"] = round(timed(lambda i: [_lib.check(L.gq_qtip_matvec(y32[j].data_ptr(), tr[i][j].data_ptr(), x.data_ptr(), tl.data_ptr(), M, K, R, st()), "mv")
                                                      for j, M in enumerate(Ms)], n), 2)
        res["MB"] = round(per / 1e6, 1)
        out.append(res)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
