"""start / end of EVERY block of a fast-mode GEMV launch (GQ_STAMPS=2 builds: tools/build_variant.sh ramp_stream ap_stream.hip -DGQ_STAMPS=2,
tools/build_variant.sh ramp_plane ap_plane.hip -DGQ_STAMPS=2; GQ_LIB_PATH selects one): s_memrealtime (10 ns, one clock for the chip).
Printed per launch form of the 8B decode step: when blocks start / end relative to the first start, by block index octile."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from guidedquant_amd import _lib
L = _lib.lib(); L.gq_set_ap_mode(0)
d = torch.device("cuda:0")
bits = int(os.environ.get("BITS", "2"))
FORMS = {"wqkv": ((6144, 4096), "norm"), "wo": ((4096, 4096), "resid"), "w1w3": ((28672, 4096), "norm_pairs"), "w2": ((4096, 14336), "resid")}
for nm in os.environ.get("SHAPES", "wqkv,wo,w1w3,w2").split(","):
    (N, K), form = FORMS[nm]
    g = torch.Generator(device=d); g.manual_seed(1)
    nbuf = 24
    qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g) for _ in range(nbuf)]
    lut = (torch.randn(N, 1 << bits, device=d, generator=g) * 0.02).half().sort(dim=1).values.contiguous()
    x = torch.randn(K, device=d, generator=g).half()
    nw = (1 + 0.1 * torch.randn(K, device=d, generator=g)).half()
    res = torch.randn(N, device=d, generator=g).half()
    out = torch.empty(N, dtype=torch.float16, device=d)
    dbg = torch.zeros(2 * 2048, dtype=torch.int64, device=d)
    L.gq_debug_set_timing_buffer(dbg.data_ptr())
    rows = []
    def launch(i):
        rc = L.gq_anyprec_gemv_fused(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), lut.data_ptr(), N, K, bits,
                                     nw.data_ptr() if form.startswith("norm") else None, 1e-5, res.data_ptr() if form == "resid" else None,
                                     (4 if form == "norm_pairs" else 0) | (1 if form == "resid" else 0), _lib.current_stream_ptr())
        assert rc == 0, rc
    graphs = None
    if os.environ.get("GRAPH", "0") != "0":  # the launch as the LAST of six in a captured graph (kernel arguments where the graph keeps them)
        from guidedquant_amd import _graphs
        launch(0); torch.cuda.synchronize()
        graphs = []
        for i in range(nbuf):
            gr = torch.cuda.CUDAGraph()
            with _graphs.capture(gr):
                for j in range(6):
                    launch((i + j) % nbuf)
            graphs.append(gr)
    for i in range(nbuf):
        dbg.zero_()
        torch.cuda.synchronize()
        if graphs:
            graphs[i].replay()
        else:
            launch(i)
        torch.cuda.synchronize()
        t = dbg.cpu().numpy().reshape(-1, 2).astype(np.float64)
        nb = int((t[:, 0] > 0).sum())
        t = t[:nb]
        if i < 4:
            continue
        t0 = t[:, 0].min()
        rows.append(((t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0))
    st = np.median(np.array([r[0] for r in rows]), axis=0); en = np.median(np.array([r[1] for r in rows]), axis=0)
    pct = lambda a: " ".join("%.2f" % np.percentile(a, p) for p in (0, 10, 50, 90, 100))
    print("%s bits %d %s: %d blocks | start (min p10 p50 p90 max) %s | end %s | block time %s" % (nm, bits, form, len(st), pct(st), pct(en), pct(en - st)))
    oct_ = np.array_split(np.arange(len(st)), 8)
    print("    by block index octile: start " + " ".join("%.2f" % st[o].mean() for o in oct_) + " | end " + " ".join("%.2f" % en[o].mean() for o in oct_) +
          " | time " + " ".join("%.2f" % (en[o] - st[o]).mean() for o in oct_), flush=True)
    if os.environ.get("DUMP"):
        os.makedirs(os.environ["DUMP"], exist_ok=True)
        np.save(os.path.join(os.environ["DUMP"], "ramp_%s.npy" % nm), np.stack([st, en]))
    L.gq_debug_set_timing_buffer(None)
    del qs
