"""attention decode kernel alone (one hipGraph of back-to-back launches), Llama-3-8B head geometry"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib
L = _lib.lib()
d = torch.device("cuda:0")
H, Hkv, HD = 32, 8, 128
S = int(sys.argv[1]) if len(sys.argv) > 1 else 101  # cache length; positions measured: a few up to S - 1
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 1    # split-KV blocks per head
ws = torch.zeros(H * NS * (HD + 2), dtype=torch.float32, device=d)
qkv = torch.randn((H + 2 * Hkv) * HD, device=d).half()
cos = torch.randn(S, HD, device=d).half(); sin = torch.randn(S, HD, device=d).half()
kc = torch.randn(Hkv, S, HD, device=d).half(); vc = torch.randn(Hkv, S, HD, device=d).half()
out = torch.empty(H * HD, dtype=torch.float16, device=d)
for p in sorted({0, 10, 50, 99, S // 4, S // 2, S - 2} & set(range(S))):
    pos = torch.tensor([p], dtype=torch.int32, device=d)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def run():
            if os.environ.get("ROPED"):
                _lib.check(L.gq_attn_decode_roped(qkv.data_ptr(), pos.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), H, Hkv, HD, S,
                                                  0.088, NS, ws.data_ptr(), _lib.current_stream_ptr()), "attn")
                return
            _lib.check(L.gq_attn_decode_split(qkv.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                              out.data_ptr(), H, Hkv, HD, S, 0.088, NS, ws.data_ptr(), _lib.current_stream_ptr()), "attn")
        run(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(200):
                run()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(s); g.replay(); e1.record(s); s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    print(f"nsplit={NS} pos={p}: {best:.2f} us")
