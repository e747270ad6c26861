"""decode step time of the 8B 2-bit model at a long context: one captured decode step replayed at a fixed position of a
long KV cache (contents irrelevant for the timing), split-KV attention (default for caches > 1024) against one block per
head (GQ_ATTN_SPLIT=1)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd.generate import load_model

cache = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = load_model("meta-llama/Meta-Llama-3.1-8B-Instruct", "cuda:0", "ap", 2, random_init=True)
d = torch.device("cuda:0")
with torch.device(d):
    m.setup_caches(1, cache)
assert m.native_ready()
tok = torch.tensor([1], dtype=torch.int32, device=d)
out = {"cache": cache, "attn_split": m._native_state()["attn_split"]}
for p in sorted({100, cache // 4, cache // 2, cache - 2}):
    pos = torch.tensor([p], dtype=torch.int32, device=d)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m.decode_native(tok, pos); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            m.decode_native(tok, pos)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(50):
            g.replay()
        e1.record(s); s.synchronize()
    out[f"ms_per_token@{p}"] = round(e0.elapsed_time(e1) / 50, 4)
print(json.dumps(out))
