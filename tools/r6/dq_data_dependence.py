"""does the time of a GEMV launch depend on the VALUES it reads?  dq kernel (and the default 2-bit stream kernel), w1w3 shape, > 512 MB rotating:
random plane words vs all-zero plane words vs all-ones, random / zero activations -- same instructions, same bytes, same addresses"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from guidedquant_amd import _lib
L = _lib.lib(); L.gq_set_ap_mode(0)
d = torch.device("cuda:0")
N, K = 28672, 4096


def run(bits, fill, xfill, dq):
    os.environ["GQ_DQ"] = "7" if dq else "0"; os.environ["GQ_DQ_MIN_MWEIGHTS"] = "0"; L.gq_reset_env_cache()
    per = bits * N * K // 8
    nbuf = max(2, (512 << 20) // per + 1)
    g = torch.Generator(device=d); g.manual_seed(1)
    if fill == "random":
        qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g) for _ in range(nbuf)]
    else:
        qs = [torch.full((bits, N, K // 32), 0 if fill == "zeros" else -1, dtype=torch.int32, device=d) for _ in range(nbuf)]
    lut = (torch.randn(N, 1 << bits, device=d, generator=g) * 0.02).half().sort(dim=1).values.contiguous()
    x = torch.randn(1, 1, K, device=d, generator=g).half() if xfill == "random" else torch.zeros(1, 1, K, dtype=torch.float16, device=d)
    out = torch.empty(1, 1, N, dtype=torch.float16, device=d)

    def launch(i):
        assert L.gq_anyprec_gemv(x.data_ptr(), out.data_ptr(), qs[i].data_ptr(), lut.data_ptr(), 1, N, K, bits, 0, _lib.current_stream_ptr()) == 0
    return bench.graph_time_us(launch, nbuf, 100)


for bits, dq in ((4, True), (3, True), (2, True), (2, False), (4, False)):
    row = {"bits": bits, "kernel": "dq" if dq else "default dispatch"}
    for fill in ("random", "zeros", "ones"):
        for xf in ("random", "zero"):
            row["planes %s / x %s" % (fill, xf)] = round(run(bits, fill, xf, dq), 2)
    print(json.dumps(row), flush=True)
