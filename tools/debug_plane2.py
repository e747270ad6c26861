import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib, ap_gemv
from oracle import oracle
L = _lib.lib(); L.gq_set_ap_mode(0)
d = torch.device("cuda:0")
bits, N, K = 2, 16, 1024
codes = np.ones((N, K), dtype=np.uint8)  # L bit set everywhere
q = torch.from_numpy(oracle.ap_pack(codes, bits)).to(d)
lut = torch.from_numpy(np.tile(np.array([0,1,0,1], dtype=np.float16), (N,1))).to(d)
out = torch.zeros(1,1,N,dtype=torch.float16,device=d)
res = np.zeros(K)
for e in range(K):
    x = torch.zeros(1,1,K,dtype=torch.float16,device=d); x[0,0,e] = 1.0
    ap_gemv.anyprec_gemv(x, out, q, lut, bits)
    res[e] = float(out[0,0,0])
bad = np.nonzero(res != 1.0)[0]
print("n bad", len(bad))
for e in bad[:40]:
    c, t, j = e // 256, (e % 256) // 8, e % 8
    print(e, "c", c, "t", t, "v", t % 8, "kb", t // 8, "j", j, "s", 7 - j, "got", res[e])
import collections
print("by s:", collections.Counter((7 - (b % 8)) for b in bad))
print("by c:", collections.Counter((b // 256) for b in bad))
print("by v:", collections.Counter(((b % 256) // 8) % 8 for b in bad))
print("by kb:", collections.Counter(((b % 256) // 8) // 8 for b in bad))
print("values:", collections.Counter(res[bad]))
