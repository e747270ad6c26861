import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib, ap_gemv, pack
from oracle import oracle
L = _lib.lib(); L.gq_set_ap_mode(0)
d = torch.device("cuda:0")
def run(x, q, lut, bits):
    K = q.shape[2]*32; N = q.shape[1]
    out = torch.zeros(1,1,N,dtype=torch.float16,device=d)
    ap_gemv.anyprec_gemv(torch.from_numpy(x.reshape(1,1,K)).to(d), out, torch.from_numpy(q).to(d), torch.from_numpy(lut).to(d), bits)
    torch.cuda.synchronize(); return out.cpu().numpy().reshape(N).astype(np.float64)
rng = np.random.default_rng(0)
bits, N, K = 2, 16, 1024
codes = rng.integers(0, 4, (N, K), dtype=np.uint8)
q = oracle.ap_pack(codes, bits)
for name, lutrow, x in [
    ("L plane, x=1", [0,1,0,1], np.ones(K)),
    ("H plane, x=1", [0,0,1,1], np.ones(K)),
    ("HL plane, x=1", [0,0,0,1], np.ones(K)),
    ("const, x=1", [1,1,1,1], np.ones(K)),
    ("L plane, x=ramp", [0,1,0,1], (np.arange(K)%64)/8.0),
    ("L plane, x=onehot0", [0,1,0,1], np.eye(1,K,0)[0]*1.0),
    ("L plane, x=onehot5", [0,1,0,1], np.eye(1,K,5)[0]*1.0),
    ("L plane, x=onehot300", [0,1,0,1], np.eye(1,K,300)[0]*1.0),
    ("L plane, x=randn", [0,1,0,1], rng.normal(0,1,K)),
]:
    lut = np.tile(np.array(lutrow, dtype=np.float16), (N,1))
    x16 = x.astype(np.float16)
    got = run(x16, q, lut, bits)
    want = oracle.ap_gemv_f64(x16, q, lut, bits)[0]
    print(name, "\n  got ", np.round(got[:8],3), "\n  want", np.round(want[:8],3))
