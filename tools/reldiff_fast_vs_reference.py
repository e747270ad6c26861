import numpy as np, torch, sys, os
sys.path.insert(0, "/root/repo")
from oracle import oracle
from guidedquant_amd import pack, ap_gemv, _lib
d = torch.device("cuda:0")
for bits, N, K in ((2, 28672, 4096), (2, 4096, 14336), (3, 28672, 4096), (4, 28672, 4096)):
    rng = np.random.default_rng(bits + N)
    q = pack.random_planes(N, K, bits, seed=7)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    x = rng.normal(0, 1, K).astype(np.float16)
    rows = np.unique(rng.integers(0, N, 512))
    qs, ls = np.ascontiguousarray(q[:, rows, :]), lut[rows]
    ref = oracle.ap_gemv_f16(x, qs, ls, bits)[0].astype(np.float64)
    y64 = oracle.ap_gemv_f64(x, qs, ls, bits)[0]
    out = torch.empty(1, 1, N, dtype=torch.float16, device=d)
    _lib.lib().gq_set_ap_mode(0)
    ap_gemv.anyprec_gemv(torch.from_numpy(x.reshape(1, 1, K)).to(d), out, torch.from_numpy(q).to(d), torch.from_numpy(lut).to(d), bits)
    g = out.cpu().numpy().reshape(N)[rows].astype(np.float64)
    nr = np.linalg.norm(ref)
    print(f"bits={bits} {N}x{K}: ||fast-ref||/||ref|| = {np.linalg.norm(g-ref)/nr:.2e}   ||exact-ref||/||ref|| = {np.linalg.norm(y64-ref)/nr:.2e}   ||fast-exact||/||exact|| = {np.linalg.norm(g-y64)/np.linalg.norm(y64):.2e}   max|fast-ref|/rms(ref) = {np.abs(g-ref).max()/np.sqrt((ref**2).mean()):.2e}")
