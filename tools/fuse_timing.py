"""Phase stamps of the one-launch wqkv + attention kernel (ap_qkv_attn_kernel; a GQ_STAMPS=1 build of ap_stream.hip:
tools/build_variant.sh fstamps ap_stream.hip -DGQ_STAMPS=1, run with GQ_LIB_PATH=guidedquant_amd/abl_fstamps/libgq_hip.so).
s_memrealtime (100 MHz) per block: GEMV blocks {start, body done, signalled}; head blocks {start, history rows requested, flag seen,
q / row pos in hand, streams done, after the barrier, end}.  Prints, per launch, microseconds relative to the first block's start."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import _lib  # noqa: E402
from guidedquant_amd.model import rope_tables  # noqa: E402

L = _lib.lib()
L.gq_set_ap_mode(0)
d = torch.device("cuda:0")
H, Hkv, hd, K, bits, max_seq = 32, 8, 128, 4096, 2, 101
N = (H + 2 * Hkv) * hd
g = torch.Generator(device=d)
g.manual_seed(1)
nq = 40
qs = [torch.randint(-2**31, 2**31 - 1, (bits, N, K // 32), dtype=torch.int32, device=d, generator=g) for _ in range(nq)]
lut = (torch.randn(N, 1 << bits, device=d, generator=g) * 0.03).half()
nw = (1 + 0.1 * torch.randn(K, device=d, generator=g)).half()
cos, sin = rope_tables(hd, max_seq, 500000.0, d)
kc = torch.randn(1, Hkv, max_seq, hd, device=d, generator=g).half()
vc = torch.randn(1, Hkv, max_seq, hd, device=d, generator=g).half()
qkv = torch.zeros(N, dtype=torch.float16, device=d)
out = torch.zeros(H * hd, dtype=torch.float16, device=d)
flags = torch.zeros(H * _lib.ATTN_FLAG_STRIDE, dtype=torch.int32, device=d)
x = torch.randn(K, device=d, generator=g).half()
dbg = torch.zeros(4096, dtype=torch.int64, device=d)
L.gq_debug_set_timing_buffer(dbg.data_ptr())
pos = torch.tensor([int(os.environ.get("POS", "50"))], dtype=torch.int32, device=d)
scale = 1.0 / math.sqrt(hd)
rows = []
for it in range(nq):
    dbg.zero_()
    _lib.check(L.gq_anyprec_gemv_qkv_rope_attn(x.data_ptr(), qkv.data_ptr(), qs[it].data_ptr(), lut.data_ptr(), N, K, bits, nw.data_ptr(), 1e-5,
                                               pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr(), H, Hkv, hd, max_seq,
                                               out.data_ptr(), scale, flags.data_ptr(), _lib.current_stream_ptr()), "fused")
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(-1, 8)[:224].astype(np.float64)
    if it < 8:
        continue
    t0 = t[:192, 0].min()
    ge, he = (t[:192, :3] - t0) / 100.0, (t[192:, :7] - t0) / 100.0
    rows.append(np.concatenate([[ge[:, 0].max(), np.median(ge[:, 1]), ge[:, 1].max(), ge[:, 2].max()], np.median(he, axis=0), he.max(axis=0)]))
r = np.median(np.array(rows), axis=0)
print("GEMV blocks: last start %.2f us, body done median %.2f / max %.2f, signalled max %.2f" % tuple(r[:4]))
names = ["start", "history requested", "flag seen", "q / row pos in hand", "streams done", "after barrier", "end"]
print("head blocks (median over heads | latest head):")
for i, n in enumerate(names):
    print("  %-22s %6.2f | %6.2f us" % (n, r[4 + i], r[11 + i]))
