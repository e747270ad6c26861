"""ctypes front-end of the CPU oracle + independent numpy restatements.

TEST INFRASTRUCTURE ONLY (see oracle/gq_oracle.c header): importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from
guidedquant_amd/.

Two independent statements are kept on purpose:
  * the C library (fast, used at full sizes and as the CPU baseline), and
  * the numpy functions below (`*_np`), written separately from the closed
    form, used to cross-check the C code at small sizes.
Reference citations are relative to the upstream GuidedQuant tree.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgq_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "gq_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p, u16p, u32p = (ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint16),
                           ctypes.POINTER(ctypes.c_uint32))
        f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
        u32, i32 = ctypes.c_uint32, ctypes.c_int
        L.gq_oracle_d2h.restype = ctypes.c_uint16
        L.gq_oracle_d2h.argtypes = [ctypes.c_double]
        L.gq_oracle_h2d.restype = ctypes.c_double
        L.gq_oracle_h2d.argtypes = [ctypes.c_uint16]
        L.gq_oracle_hfma.restype = ctypes.c_uint16
        L.gq_oracle_hfma.argtypes = [ctypes.c_uint16] * 3
        L.gq_oracle_ap_pack.argtypes = [u8p, u32, u32, i32, u32p]
        L.gq_oracle_ap_unpack.argtypes = [u32p, u32, u32, i32, u8p]
        L.gq_oracle_ap_dequant.argtypes = [u32p, u16p, u32, u32, i32, u16p]
        L.gq_oracle_ap_gemv_f64.argtypes = [u16p, u32p, u16p, u32, u32, u32, i32, f64p]
        L.gq_oracle_ap_gemv_f16.argtypes = [u16p, u32p, u16p, u32, u32, u32, i32, u16p]
        L.gq_oracle_ap_gemv_f32.argtypes = [u16p, u32p, u16p, u32, u32, u32, i32, u16p]
        L.gq_oracle_set_threads.argtypes = [i32]
        L.gq_oracle_max_threads.restype = i32
        L.gq_oracle_lutgemm_f16.argtypes = [u16p, u32p, u16p, u16p, u32, u32, i32, i32, u16p]
        L.gq_oracle_lutgemm_f64.argtypes = [u16p, u32p, u16p, u16p, u32, u32, i32, i32, f64p]
        L.gq_oracle_qtip_decode.argtypes = [u32p, u16p, u32, u32, i32, u16p]
        L.gq_oracle_qtip_matvec.argtypes = [u32p, u16p, u16p, u32, u32, i32, f64p]
        L.gq_oracle_quantlut_sym.argtypes = [u16p, u16p]
        L.gq_oracle_quantlut_sym.restype = None
        L.gq_oracle_hadamard.argtypes = [f32p, f32p, u32, u32, ctypes.c_float]
        L._f32p, L._f64p = f32p, f64p
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _u16(a):
    """view fp16 arrays as their bit patterns (uint16)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        return a.view(np.uint16)
    assert a.dtype == np.uint16
    return a


def set_threads(n):
    lib().gq_oracle_set_threads(int(n))


def max_threads():
    return int(lib().gq_oracle_max_threads())


# --------------------------------------------------------------------------- AP format
def ap_pack(codes, bits):
    """codes uint8[N,K] -> qweight int32[bits,N,K/32] (pack.py:304-321)."""
    codes = _c(codes, np.uint8)
    N, K = codes.shape
    q = np.zeros((bits, N, K // 32), dtype=np.uint32)
    rc = lib().gq_oracle_ap_pack(_p(codes, ctypes.c_uint8), N, K, bits, _p(q, ctypes.c_uint32))
    assert rc == 0, rc
    return q.view(np.int32)


def ap_unpack(qweight, bits):
    """qweight int32[>=bits,N,K/32] -> codes uint8[N,K] using the first `bits` planes (pack.py:324-347)."""
    q = _c(qweight, np.int32).view(np.uint32)
    _, N, wpr = q.shape
    codes = np.zeros((N, wpr * 32), dtype=np.uint8)
    rc = lib().gq_oracle_ap_unpack(_p(q, ctypes.c_uint32), N, wpr * 32, bits, _p(codes, ctypes.c_uint8))
    assert rc == 0, rc
    return codes


def ap_dequant(qweight, lut, bits):
    """-> W fp16[N,K]   (anyprec.cu:294-359 / finetune_utils.py:19-38)."""
    q = _c(qweight, np.int32).view(np.uint32)
    _, N, wpr = q.shape
    l16 = _u16(lut)
    assert l16.shape == (N, 1 << bits)
    W = np.zeros((N, wpr * 32), dtype=np.uint16)
    rc = lib().gq_oracle_ap_dequant(_p(q, ctypes.c_uint32), _p(l16, ctypes.c_uint16), N, wpr * 32, bits,
                                    _p(W, ctypes.c_uint16))
    assert rc == 0, rc
    return W.view(np.float16)


def ap_gemv_f64(x, qweight, lut, bits):
    """Exact GEMV, x fp16[M,K] -> float64[M,N]."""
    q = _c(qweight, np.int32).view(np.uint32)
    _, N, wpr = q.shape
    K = wpr * 32
    x16 = _u16(x).reshape(-1, K)
    M = x16.shape[0]
    l16 = _u16(lut)
    y = np.zeros((M, N), dtype=np.float64)
    rc = lib().gq_oracle_ap_gemv_f64(_p(x16, ctypes.c_uint16), _p(q, ctypes.c_uint32), _p(l16, ctypes.c_uint16), M,
                                     N, K, bits, _p(y, ctypes.c_double))
    assert rc == 0, rc
    return y


def ap_gemv_f16(x, qweight, lut, bits):
    """Order-faithful fp16 GEMV (anyprec.cu:372-542), x fp16[M,K] -> fp16[M,N]."""
    q = _c(qweight, np.int32).view(np.uint32)
    _, N, wpr = q.shape
    K = wpr * 32
    x16 = _u16(x).reshape(-1, K)
    M = x16.shape[0]
    l16 = _u16(lut)
    assert l16.shape == (N, 1 << bits)
    y = np.zeros((M, N), dtype=np.uint16)
    rc = lib().gq_oracle_ap_gemv_f16(_p(x16, ctypes.c_uint16), _p(q, ctypes.c_uint32), _p(l16, ctypes.c_uint16), M,
                                     N, K, bits, _p(y, ctypes.c_uint16))
    assert rc == 0, rc
    return y.view(np.float16)


def ap_gemv_f32(x, qweight, lut, bits):
    """Native-float GEMV (float accumulation, no binary16 emulation): the packed CPU baseline of BASELINE.md section 4.
    x fp16[M,K] -> fp16[M,N]."""
    q = _c(qweight, np.int32).view(np.uint32)
    _, N, wpr = q.shape
    K = wpr * 32
    x16 = _u16(x).reshape(-1, K)
    M = x16.shape[0]
    l16 = _u16(lut)
    assert l16.shape == (N, 1 << bits)
    y = np.zeros((M, N), dtype=np.uint16)
    rc = lib().gq_oracle_ap_gemv_f32(_p(x16, ctypes.c_uint16), _p(q, ctypes.c_uint32), _p(l16, ctypes.c_uint16), M, N, K, bits,
                                     _p(y, ctypes.c_uint16))
    assert rc == 0, rc
    return y.view(np.float16)


# --------------------------------------------------------------------------- numpy restatements
def ap_pack_np(codes, bits):
    """Independent statement of pack.py:304-321 + :12-83 by the byte route the
    reference takes (packbits MSB-first per plane, then the warp byte
    permutation with endianness flip), NOT by the closed form used in C."""
    codes = np.asarray(codes, dtype=np.uint8)
    N, K = codes.shape
    nbytes = K // 8
    planes = np.empty((bits, N, nbytes), dtype=np.uint8)
    for p in range(bits):
        planes[p] = np.packbits(((codes >> (bits - 1 - p)) & 1).astype(bool), axis=1)
    out = np.empty_like(planes)
    full = (nbytes // 128) * 128

    def perm(src, tpw):
        # src [..., nb] with nb = 4*tpw*chunks: byte index b = chunk*4*tpw + c*tpw + t
        # goes to chunk*4*tpw + t*4 + (3-c)  (thread t's little-endian word, MSB byte first)
        nb = src.shape[-1]
        s = src.reshape(*src.shape[:-1], nb // (4 * tpw), 4, tpw)
        s = np.flip(np.swapaxes(s, -1, -2), axis=-1)  # [..., chunk, t, 3-c]
        return s.reshape(*src.shape[:-1], nb)

    if full:
        out[..., :full] = perm(planes[..., :full], 32)
    if nbytes > full:
        out[..., full:] = perm(planes[..., full:], (nbytes - full) // 4)
    return np.ascontiguousarray(out).view("<u4").view(np.int32).reshape(bits, N, K // 32)


def _f16(a64):
    return a64.astype(np.float16)


def ap_gemv_f16_np(x, qweight, lut, bits):
    """Independent (vectorised-over-rows) restatement of anyprec.cu:372-542 for M==1,
    no ksplit.  Uses numpy's float64->float16 RNE conversion for every rounding."""
    codes = ap_unpack(qweight, bits)
    N, K = codes.shape
    x = np.asarray(x, dtype=np.float16).reshape(K).astype(np.float64)
    lut = np.asarray(lut, dtype=np.float16)
    W = np.take_along_axis(lut, codes.astype(np.int64), axis=1).astype(np.float64)  # [N,K]
    nfull, tail = divmod(K, 1024)
    eff = tail // 32
    partial = np.zeros((N, 32), dtype=np.float16)
    for i in range(nfull + (1 if tail else 0)):
        tpw = 32 if i < nfull else eff
        sx = np.zeros((N, tpw), dtype=np.float16)
        sy = np.zeros((N, tpw), dtype=np.float16)
        t = np.arange(tpw)
        for c in (3, 2, 1, 0):
            for k in range(4):
                e0 = 1024 * i + 8 * tpw * c + 8 * t + 2 * k
                sx = _f16(W[:, e0] * x[e0][None, :] + sx.astype(np.float64))
                sy = _f16(W[:, e0 + 1] * x[e0 + 1][None, :] + sy.astype(np.float64))
        s = _f16(sx.astype(np.float64) + sy.astype(np.float64))
        partial[:, :tpw] = _f16(partial[:, :tpw].astype(np.float64) + s.astype(np.float64))
    p = partial
    for sh in (16, 8, 4, 2, 1):
        p = _f16(p[:, :sh].astype(np.float64) + p[:, sh:2 * sh].astype(np.float64))
    return p[:, 0].reshape(1, N)


# --------------------------------------------------------------------------- LUT-GEMM (parity unpinned, see gq_oracle.c)
def lutgemm_f16(x, qweight, alpha, q_bias, bits, group_size, out=None):
    """Reference-order fp16 LUT-GEMM GEMV (lutgemm.cu:24-149), tiles added in ascending order.
    qweight int32[K/32, bits, N], alpha fp16[K/g, bits, N], q_bias fp16[K/g, N]; returns fp16[N] (out += ...)."""
    q = _c(qweight, np.int32).view(np.uint32)
    kt, b, N = q.shape
    assert b == bits
    K = kt * 32
    o = np.zeros(N, dtype=np.uint16) if out is None else _u16(out).copy()
    rc = lib().gq_oracle_lutgemm_f16(_p(_u16(x), ctypes.c_uint16), _p(q, ctypes.c_uint32), _p(_u16(alpha), ctypes.c_uint16),
                                     _p(_u16(q_bias), ctypes.c_uint16), N, K, bits, group_size, _p(o, ctypes.c_uint16))
    assert rc == 0, rc
    return o.view(np.float16)


def lutgemm_f64(x, qweight, alpha, q_bias, bits, group_size):
    q = _c(qweight, np.int32).view(np.uint32)
    kt, b, N = q.shape
    o = np.zeros(N, dtype=np.float64)
    rc = lib().gq_oracle_lutgemm_f64(_p(_u16(x), ctypes.c_uint16), _p(q, ctypes.c_uint32), _p(_u16(alpha), ctypes.c_uint16),
                                     _p(_u16(q_bias), ctypes.c_uint16), N, kt * 32, bits, group_size, _p(o, ctypes.c_double))
    assert rc == 0, rc
    return o


# --------------------------------------------------------------------------- LNQ inner loop
def lnq_cd_block_np(W, B, Hn, C, group_rows, st, end):
    """The sequential 128-column inner loop of update_P (any_precision/quantization/layerwise_quantize.py:93-118), float32
    with the reference's operation order (product rounded, then added), vectorised over the independent rows.
    W, B f32 [N, d]; Hn f32 [G, d, d] (column k of H divided by H[k][k]); C f32 [N, n_cluster].
    Returns (assign u8 [N, end - st], What f32 [N, end - st]); ties of the argmin go to the lowest index."""
    W, B, Hn, C = (np.asarray(a, dtype=np.float32) for a in (W, B, Hn, C))
    N = W.shape[0]
    nb = end - st
    Bb = B[:, st:end].copy()
    grp = np.arange(N) // group_rows
    assign = np.zeros((N, nb), dtype=np.uint8)
    What = np.zeros((N, nb), dtype=np.float32)
    for j in range(nb):
        w = W[:, st + j]
        sol = (w - Bb[:, j]).astype(np.float32)
        dist = np.abs((sol[:, None] - C).astype(np.float32))
        arg = dist.argmin(axis=1)
        val = C[np.arange(N), arg]
        assign[:, j], What[:, j] = arg, val
        if j + 1 < nb:
            delta = (val - w).astype(np.float32)
            h = Hn[grp, st + j, st + j + 1:end]
            Bb[:, j + 1:] = (Bb[:, j + 1:] + (delta[:, None] * h).astype(np.float32)).astype(np.float32)
    return assign, What


# --------------------------------------------------------------------------- QTIP
def qtip_decode(compressed, tlut, M, K, R):
    """compressed int32[R*M*K/32], tlut fp16[512,2] -> W fp16[M,K]  (kernel_decompress.py:5-55)"""
    c = _c(compressed, np.int32).view(np.uint32).reshape(-1)
    assert c.size * 32 == R * M * K
    W = np.zeros((M, K), dtype=np.uint16)
    rc = lib().gq_oracle_qtip_decode(_p(c, ctypes.c_uint32), _p(_u16(tlut).reshape(-1), ctypes.c_uint16), M, K, R,
                                     _p(W, ctypes.c_uint16))
    assert rc == 0, rc
    return W.view(np.float16)


def qtip_matvec(compressed, tlut, x, M, K, R):
    c = _c(compressed, np.int32).view(np.uint32).reshape(-1)
    out = np.zeros(M, dtype=np.float64)
    rc = lib().gq_oracle_qtip_matvec(_p(c, ctypes.c_uint32), _p(_u16(tlut).reshape(-1), ctypes.c_uint16),
                                     _p(_u16(x).reshape(-1), ctypes.c_uint16), M, K, R, _p(out, ctypes.c_double))
    assert rc == 0, rc
    return out


def quantlut_sym(tlut):
    out = np.zeros((65536, 2), dtype=np.uint16)
    lib().gq_oracle_quantlut_sym(_p(_u16(tlut).reshape(-1), ctypes.c_uint16), _p(out, ctypes.c_uint16))
    return out.view(np.float16)


def hadamard(x, scale):
    x = _c(x, np.float32)
    rows, n = x.reshape(-1, x.shape[-1]).shape
    y = np.zeros_like(x)
    rc = lib().gq_oracle_hadamard(_p(x, ctypes.c_float), _p(y, ctypes.c_float), rows, n, float(scale))
    assert rc == 0, rc
    return y


def matmul_hadU(x, hadK=None, transpose=False):
    """x @ H_n / sqrt(n) as matmul_hadU / matmul_hadU_cuda define it (matmul_had.py:70-119); hadK: the K x K factor
    for n = K * 2^j (None for a power of two)."""
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[-1]
    if hadK is None or hadK.size == 0:
        return hadamard(x, n**-0.5)
    Kf = hadK.shape[0]
    hk = hadK.T if transpose else hadK
    inp = hadamard(x.reshape(-1, Kf, n // Kf), n**-0.5).reshape(-1, Kf, n // Kf)
    return (hk.astype(np.float32) @ inp).reshape(x.shape)
