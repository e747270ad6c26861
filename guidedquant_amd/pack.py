"""Any-Precision packed-weight format, host side (numpy).

Writer/reader for `qweight int32[bits, N, K/32]` bit-identical to the reference's
any_precision/quantization/pack.py:304-347 (pack_single_weight / unpack_single_weight), implemented from the
closed form of the layout instead of the reference's packbits + byte-permutation route:

  weight e of row n, plane p (0 = MSB of the code):
     full = (K // 1024) * 1024
     e <  full: chunk = e // 1024, r = e % 1024, tpw = 32,              base = 32 * chunk
     e >= full:                    r = e - full, tpw = (K - full) // 32, base = full // 32
     c = r // (8*tpw), t = (r % (8*tpw)) // 8, j = r % 8
     -> word base+t of qweight[p, n, :], bit 31 - (8c + j)

Used to build synthetic checkpoints and by the converters; it is host logic, the GPU never runs it.
"""
import numpy as np


def _locate(K):
    e = np.arange(K, dtype=np.int64)
    full = (K // 1024) * 1024
    r = np.where(e < full, e % 1024, e - full)
    tpw = np.where(e < full, 32, max((K - full) // 32, 1))
    base = np.where(e < full, 32 * (e // 1024), full // 32)
    c = r // (8 * tpw)
    t = (r % (8 * tpw)) // 8
    j = r % 8
    return (base + t).astype(np.int64), (31 - (8 * c + j)).astype(np.uint32)


def pack_codes(codes, bits):
    """codes uint8[N, K] (values < 2**bits) -> int32[bits, N, K//32]."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    N, K = codes.shape
    if K % 32:
        raise ValueError("K must be a multiple of 32")
    word, bitpos = _locate(K)
    out = np.zeros((bits, N, K // 32), dtype=np.uint32)
    # element e sits in word[e]; elements sharing a word are 32 distinct bit positions -> sum == or
    order = np.argsort(word * 32 + (31 - bitpos.astype(np.int64)), kind="stable")  # word-major, MSB first
    for p in range(bits):
        plane = ((codes >> (bits - 1 - p)) & 1).astype(np.uint8)[:, order]  # [N, K] grouped 32 per word, MSB first
        packed = np.packbits(plane.reshape(N, K // 32, 32), axis=-1)  # [N, K/32, 4] big-endian bytes
        out[p] = (packed[..., 0].astype(np.uint32) << 24) | (packed[..., 1].astype(np.uint32) << 16) | (
            packed[..., 2].astype(np.uint32) << 8) | packed[..., 3].astype(np.uint32)
    return out.view(np.int32)


def unpack_codes(qweight, bits):
    """int32[>=bits, N, K//32] -> uint8[N, K] from the first `bits` planes (any-precision prefix property)."""
    q = np.ascontiguousarray(qweight).view(np.uint32)
    _, N, wpr = q.shape
    K = wpr * 32
    word, bitpos = _locate(K)
    codes = np.zeros((N, K), dtype=np.uint8)
    for p in range(bits):
        b = (q[p][:, word] >> bitpos[None, :]) & 1
        codes |= (b.astype(np.uint8) << (bits - 1 - p))
    return codes


def random_quantized_linear(N, K, bits, seed, lut_std=0.02):
    """Synthetic layer in the reference's format: uniform codes, per-row sorted
    N(0, lut_std^2) fp16 centroids.  Returns (qweight int32[bits,N,K/32], lut fp16[N,2**bits])."""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 1 << bits, size=(N, K), dtype=np.uint8)
    lut = np.sort(rng.normal(0.0, lut_std, size=(N, 1 << bits)).astype(np.float16), axis=1)
    return pack_codes(codes, bits), lut


def random_planes(N, K, bits, seed):
    """Cheap synthetic planes for benchmarking at full size: uniformly random plane words are exactly the
    packed image of uniformly random codes (the packing is a bijection on bits)."""
    rng = np.random.default_rng(seed)
    return rng.integers(-2**31, 2**31, size=(bits, N, K // 32), dtype=np.int64).astype(np.int32)
