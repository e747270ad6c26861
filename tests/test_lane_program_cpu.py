"""CPU: the gfx950 lane program (v_perm decode + packed fp16 FMA chains + LDS staging index math)
emulated on the host from the SAME header the HIP kernel compiles, checked bit-for-bit against the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_files

SRC = os.path.join(ROOT, "tests", "host_emul", "ap_emul.cpp")
SO = os.path.join(ROOT, "tests", "host_emul", "libap_emul.so")


@pytest.fixture(scope="module")
def emul():
    hdr = os.path.join(ROOT, "guidedquant_amd", "csrc", "ap_core.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                               "-I", os.path.dirname(hdr), SRC, "-o", SO])
    L = ctypes.CDLL(SO)
    u16p, u32p = ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint32)
    L.gq_emul_ap_gemv.argtypes = [u16p, u32p, u16p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, u16p]

    def run(x, q, lut, bits):
        q = np.ascontiguousarray(q).view(np.uint32)
        _, N, wpr = q.shape
        x = np.ascontiguousarray(x, dtype=np.float16).view(np.uint16)
        lut = np.ascontiguousarray(lut, dtype=np.float16).view(np.uint16)
        out = np.zeros(N, dtype=np.uint16)
        rc = L.gq_emul_ap_gemv(x.ctypes.data_as(u16p), q.ctypes.data_as(u32p), lut.ctypes.data_as(u16p), N,
                               wpr * 32, bits, out.ctypes.data_as(u16p))
        assert rc == 0, rc
        return out

    return run


@pytest.mark.parametrize("path", [p for p in golden_files("ap_b") if int(np.load(p)["bits"]) in (2, 3, 4)])
def test_emulated_lane_program_on_goldens(emul, oracle, path):
    g = np.load(path)
    bits = int(g["bits"])
    K = g["codes"].shape[1]
    if K % 128:
        pytest.skip("quad path needs K % 128 == 0")
    want = oracle.ap_gemv_f16(g["x"], g["qweight"], g["lut"], bits)[0].view(np.uint16)
    got = emul(g["x"], g["qweight"], g["lut"], bits)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("K", [128, 1024, 1152, 2048, 4096, 11008, 14336])
def test_emulated_lane_program_random(emul, oracle, bits, K):
    rng = np.random.default_rng(bits * 100003 + K)
    N = 6
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    # wide-dynamic-range LUT / x so that roundings, subnormals and sign cases all occur
    lut = (rng.normal(0, 1, (N, 1 << bits)) * 10.0**rng.integers(-6, 1, (N, 1))).astype(np.float16)
    x = (rng.normal(0, 1, K) * 10.0**rng.integers(-3, 2, K)).astype(np.float16)
    want = oracle.ap_gemv_f16(x, q, lut, bits)[0].view(np.uint16)
    got = emul(x, q, lut, bits)
    assert np.array_equal(got, want)


# ----------------------------------------------------------------------------- plane-MFMA (fast mode) algorithm
PSRC = os.path.join(ROOT, "tests", "host_emul", "plane_emul.cpp")
PSO = os.path.join(ROOT, "tests", "host_emul", "libplane_emul.so")


@pytest.fixture(scope="module")
def plane_emul():
    hdr = os.path.join(ROOT, "guidedquant_amd", "csrc", "plane_core.h")
    if not os.path.exists(PSO) or os.path.getmtime(PSO) < max(os.path.getmtime(PSRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                               "-I", os.path.dirname(hdr), PSRC, "-o", PSO])
    L = ctypes.CDLL(PSO)
    u16p, u32p, f64p = ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_double)
    L.gq_emul_plane_gemv.argtypes = [u16p, u32p, u16p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, f64p]

    def run(x, q, lut, bits):
        q = np.ascontiguousarray(q).view(np.uint32)
        _, N, wpr = q.shape
        x = np.ascontiguousarray(x, dtype=np.float16).view(np.uint16)
        lut = np.ascontiguousarray(lut, dtype=np.float16).view(np.uint16)
        out = np.zeros(N, dtype=np.float64)
        rc = L.gq_emul_plane_gemv(x.ctypes.data_as(u16p), q.ctypes.data_as(u32p), lut.ctypes.data_as(u16p), N,
                                  wpr * 32, bits, out.ctypes.data_as(f64p))
        assert rc == 0, rc
        return out

    return run


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("K", [256, 1024, 1280, 4096, 11008, 14336])
def test_plane_decomposition_algorithm(plane_emul, oracle, bits, K):
    """multilinear plane decomposition + exact bf8 piece split + B-image addressing reproduce the exact GEMV"""
    rng = np.random.default_rng(bits * 1009 + K)
    N = 20
    codes = rng.integers(0, 1 << bits, (N, K), dtype=np.uint8)
    q = oracle.ap_pack(codes, bits)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
    x = (rng.normal(0, 1, K) * np.where(rng.random(K) < 0.02, 40.0, 1.0)).astype(np.float16)
    y64 = oracle.ap_gemv_f64(x, q, lut, bits)[0]
    got = plane_emul(x, q, lut, bits)
    W = oracle.ap_dequant(q, lut, bits).astype(np.float64)
    scale = np.abs(W) @ np.abs(x.astype(np.float64))
    assert (np.abs(got - y64) <= 2e-6 * scale).all(), (np.abs(got - y64) / scale).max()


def test_plane_tile_swizzle_and_fp4_masks(plane_emul):
    """direct-to-LDS tile layout: lane <-> (row, segment) is a bijection, every load instruction covers whole 128-byte
    lines, the MFMA lanes' 16-byte reads are bank-conflict free; the 4 nibble masks partition the plane bits and the
    scale operand cancels every single-bit FP4 pattern exactly"""
    import ctypes
    L = ctypes.CDLL(PSO)
    assert L.gq_emul_atile_check() == 0
