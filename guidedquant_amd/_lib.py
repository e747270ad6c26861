"""ctypes binding of libgq_hip.so (the C ABI declared in include/gq_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel call fails, a RuntimeError is
raised (the CPU twins of the Any-Precision ops are entry points of the same library, not a Python fallback).  Nothing here
imports the CPU oracle.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GQ_LIB_PATH") or os.path.join(_HERE, "libgq_hip.so")  # (override: kernel experiments)
_lib = None

# symbols include/gq_hip.h declares; tests check that the built library exports every one
EXPORTS = [
    "gq_version", "gq_last_error", "gq_device_count", "gq_anyprec_gemv", "gq_anyprec_dequant", "gq_lutgemm_gemv",
    "gq_qtip_matvec", "gq_hadamard", "gq_anyprec_gemv_fused", "gq_set_ap_mode", "gq_embed_lookup", "gq_attn_decode",
    "gq_dense_gemv_f16", "gq_sample_topk", "gq_qtip_linear_in", "gq_qtip_linear_out", "gq_qtip_transform", "gq_lutgemm_gemv_ws", "gq_attn_decode_split",
    "gq_attn_decode_qtip", "gq_qtip_linear_out_seg",
    "gq_qtip_plan_ksplit", "gq_qtip_linear", "gq_anyprec_gemv_cpu", "gq_anyprec_dequant_cpu", "gq_anyprec_gemm", "gq_anyprec_gemm_ws", "gq_anyprec_gemm_ws_bytes", "gq_rmsnorm_rows", "gq_rope_cache_rows", "gq_silu_mul_rows", "gq_anyprec_pack", "gq_lnq_cd_block", "gq_reset_env_cache", "gq_debug_set_timing_buffer", "gq_debug_set_qtip_timing_buffer", "gq_debug_exact_plan",
    "gq_anyprec_qkv_rope_supported", "gq_anyprec_gemv_qkv_rope", "gq_attn_decode_roped", "gq_selfcheck", "gq_hop_send", "gq_hop_wait",
    "gq_qtip_mlp_mid", "gq_qtip_linear_in_rows", "gq_anyprec_gemv_fused_ws", "gq_anyprec_gemv_fused_ws_bytes",
    "gq_qtip_linear_out_in", "gq_debug_stream_read", "gq_hop_alloc", "gq_hop_free", "gq_hop_export", "gq_hop_import", "gq_hop_close", "gq_hop_wait_copy",
    "gq_sample_topk_ex", "gq_anyprec_gemv_fused_ho", "gq_ssq_rows", "gq_anyprec_handover_plan", "gq_embed_lookup_ho", "gq_anyprec_gemv_qkv_rope_ho",
    "gq_anyprec_qkv_rope_attn_supported", "gq_anyprec_gemv_qkv_rope_attn", "gq_hop_is_finegrained", "gq_sample_topk_p",
]
ATTN_FLAG_STRIDE = 32  # include/gq_hip.h GQ_ATTN_FLAG_STRIDE
SSQ_SLOTS = 1024  # include/gq_hip.h GQ_SSQ_SLOTS
_VOID = ("gq_reset_env_cache", "gq_debug_set_timing_buffer", "gq_debug_set_qtip_timing_buffer")


class GqQtipIn(ctypes.Structure):  # include/gq_hip.h
    _fields_ = [("trellis", ctypes.c_void_p), ("SU", ctypes.c_void_p), ("tlut", ctypes.c_void_p), ("y32", ctypes.c_void_p),
                ("M", ctypes.c_uint32)]


class GqQtipXf(ctypes.Structure):
    _fields_ = [("y32", ctypes.c_void_p), ("vec", ctypes.c_void_p), ("hadK", ctypes.c_void_p), ("resid", ctypes.c_void_p),
                ("out", ctypes.c_void_p)]


class GqQtipMid(ctypes.Structure):
    _fields_ = [("y32_gate", ctypes.c_void_p), ("y32_up", ctypes.c_void_p), ("SV32_gate", ctypes.c_void_p), ("SV32_up", ctypes.c_void_p),
                ("hadT_right16", ctypes.c_void_p), ("SU_down", ctypes.c_void_p), ("had_left_down16", ctypes.c_void_p), ("z32", ctypes.c_void_p),
                ("gate_out", ctypes.c_void_p), ("up_out", ctypes.c_void_p)]


class GqQtipOut(ctypes.Structure):
    _fields_ = [("y32", ctypes.c_void_p), ("SV32", ctypes.c_void_p), ("resid", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("M", ctypes.c_uint32), ("parts", ctypes.c_uint32)]


def build(force=False):
    """Compile every HIP source for gfx950 into guidedquant_amd/libgq_hip.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j4"]
    if force or os.environ.get("GQ_BUILD_FORCE", "0") != "0":  # (objects of a build with other flags are caught by csrc/.flags)
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C guidedquant_amd/csrc).  There is no fallback: the CPU twins live in the same library.")
        # torch first: it ships its own libamdhip64; the extension must bind to the HIP runtime torch has already loaded, or
        # the process ends up with two runtimes and the extension's sees no device ("no ROCm-capable device is detected")
        import torch  # noqa: F401
        # the CPU twins run on LLVM's OpenMP runtime next to torch's libgomp pool: its idle workers must sleep at once
        # (default: spin for 200 ms), or they take the cores from the torch ops that follow
        os.environ.setdefault("KMP_BLOCKTIME", "0")
        L = ctypes.CDLL(LIB_PATH)
        vp, u32, i32, f32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_float
        L.gq_version.restype = i32
        L.gq_last_error.restype = ctypes.c_char_p
        L.gq_device_count.restype = i32
        L.gq_anyprec_gemv.argtypes = [vp, vp, vp, vp, u32, u32, u32, i32, i32, vp]
        L.gq_anyprec_dequant.argtypes = [vp, vp, vp, u32, u32, i32, vp]
        L.gq_lutgemm_gemv.argtypes = [vp, vp, vp, vp, vp, u32, u32, i32, i32, vp]
        L.gq_lutgemm_gemv_ws.argtypes = [vp, vp, vp, vp, vp, u32, u32, i32, i32, vp, ctypes.c_uint64, vp]
        L.gq_qtip_matvec.argtypes = [vp, vp, vp, vp, u32, u32, i32, vp]
        L.gq_hadamard.argtypes = [vp, vp, u32, u32, f32, vp]
        L.gq_qtip_linear_in.argtypes = [vp, vp, vp, f32, i32, u32, i32, i32, ctypes.POINTER(GqQtipIn), i32, ctypes.POINTER(GqQtipOut), i32, vp]
        L.gq_qtip_linear_out.argtypes = [i32, ctypes.POINTER(GqQtipOut), vp]
        L.gq_qtip_linear_out_seg.argtypes = [i32, ctypes.POINTER(GqQtipOut), vp]
        L.gq_qtip_linear_out_in.argtypes = [ctypes.POINTER(GqQtipOut), vp, f32, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), vp]
        L.gq_qtip_plan_ksplit.argtypes = [i32, ctypes.POINTER(u32), u32, i32]
        L.gq_qtip_linear.argtypes = [vp, vp, vp, f32, i32, u32, i32, i32, ctypes.POINTER(GqQtipIn), ctypes.POINTER(GqQtipOut), i32, vp, vp]
        L.gq_qtip_transform.argtypes = [i32, vp, vp, vp, f32, i32, i32, ctypes.POINTER(GqQtipXf), u32, u32, i32, vp]
        L.gq_qtip_mlp_mid.argtypes = [ctypes.POINTER(GqQtipMid), u32, u32, u32, vp]
        L.gq_qtip_linear_in_rows.argtypes = [vp, u32, u32, i32, i32, ctypes.POINTER(GqQtipIn), i32, vp]
        L.gq_anyprec_gemv_fused.argtypes = [vp, vp, vp, vp, u32, u32, i32, vp, f32, vp, u32, vp]
        L.gq_anyprec_gemv_fused_ws.argtypes = [vp, vp, vp, vp, u32, u32, i32, vp, f32, vp, u32, vp, ctypes.c_size_t, vp]
        L.gq_anyprec_gemv_fused_ws_bytes.argtypes = [u32, u32, i32, u32]
        L.gq_anyprec_gemv_fused_ho.argtypes = [vp, vp, vp, vp, u32, u32, i32, vp, f32, vp, u32, vp, ctypes.c_size_t, vp, vp, vp]
        L.gq_ssq_rows.argtypes = [vp, u32, vp, vp]
        L.gq_anyprec_handover_plan.argtypes = [u32, u32, i32, i32, u32]
        L.gq_embed_lookup_ho.argtypes = [vp, vp, vp, u32, u32, vp, vp]
        L.gq_anyprec_gemv_qkv_rope_ho.argtypes = [vp, vp, vp, vp, u32, u32, i32, vp, f32, vp, vp, vp, vp, vp, u32, u32, u32, u32, vp, vp]
        L.gq_set_ap_mode.argtypes = [i32]
        L.gq_embed_lookup.argtypes = [vp, vp, vp, u32, u32, vp]
        L.gq_attn_decode.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, vp]
        L.gq_attn_decode_split.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, vp, vp]
        L.gq_attn_decode_qtip.argtypes = [ctypes.POINTER(GqQtipOut), vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, vp, vp]
        L.gq_dense_gemv_f16.argtypes = [vp, vp, vp, u32, u32, vp, f32, vp]
        L.gq_sample_topk.argtypes = [vp, u32, i32, f32, u32, vp, vp, vp, vp, vp, vp, vp]
        L.gq_sample_topk_ex.argtypes = [vp, u32, i32, f32, u32, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp, vp, u32, vp, vp]
        L.gq_sample_topk_p.argtypes = [vp, u32, i32, f32, f32, u32, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp, vp, u32, vp, vp]
        L.gq_anyprec_gemv_cpu.argtypes = [vp, vp, vp, vp, u32, u32, u32, i32, i32, i32]
        L.gq_anyprec_dequant_cpu.argtypes = [vp, vp, vp, u32, u32, i32, i32]
        L.gq_anyprec_gemm.argtypes = [vp, vp, vp, vp, u32, u32, u32, i32, vp]
        L.gq_anyprec_gemm_ws.argtypes = [vp, vp, vp, vp, u32, u32, u32, i32, vp, ctypes.c_size_t, vp]
        L.gq_anyprec_gemm_ws_bytes.argtypes = [u32, u32, u32, i32]
        L.gq_rmsnorm_rows.argtypes = [vp, vp, vp, vp, u32, u32, f32, vp]
        L.gq_rope_cache_rows.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, u32, vp]
        L.gq_silu_mul_rows.argtypes = [vp, vp, u32, u32, i32, vp]
        L.gq_anyprec_pack.argtypes = [vp, vp, u32, u32, i32, vp]
        L.gq_lnq_cd_block.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, u32, u32, vp]
        L.gq_debug_set_timing_buffer.argtypes = [vp]
        L.gq_debug_set_qtip_timing_buffer.argtypes = [vp]
        L.gq_debug_exact_plan.argtypes = [u32, u32, i32, i32, vp]
        L.gq_hop_send.argtypes = [vp, vp, u32, vp, vp, u32, vp]
        L.gq_hop_wait.argtypes = [vp, vp, u32, vp, u32, vp]
        L.gq_debug_stream_read.argtypes = [vp, ctypes.c_size_t, vp, vp]
        L.gq_hop_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
        L.gq_hop_free.argtypes = [vp]
        L.gq_hop_export.argtypes = [vp, vp]
        L.gq_hop_import.argtypes = [vp, ctypes.POINTER(vp)]
        L.gq_hop_close.argtypes = [vp]
        L.gq_hop_is_finegrained.argtypes = [vp]
        L.gq_hop_wait_copy.argtypes = [vp, vp, u32, vp, u32, vp, vp, u32, vp]
        L.gq_anyprec_qkv_rope_supported.argtypes = [u32, u32, i32, u32]
        L.gq_anyprec_gemv_qkv_rope.argtypes = [vp, vp, vp, vp, u32, u32, i32, vp, f32, vp, vp, vp, vp, vp, u32, u32, u32, u32, vp]
        L.gq_attn_decode_roped.argtypes = [vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, vp, vp]
        L.gq_anyprec_qkv_rope_attn_supported.argtypes = [u32, u32, i32, u32, u32, u32]
        L.gq_anyprec_gemv_qkv_rope_attn.argtypes = [vp, vp, vp, vp, u32, u32, i32, vp, f32, vp, vp, vp, vp, vp, u32, u32, u32, u32, vp, f32, vp, vp]
        for name in EXPORTS:
            if name in _VOID:
                getattr(L, name).restype = None
            elif name not in ("gq_last_error", "gq_anyprec_gemm_ws_bytes", "gq_anyprec_gemv_fused_ws_bytes"):
                getattr(L, name).restype = i32
        L.gq_anyprec_gemm_ws_bytes.restype = ctypes.c_size_t
        L.gq_anyprec_gemv_fused_ws_bytes.restype = ctypes.c_size_t
        L.gq_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


_selfchecked = False


def _ensure_selfcheck():
    """one-time hardware self-check (LDS out-of-range reads return zero: csrc/capi.hip) -- fail loudly, never corrupt sums.  Run from
    the first GPU entry point (every launch asks for the stream pointer), not from lib(): processes that only use the CPU entry
    points never initialise the GPU runtime, and a first use under graph capture DEFERS the check instead of dropping it."""
    global _selfchecked
    if _selfchecked:
        return
    import torch
    if os.environ.get("GQ_SELFCHECK", "1") == "0" or not torch.cuda.is_available():
        _selfchecked = True
        return
    if torch.cuda.is_current_stream_capturing():
        return  # (a synchronous copy is illegal here: the next call outside a capture runs it)
    rc = lib().gq_selfcheck()
    if rc != 0:
        msg = lib().gq_last_error()
        raise RuntimeError(f"gq_selfcheck: {msg.decode() if msg else rc}")
    _selfchecked = True


GQ_ENOTSUP = -95  # include/gq_hip.h: valid request this build has no kernel for


def check(rc, what):
    if rc != 0:
        msg = lib().gq_last_error()
        raise RuntimeError(f"{what}: {msg.decode() if msg else 'error'} (code {rc})")


def current_stream_ptr():
    import torch
    _ensure_selfcheck()
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
