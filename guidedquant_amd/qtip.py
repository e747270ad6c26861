"""QTIP decode path on MI355X: the operator surface of the reference's QTIP backend.

  * `qtip_kernels` namespace: `decompress_matvec_16_9_{R}_1_{M}_1_{K}(out, compressed, x, codebook)` for ANY (R, M, K)
    -- the reference compiles 73 fixed shapes (qtip/qtip-kernels/src/wrapper.cpp:557-645, qtip_torch.cu:14-57); here
    one runtime-shaped HIP kernel (gq_qtip_matvec) serves every name.  Same argument order and checks.
  * torch ops `quip_lib::decompress_matvec_qtip_{m}_{n}_{k}_{R}(compressed, x, codebook) -> Tensor[1, m]`
    (inference/lib/codebook/__init__.py:90-112) with fake impls; registered on demand by `quip_lib_op(m, k, R)` and
    up-front for every linear shape of the models in `model.transformer_configs`.
  * `hadamard::hadamard(x, scale)` (inference/lib/utils/matmul_had.py:96-106) on gq_hadamard, `get_hadK`,
    `matmul_hadU_cuda` / `matmul_hadUt_cuda` (matmul_had.py:13-67,109-123).
  * `BitshiftLinear.forward` eval path (inference/lib/codebook/bitshift.py:415-472) and `QuantizedLinear`
    (qtip/lib/linear/quantized_linear.py:12-153): buffers `trellis int16[(N/16)(K/16), 16R]`, `tlut fp16[512,2]`,
    `SU fp16[K]`, `SV fp32[N]`, `rcp`, `tp_rank`, non-persistent `had_left/had_right`.

The non-power-of-two Hadamard factors (172, 156, 140, 124, 116, 108, 60, 52, 36, 28, 20, 12) are literal tables in the
reference (~95k lines of matmul_had.py); they are data of the checkpoint format, not code: `get_hadK` reads them from the
bit-packed data file guidedquant_amd/data/hadamard_factors.npz (11 KB, written by tools/make_hadamard_tables.py from the
reference in the authoring container); GQ_HADAMARD_TABLES may name an .npz (keys "had{K}") that overrides it.
"""
import math
import os
import re
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import _lib


def _chk(cond, msg):
    if not cond:
        raise RuntimeError(msg)


# --------------------------------------------------------------------------------------------- qtip_kernels namespace
def decompress_matvec(out, compressed, x, codebook, R, M, K):
    """qtip_torch.cu:14-57 semantics: out f32 [M,1] (written), compressed i32 [R*M*K/32], x f16 [K,1], codebook f16 [1024]."""
    _chk(out.dtype == torch.float32 and out.dim() == 2, "out must be a 2-D float32 tensor")
    _chk(compressed.dtype == torch.int32 and compressed.dim() == 1, "compressed must be a 1-D int32 tensor")
    _chk(x.dtype == torch.float16 and x.dim() == 2, "x must be a 2-D float16 tensor")
    _chk(codebook.dtype == torch.float16 and codebook.dim() == 1, "codebook must be a 1-D float16 tensor")
    _chk(out.is_cuda and compressed.is_cuda and x.is_cuda and codebook.is_cuda, "all tensors must be on the GPU")
    _chk(out.is_contiguous() and compressed.is_contiguous() and codebook.is_contiguous(), "tensors must be contiguous")
    _chk(out.size(0) == M and out.size(1) == 1, f"out must have shape ({M}, 1)")
    _chk(x.numel() == K, f"x must have {K} elements")
    _chk(compressed.numel() * 32 == R * M * K, "compressed.numel() * 32 != R * m * k")
    _chk(codebook.size(0) == 1 << (9 + 1), "codebook must have 1 << (S + V) = 1024 entries")
    xc = x.contiguous().view(-1)
    with torch.cuda.device(x.device):
        rc = _lib.lib().gq_qtip_matvec(out.data_ptr(), compressed.data_ptr(), xc.data_ptr(), codebook.data_ptr(), M, K, R,
                                       _lib.current_stream_ptr())
    _lib.check(rc, "decompress_matvec")


class _QtipKernels(types.ModuleType):
    _pat = re.compile(r"^decompress_matvec_16_9_(\d+)_1_(\d+)_1_(\d+)$")

    def __getattr__(self, name):
        m = self._pat.match(name)
        if not m:
            raise AttributeError(name)
        R, M, K = (int(g) for g in m.groups())

        def fn(out, compressed, x, codebook, _R=R, _M=M, _K=K):
            return decompress_matvec(out, compressed, x, codebook, _R, _M, _K)

        fn.__name__ = name
        setattr(self, name, fn)
        return fn


qtip_kernels = _QtipKernels("qtip_kernels")

# --------------------------------------------------------------------------------------------- torch ops
_registered = set()


def quip_lib_op(m, k, R, n=1):
    """returns torch.ops.quip_lib.decompress_matvec_qtip_{m}_{n}_{k}_{R}, registering it on first use"""
    name = f"decompress_matvec_qtip_{m}_{n}_{k}_{R}"
    if name not in _registered:
        torch.library.define(f"quip_lib::{name}", "(Tensor compressed, Tensor x, Tensor codebook) -> Tensor")

        def _fake(compressed, x, codebook, _m=m):
            return torch.zeros(1, _m, dtype=torch.float32, device=x.device)

        def _impl(compressed, x, codebook, _m=m, _k=k, _R=R):
            out = torch.zeros((_m, 1), dtype=torch.float32, device=x.device)
            fn = getattr(qtip_kernels, f"decompress_matvec_16_9_{_R}_1_{_m}_1_{_k}")
            fn(out, compressed.reshape(-1).view(torch.int32), x.to(torch.float16).T, codebook.reshape(-1))
            return out.T

        torch.library.register_fake(f"quip_lib::{name}")(_fake)
        torch.library.impl(f"quip_lib::{name}", "cuda")(_impl)
        _registered.add(name)
    return getattr(torch.ops.quip_lib, name)


def register_model_shapes():
    """the (m, k) of every linear of every model in the table, un-fused and fused, R in 2..4"""
    from .model import transformer_configs
    for cfg in transformer_configs.values():
        d, i = cfg["dim"], cfg["intermediate_size"]
        hd = d // cfg["n_head"]
        kv = cfg["n_local_heads"] * hd
        for m, k in {(d, d), (kv, d), (i, d), (d, i), (d + 2 * kv, d), (2 * i, d)}:
            for R in (2, 3, 4):
                quip_lib_op(m, k, R)


if "hadamard::hadamard" not in _registered:
    torch.library.define("hadamard::hadamard", "(Tensor x, float scale) -> Tensor")

    @torch.library.register_fake("hadamard::hadamard")
    def _hadamard_abstract(x, scale):
        return torch.empty_like(x)  # (a new tensor, like the implementation: returning x itself claims an alias -- opcheck)

    @torch.library.impl("hadamard::hadamard", "cuda")
    def _hadamard_cuda(x, scale):
        n = x.shape[-1]
        xc = x.contiguous().float()
        y = torch.empty_like(xc)
        with torch.cuda.device(x.device):
            rc = _lib.lib().gq_hadamard(xc.data_ptr(), y.data_ptr(), xc.numel() // n, n, float(scale), _lib.current_stream_ptr())
        _lib.check(rc, "hadamard")
        return y.to(x.dtype)

    _registered.add("hadamard::hadamard")

# --------------------------------------------------------------------------------------------- Hadamard factors
_HAD_FACTORS = (172, 156, 140, 124, 116, 108, 60, 52, 36, 28, 20, 12)  # order of matmul_had.py:13-67
_tables = None


def _is_pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def _load_hadamard_tables():
    """the reference's factor matrices (matmul_had.py:133-..., get_had12 .. get_had172) from the packed data file shipped
    with the package (tools/make_hadamard_tables.py: K x K signs, np.packbits row-major); an .npz named by
    GQ_HADAMARD_TABLES (keys "had{K}", +-1 matrices) overrides / extends it"""
    tabs = {}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "hadamard_factors.npz")
    if os.path.exists(path):
        with np.load(path) as z:
            for key in z.files:
                K = int(key[3:])
                bits = np.unpackbits(z[key])[:K * K].reshape(K, K)
                tabs[key] = (bits.astype(np.int8) * 2 - 1)
    user = os.environ.get("GQ_HADAMARD_TABLES")
    if user and os.path.exists(user):
        tabs.update(dict(np.load(user)))
    return tabs


def get_hadK(n, transpose=False):
    """(hadK, K) as matmul_had.py:13-67: the first factor K in the reference's order with n % K == 0 and n / K a power
    of two; K == 1 and hadK None for a power of two."""
    global _tables
    if _is_pow2(n):
        return None, 1
    for K in _HAD_FACTORS:
        if n % K == 0 and _is_pow2(n // K):
            if _tables is None:
                _tables = _load_hadamard_tables()
            if f"had{K}" not in _tables:
                raise NotImplementedError(f"no Hadamard factor table of order {K} (n = {n}) in guidedquant_amd/data/"
                                          "hadamard_factors.npz or GQ_HADAMARD_TABLES")
            h = torch.from_numpy(_tables[f"had{K}"].astype(np.float32))
            return (h.T if transpose else h), K
    raise AssertionError(f"no Hadamard factorisation for n = {n}")


def matmul_hadU_cuda(X, hadK, K, transpose=False):
    n = X.shape[-1]
    if K == 1:
        return torch.ops.hadamard.hadamard(X.contiguous(), n**(-0.5))
    if transpose:
        hadK = hadK.T.contiguous()
    inp = X.float().view(-1, K, n // K)
    inp = torch.ops.hadamard.hadamard(inp.contiguous(), n**(-0.5))
    inp = hadK.to(inp.device).to(inp.dtype) @ inp
    return inp.to(X.device).to(X.dtype).reshape(X.shape)


def matmul_hadUt_cuda(X, hadK, K):
    return matmul_hadU_cuda(X, hadK, K, transpose=True)


def has_kernel(decode_mode, L, K, V, tlut_bits, td_x, td_y):
    """inference/lib/utils/kernel_check.py:1-14"""
    return (decode_mode == 'quantlut_sym' and L == 16 and V == 2 and 2 <= K <= 4 and tlut_bits == 9 and td_x == 16
            and td_y == 16)


# --------------------------------------------------------------------------------------------- modules
class BitshiftLinear(nn.Module):
    """eval-mode forward of inference/lib/codebook/bitshift.py:415-472 for the kernel-served configuration
    (quantlut_sym, L=16, V=2, K in 2..4, tlut_bits=9, 16x16 tiles)."""

    def __init__(self, td_x, td_y, L, K, V, tlut_bits, decode_mode, dtype=torch.float16, tlut=None, has_kernel=False):
        super().__init__()
        _chk(has_kernel, "only the kernel-served QTIP configuration (kernel_check.py) is implemented")
        self.td_x, self.td_y, self.L, self.K, self.V = td_x, td_y, L, K, V
        self.tlut_bits = tlut_bits
        self.tlut = tlut
        self.internal_dtype = dtype
        self.has_kernel = has_kernel
        self.scale = 32

    def forward(self, input, trellis, SU, SV, had_left, had_right, K_left, K_right, rcp, tp_rank, mode='eval', **kwargs):
        n, m = len(SU), len(SV)
        x = input.view(-1, n).to(torch.float32)
        x = x * SU
        bs = x.shape[0]
        if rcp == 1:
            x = matmul_hadUt_cuda(x.reshape(-1, n // tp_rank), had_left, K_left).reshape(x.shape) / self.scale
        else:
            x = matmul_hadUt_cuda(x, had_left, K_left) / self.scale
        if bs == 1:
            x = quip_lib_op(m, n, self.K)(trellis, x, self.tlut)
        else:
            # batched: row-by-row through the same kernel (the reference decodes + matmuls; same arithmetic class)
            x = torch.cat([quip_lib_op(m, n, self.K)(trellis, x[i:i + 1], self.tlut) for i in range(bs)], dim=0)
        if rcp == 2:
            x = matmul_hadU_cuda(x.reshape(-1, m // tp_rank), had_right, K_right).reshape(x.shape)
        else:
            x = matmul_hadU_cuda(x, had_right, K_right)
        x = x.to(SV.device) * (SV * self.scale)
        return x.view(*input.shape[:-1], m).to(input.dtype)


class QuantizedLinear(nn.Module):
    """qtip/lib/linear/quantized_linear.py:12-153 (eval mode)."""

    def __init__(self, in_features, out_features, td_x, td_y, L, K, V, tlut_bits, decode_mode, bias=False,
                 dtype=torch.float16, mode='eval', use_prev_kernel=True, grad_ckpt=False, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.td_x, self.td_y, self.L, self.K, self.V = td_x, td_y, L, K, V
        self.tlut_bits, self.decode_mode = tlut_bits, decode_mode
        self.register_buffer('rcp', torch.tensor(0))
        self.register_buffer('tp_rank', torch.tensor(8))
        self.dtype = dtype
        self.register_buffer('trellis', torch.zeros((out_features // td_x) * (in_features // td_y),
                                                    math.ceil((td_x * td_y) * K / 16), dtype=torch.int16, device=device))
        if decode_mode in ['lut', 'quantlut', 'quantlut_sym']:
            self.tlut = nn.Parameter(torch.zeros(2**tlut_bits, V, dtype=torch.float16, device=device), requires_grad=False)
        else:
            self.tlut = None
        if bias:
            self.register_buffer('bias', torch.ones(out_features, device=device))
        else:
            self.bias = None
        self.register_buffer("SU", torch.ones(in_features, dtype=self.dtype, device=device))
        self.register_buffer("SV", torch.ones(out_features, dtype=torch.float32, device=device))
        had_left, K_left = get_hadK(in_features)
        had_right, K_right = get_hadK(out_features)
        self.register_buffer('had_left', had_left, persistent=False)
        self.register_buffer('had_right', had_right, persistent=False)
        self.K_left, self.K_right = K_left, K_right
        self.mode = mode
        self.has_kernel = has_kernel(decode_mode, L, K, V, tlut_bits, td_x, td_y)
        self.codebook_class = None

    def forward(self, input):
        if self.codebook_class is None:
            self.codebook_class = BitshiftLinear(self.td_x, self.td_y, self.L, self.K, self.V, self.tlut_bits,
                                                 self.decode_mode, dtype=self.dtype, tlut=self.tlut,
                                                 has_kernel=self.has_kernel)
            self._rcp = int(self.rcp.item())
            self._tp = int(self.tp_rank.item())
        result = self.codebook_class(input, self.trellis, self.SU, self.SV, self.had_left, self.had_right, self.K_left,
                                     self.K_right, self._rcp, self._tp, mode=self.mode) + 0
        if self.bias is not None:
            return result + self.bias
        return result


QTIPLinear = QuantizedLinear  # name used by inference/generate.py:22-23
sys.modules.setdefault("guidedquant_amd.qtip_kernels", qtip_kernels)
