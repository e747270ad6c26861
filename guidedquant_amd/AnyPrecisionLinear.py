"""AnyPrecisionLinear -- the HF-path module (any_precision/modules/AnyPrecisionLinear.py:17-89) on MI355X.

Multi-precision: one `qweight int32[max(supported_bits), N, K/32]` parent tensor and one `lut{bit}` buffer per
supported precision; `forward(x, precision=b)` picks the LUT and uses the first b planes.  rows > 1 ->
dequant + matmul (:69-71); rows == 1 -> plugin::anyprec_gemv into the persistent output (:73-74); bias; clamp to
+-(1 - 5e-3) * finfo.max (:79).  The op is imported from .plugin (single registration point).
"""
import torch
import torch.nn as nn

from . import ap_gemv
from .plugin import anyprec_gemv


class AnyPrecisionLinear(nn.Module):

    def __init__(self, in_features, out_features, supported_bits, bias=True, precisions=None, device=None, dtype=None):
        super().__init__()
        if precisions is None:
            precisions = supported_bits
        if not isinstance(precisions, list):
            raise RuntimeError('supported_bits must be a list of integers.')
        if dtype is not None and dtype != torch.float16:
            raise RuntimeError('Only float16 is supported for now.')
        self.in_features = in_features
        self.out_features = out_features
        self.precisions = precisions
        self.precision = max(self.precisions)
        self.supported_bits = supported_bits
        self.register_buffer(
            'qweight', torch.empty((max(supported_bits), out_features, in_features // 32), dtype=torch.int32, device=device))
        for bit in supported_bits:
            self.register_buffer(f'lut{bit}', torch.empty((out_features, 2**bit), dtype=dtype, device=device))
        if bias:
            self.register_buffer("bias", torch.empty((out_features, ), dtype=dtype, device=device))
        else:
            self.bias = None
        self.output = torch.zeros((1, 1, self.out_features), dtype=torch.float16,
                                  device=device if device is not None else 'cuda')

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        out = fn(self.output)
        self.output = out if out.dtype == torch.float16 else out.to(torch.float16)
        return self

    def prune_precisions(self):
        self.qweight = self.qweight[:max(self.precisions)].contiguous()
        for bit in self.supported_bits:
            if bit not in self.precisions:
                delattr(self, f'lut{bit}')

    def forward(self, x, **kwargs):
        if self.qweight.numel() == 0 and getattr(self, "_gq_restore", None) is not None:
            self._gq_restore()  # the planes live in the fused decode model (AnyPrecisionForCausalLM.native_decoder): take them back
        if 'precision' in kwargs:
            w_bits = kwargs['precision']
        else:
            w_bits = self.precision
        if x.numel() // x.shape[-1] > 1 and ap_gemv.anyprec_gemm_supported(x, self.qweight, w_bits):
            x = ap_gemv.anyprec_gemm(x, self.qweight, self._buffers[f'lut{w_bits}'].to(torch.float16), w_bits)
        elif x.numel() // x.shape[-1] > 1:
            weight = ap_gemv.anyprec_dequant(self.qweight, self._buffers[f'lut{w_bits}'].to(torch.float16),
                                             w_bits).to(x.dtype)
            x = torch.matmul(x, weight.T)
        else:
            anyprec_gemv(x.to(torch.float16).reshape(1, 1, -1), self.qweight,
                         self._buffers[f'lut{w_bits}'].to(torch.float16), self.output, w_bits)
            x = self.output.to(x.dtype).reshape(*x.shape[:-1], self.out_features)
        if self.bias is not None:
            x += self.bias
        return x.clamp_(torch.finfo(x.dtype).min * (1.0 - 5e-3), torch.finfo(x.dtype).max * (1.0 - 5e-3))

    def set_precision(self, precision):
        if precision not in self.precisions:
            raise RuntimeError(f"{self.precisions}-bit precisions are supported but {precision}-bit was specified.")
        self.precision = precision

    def extra_repr(self) -> str:
        return f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}'
