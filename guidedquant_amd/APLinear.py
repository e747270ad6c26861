"""APLinear -- the gpt-fast-path module of the reference (inference/APLinear.py:6-60), on MI355X.

Same constructor, buffer names/shapes/dtypes (`qweight int32[bitwidth, N, K/32]`, `lut fp16[N, 2**bitwidth]`,
optional `bias`), same dispatch: seq_len > 1 -> dequantise + matmul (APLinear.py:35-50), seq_len == 1 ->
plugin::anyprec_gemv into the persistent `self.output[1,1,N]`, which is returned BY REFERENCE every call
(APLinear.py:52-60; callers consume it before the next call, model.py:211,261).

Differences, all additive: a `device` argument (the reference hard-codes 'cuda', APLinear.py:17,22,33), and the
`output.zero_()` of APLinear.py:53 is dropped -- the GEMV writes every element (anyprec.cu:532-541 stores, it
does not accumulate), so zeroing only cost a kernel launch per linear per token.
"""
import torch
import torch.nn as nn

from . import ap_gemv
from .plugin import anyprec_dequant, anyprec_gemv


class APLinear(nn.Module):

    def __init__(self, in_features, out_features, bitwidth, bias=False, dtype=torch.half, device="cuda"):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.bitwidth = bitwidth
        self.dtype = dtype
        self.register_buffer("qweight",
                             torch.empty((bitwidth, out_features, in_features // 32), dtype=torch.int32, device=device))
        self.register_buffer("lut", torch.empty((out_features, 2**bitwidth), dtype=self.dtype, device=device))
        if bias:
            self.register_buffer("bias", torch.empty((out_features, ), dtype=self.dtype, device=device))
        else:
            self.bias = None
        self.output = torch.zeros((1, 1, self.out_features), dtype=self.dtype, device=device)

    def _apply(self, fn, *a, **k):
        # keep the persistent (non-buffer) output tensor on the module's device, as `model.to(device)` expects
        super()._apply(fn, *a, **k)
        self.output = fn(self.output)
        return self

    def gemm(self, x):
        # prefill rows: the MFMA GEMM with the dequantisation fused in (no dense copy of W); shapes / devices it does not
        # serve take the reference's two steps, dequantise + matmul (APLinear.py:35-38)
        if ap_gemv.anyprec_gemm_supported(x, self.qweight, self.bitwidth):
            return ap_gemv.anyprec_gemm(x, self.qweight, self.lut, self.bitwidth)
        weight = anyprec_dequant(self.qweight, self.lut, self.bitwidth)
        return torch.matmul(x, weight.T)

    def forward(self, x, **kwargs):
        assert (x.shape[0] == 1)
        if x.shape[1] > 1:
            output = self.gemm(x)
            if self.bias is not None:
                output += self.bias
            return output
        anyprec_gemv(x, self.qweight, self.lut, self.output, self.bitwidth)
        if self.bias is not None:
            self.output += self.bias
        return self.output
