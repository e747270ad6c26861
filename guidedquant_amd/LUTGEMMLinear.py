"""LUTGEMMLinear -- inference/LUTGEMMLinear.py:5-81 of the reference on MI355X.

Buffers `qweight int32[K/32, bitwidth, N]`, `alpha fp16[K/group, bitwidth, N]`, `q_bias fp16[K/group, N]`;
forward is bs=1, seq=1 only; the kernel ACCUMULATES into `self.output`, so it is zeroed first exactly as
LUTGEMMLinear.py:74 does.  `group_size == -1` means one group over K (LUTGEMMLinear.py:12-15).
"""
import torch
import torch.nn as nn

from .plugin import lutgemm_gemv


class LUTGEMMLinear(nn.Module):

    def __init__(self, in_features, out_features, bitwidth, group_size, bias=False, dtype=torch.half, device="cuda"):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.bitwidth = bitwidth
        self.group_size = in_features if group_size == -1 else group_size
        self.dtype = dtype
        self.register_buffer("qweight",
                             torch.empty((in_features // 32, bitwidth, out_features), dtype=torch.int32, device=device))
        self.register_buffer(
            "alpha", torch.empty((in_features // self.group_size, bitwidth, out_features), dtype=self.dtype, device=device))
        self.register_buffer("q_bias",
                             torch.empty((in_features // self.group_size, out_features), dtype=self.dtype, device=device))
        if bias:
            self.register_buffer("bias", torch.empty((out_features, ), dtype=self.dtype, device=device))
        else:
            self.bias = None
        self.output = torch.zeros((1, 1, out_features), dtype=self.dtype, device=device)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.output = fn(self.output)
        return self

    def forward(self, x, **kwargs):
        assert (x.shape[0] == 1)
        assert (x.shape[1] == 1)
        self.output.zero_()
        lutgemm_gemv(x, self.output, self.qweight, self.alpha, self.q_bias, self.bitwidth, self.group_size)
        if self.bias is not None:
            self.output += self.bias
        return self.output
