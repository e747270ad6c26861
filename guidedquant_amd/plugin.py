"""Operator boundary, mirroring inference/plugin.py:7-26 of the reference: the torch.library custom ops

    plugin::anyprec_gemv(Tensor x, Tensor q_weight, Tensor lut, Tensor(a!) output, int bitwidth) -> ()
    plugin::lutgemm_gemv(Tensor x, Tensor(a!) output, Tensor q_weight, Tensor alpha, Tensor q_bias,
                         int bitwidth, int group_size) -> ()

with fake (meta) implementations returning None so torch.compile(fullgraph=True) can trace them, and the plain
function anyprec_dequant.  NB the argument-order flip between the op (x, q_weight, lut, output, bitwidth) and
the extension (x, output, q_weight, lut, bitwidth) is the reference's (plugin.py:8-9) and is kept.

Single registration point: the reference registers plugin::anyprec_gemv twice (plugin.py:7 and
any_precision/modules/AnyPrecisionLinear.py:9); here AnyPrecisionLinear imports it from this module.
"""
import torch

from . import ap_gemv


@torch.library.custom_op("plugin::anyprec_gemv", mutates_args={"output"})
def anyprec_gemv(x: torch.Tensor, q_weight: torch.Tensor, lut: torch.Tensor, output: torch.Tensor,
                 bitwidth: int) -> None:
    ap_gemv.anyprec_gemv(x, output, q_weight, lut, bitwidth)


@anyprec_gemv.register_fake
def _(x, q_weight, lut, output, bitwidth):
    return None


def anyprec_dequant(q_weight: torch.Tensor, lut: torch.Tensor, bitwidth: int) -> torch.Tensor:
    weight = ap_gemv.anyprec_dequant(q_weight, lut, bitwidth)
    return weight


@torch.library.custom_op("plugin::lutgemm_gemv", mutates_args={"output"})
def lutgemm_gemv(x: torch.Tensor, output: torch.Tensor, q_weight: torch.Tensor, alpha: torch.Tensor,
                 q_bias: torch.Tensor, bitwidth: int, group_size: int) -> None:
    ap_gemv.lutgemm_gemv(x, output, q_weight, alpha, q_bias, bitwidth, group_size)


@lutgemm_gemv.register_fake
def _(x, output, q_weight, alpha, q_bias, bitwidth, group_size):
    return None
