"""Python module with the reference's `ap_gemv` extension surface (inference/ap_gemv/bindings.cpp:12-17):

    anyprec_gemv(input, output, qweight, lut, bitwidth) -> None
    anyprec_dequant(qweight, lut, bitwidth) -> Tensor[N, K] fp16
    lutgemm_gemv(input, output, q_weight, alpha, q_bias, bitwidth, group_size) -> None

Same argument order, same validation (messages follow inference/ap_gemv/gemv.cu:64-90,180-211; failures raise
RuntimeError like TORCH_CHECK), work enqueued on the current torch stream of the tensors' device
(gemv.cu:103-105).  The arithmetic runs in libgq_hip.so through the C ABI; without the library every call raises.

Device dispatch (BASELINE.json configs[0], "CPU reference APLinear path via generate.py"): when ALL tensors of a call live
in host memory, `anyprec_gemv` / `anyprec_dequant` run the library's CPU twins (`gq_anyprec_gemv_cpu` /
`gq_anyprec_dequant_cpu`, csrc/cpu_twins.cpp: packed planes read directly, fp32 accumulation) -- the reference has no CPU
kernel and raises here.  Mixed placements raise with the reference's messages.

One deliberate relaxation: `qweight.size(0) >= bitwidth` is accepted (the reference demands equality,
gemv.cu:76) because the kernel's plane stride is N*K/32 regardless (anyprec.cu:446) -- this lets a
multi-precision parent tensor be used without slicing (AnyPrecisionLinear.py:73 passes the un-pruned tensor).
"""
import torch

from . import _lib


def _chk(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _dev_guard(t):
    return torch.cuda.device(t.device)


def anyprec_gemv(input, output, qweight, lut, bitwidth):
    bitwidth = int(bitwidth)
    _chk(2 <= bitwidth <= 8, "Bitwidth must be between 2 and 8.")
    _chk(input.dtype == lut.dtype and input.dtype == output.dtype,
         "Mismatched data types between input, lut, and output tensors.")
    _chk(qweight.dtype == torch.int32, "qweight tensor must be of type int.")
    _chk(input.dim() == 3, "input tensor must be of shape (batch_size, seq_len, hidden_size).")
    _chk(output.dim() == 3, "output tensor must be of shape (batch_size, seq_len, hidden_size).")
    _chk(lut.dim() == 2 and lut.size(1) == (1 << bitwidth) and lut.size(0) == output.size(2),
         f"lut tensor must be of shape (output_feat, 2 ** bitwidth). Expected ({output.size(2)}, {1 << bitwidth}), "
         f"got ({', '.join(str(s) for s in lut.shape)}).")
    _chk(qweight.dim() == 3 and qweight.size(0) >= bitwidth and qweight.size(2) == input.size(2) // 32
         and qweight.size(1) == output.size(2),
         f"qweight tensor must be of shape (bitwidth, output_feat, input_feat / 32). Expected ({bitwidth}, "
         f"{output.size(2)}, {input.size(2) // 32}), got ({', '.join(str(s) for s in qweight.shape)}).")
    _chk(input.size(1) == 1, "Only sequence length of 1 is supported.")
    _chk(output.size(1) == 1, "Only sequence length of 1 is supported.")
    on_cpu = not (input.is_cuda or output.is_cuda or qweight.is_cuda or lut.is_cuda)
    if not on_cpu:
        _chk(input.is_cuda and output.is_cuda, "input and output tensors must be on GPU.")
        _chk(qweight.is_cuda and lut.is_cuda, "qweight and lut tensors must be on GPU.")
    _chk(input.is_contiguous(), "input tensor must be contiguous.")
    _chk(output.is_contiguous(), "output tensor must be contiguous.")
    _chk(qweight.is_contiguous(), "qweight tensor must be contiguous.")
    _chk(lut.is_contiguous(), "lut tensor must be contiguous.")
    _chk(input.dtype == torch.float16, "only float16 is supported (the reference kernels are hard-wired to half).")
    _chk(input.size(0) == output.size(0), "input and output batch sizes differ.")
    M, K, N = input.size(0), input.size(2), output.size(2)
    _chk(1 <= M <= 8, "batch size must be between 1 and 8.")
    _chk(K % 32 == 0, "input_feat must be a multiple of 32.")
    if on_cpu:
        rc = _lib.lib().gq_anyprec_gemv_cpu(input.data_ptr(), output.data_ptr(), qweight.data_ptr(), lut.data_ptr(), M, N, K,
                                            bitwidth, 0, torch.get_num_threads())
        _lib.check(rc, "anyprec_gemv (cpu)")
        return
    with _dev_guard(qweight):
        rc = _lib.lib().gq_anyprec_gemv(input.data_ptr(), output.data_ptr(), qweight.data_ptr(), lut.data_ptr(), M, N, K,
                                        bitwidth, 0, _lib.current_stream_ptr())
    _lib.check(rc, "anyprec_gemv")


def anyprec_dequant(qweight, lut, bitwidth):
    bitwidth = int(bitwidth)
    _chk(2 <= bitwidth <= 8, "Bitwidth must be between 2 and 8.")
    _chk(qweight.dtype == torch.int32 and qweight.dim() == 3, "qweight tensor must be int32 of shape (bitwidth, N, K/32).")
    _chk(qweight.size(0) >= bitwidth, "qweight holds fewer planes than bitwidth.")
    _chk(lut.dtype == torch.float16 and lut.dim() == 2 and lut.size(0) == qweight.size(1)
         and lut.size(1) == (1 << bitwidth), "lut tensor must be float16 of shape (output_feat, 2 ** bitwidth).")
    on_cpu = not (qweight.is_cuda or lut.is_cuda)
    if not on_cpu:
        _chk(qweight.is_cuda and lut.is_cuda, "qweight and lut tensors must be on GPU.")
    _chk(qweight.is_contiguous() and lut.is_contiguous(), "qweight and lut tensors must be contiguous.")
    N, K = qweight.size(1), qweight.size(2) * 32
    weight = torch.empty((N, K), dtype=torch.float16, device=qweight.device)  # gemv.cu:121-122
    if on_cpu:
        rc = _lib.lib().gq_anyprec_dequant_cpu(qweight.data_ptr(), lut.data_ptr(), weight.data_ptr(), N, K, bitwidth,
                                               torch.get_num_threads())
        _lib.check(rc, "anyprec_dequant (cpu)")
        return weight
    with _dev_guard(qweight):
        rc = _lib.lib().gq_anyprec_dequant(qweight.data_ptr(), lut.data_ptr(), weight.data_ptr(), N, K, bitwidth,
                                           _lib.current_stream_ptr())
    _lib.check(rc, "anyprec_dequant")
    return weight


def anyprec_gemm_supported(x, qweight, bitwidth):
    """True when a seq_len > 1 call should take the fused prefill GEMM (gq_anyprec_gemm_ws).  It serves GPU fp16 tensors, 2..4
    bits, K % 64 == 0.  GQ_PREFILL_FUSED=1 sends every such call to it, =0 none; by default ("auto") only the calls it was
    measured faster on than the reference's two steps (dequantise + hipBLASLt GEMM; profiles/r03_prefill_gemm.txt, 8B shapes):
    every matrix up to 160 rows (S = 128, 2-bit: wqkv 25 vs 44 us, wo 21 vs 37, gate/up 50 vs 97, down 36 vs 83 -- K split over
    fp32 partial sums where the grid is short), up to 640 rows every matrix at 2 / 3 bits and the large ones (gate/up, down) at 4 bits
    (S = 512, 2-bit: 48 vs 52, 38 vs 41, 125 vs 158, 84 vs 101); longer prompts keep the two steps (S = 2048: the fused kernel
    reaches 0.35-0.43 of the fp16 MFMA peak against ~0.5 for hipBLASLt and loses by 12-30 %)."""
    import os
    if not (x.is_cuda and qweight.is_cuda and x.dtype == torch.float16 and 2 <= int(bitwidth) <= 4 and x.shape[-1] % 64 == 0):
        return False
    mode = os.environ.get("GQ_PREFILL_FUSED", "auto")
    if mode in ("0", "1"):
        return mode == "1"
    rows = x.numel() // x.shape[-1]
    if rows <= 160:
        return True
    return rows <= 640 and (int(bitwidth) <= 3 or qweight.size(1) * x.shape[-1] >= 50_000_000)


def anyprec_gemm(x, qweight, lut, bitwidth):
    """x fp16 [..., K] (more than one row) -> fp16 [..., N] = x @ dequant(qweight, lut).T with the dequantisation fused into
    the MFMA loop: the seq_len > 1 branch of APLinear.forward (inference/APLinear.py:35-50) without the dense copy of W."""
    bitwidth = int(bitwidth)
    _chk(2 <= bitwidth <= 4, "fused prefill GEMM serves bit widths 2..4.")
    _chk(qweight.dtype == torch.int32 and qweight.dim() == 3 and qweight.size(0) >= bitwidth, "qweight tensor must be int32 of shape (>= bitwidth, N, K/32).")
    _chk(lut.dtype == torch.float16 and lut.dim() == 2 and lut.size(0) == qweight.size(1) and lut.size(1) == (1 << bitwidth),
         "lut tensor must be float16 of shape (output_feat, 2 ** bitwidth).")
    _chk(x.dtype == torch.float16 and x.is_cuda and qweight.is_cuda and lut.is_cuda, "x, qweight and lut must be float16 / int32 tensors on the GPU.")
    _chk(qweight.is_contiguous() and lut.is_contiguous(), "qweight and lut tensors must be contiguous.")
    N, K = qweight.size(1), qweight.size(2) * 32
    _chk(x.shape[-1] == K, "input_feat mismatch.")
    x2 = x.reshape(-1, K).contiguous()
    out = torch.empty((x2.size(0), N), dtype=torch.float16, device=x.device)
    with _dev_guard(qweight):
        # short grids split K over fp32 partial sums (gq_anyprec_gemm_ws): the workspace comes from torch's caching allocator
        wsb = int(_lib.lib().gq_anyprec_gemm_ws_bytes(x2.size(0), N, K, bitwidth))
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
        rc = _lib.lib().gq_anyprec_gemm_ws(x2.data_ptr(), out.data_ptr(), qweight.data_ptr(), lut.data_ptr(), x2.size(0), N, K, bitwidth,
                                           ws.data_ptr() if wsb else None, wsb, _lib.current_stream_ptr())
    _lib.check(rc, "anyprec_gemm")
    return out.reshape(*x.shape[:-1], N)


def anyprec_pack(codes, bitwidth):
    """codes uint8 [N, K] on the GPU -> qweight int32 [bitwidth, N, K // 32], the packed format of
    any_precision/quantization/pack.py:304-321 (bit-identical to the host packer guidedquant_amd.pack.pack_codes)"""
    bitwidth = int(bitwidth)
    _chk(1 <= bitwidth <= 8, "Bitwidth must be between 1 and 8.")
    _chk(codes.dtype == torch.uint8 and codes.dim() == 2 and codes.is_cuda and codes.is_contiguous(), "codes must be a contiguous uint8 [N, K] tensor on the GPU.")
    N, K = codes.shape
    _chk(K % 32 == 0, "K must be a multiple of 32.")
    q = torch.empty((bitwidth, N, K // 32), dtype=torch.int32, device=codes.device)
    with _dev_guard(codes):
        rc = _lib.lib().gq_anyprec_pack(codes.data_ptr(), q.data_ptr(), N, K, bitwidth, _lib.current_stream_ptr())
    _lib.check(rc, "anyprec_pack")
    return q


def lutgemm_gemv(input, output, q_weight, alpha, q_bias, bitwidth, group_size):
    bitwidth, group_size = int(bitwidth), int(group_size)
    _chk(1 <= bitwidth <= 8, "Bitwidth must be between 1 and 8.")
    _chk(input.dtype == alpha.dtype and input.dtype == q_bias.dtype and input.dtype == output.dtype,
         "Mismatched data types between input, alpha, q_bias, and output tensors.")
    _chk(input.dim() == 3, "input tensor must be of shape (batch_size, seq_len, input_feat).")
    _chk(output.dim() == 3, "output tensor must be of shape (batch_size, seq_len, output_feat).")
    _chk(input.size(0) == 1, "Batch size must be 1 for input tensor.")
    _chk(input.size(1) == 1, "Sequence length must be 1 for input tensor.")
    _chk(output.size(0) == 1, "Batch size must be 1 for output tensor.")
    _chk(output.size(1) == 1, "Sequence length must be 1 for output tensor.")
    _chk(input.is_cuda and output.is_cuda, "input and output tensors must be on GPU.")
    _chk(input.is_contiguous(), "input tensor must be contiguous.")
    _chk(output.is_contiguous(), "output tensor must be contiguous.")
    _chk(q_weight.is_contiguous(), "q_weight tensor must be contiguous.")
    _chk(alpha.is_contiguous(), "alpha tensor must be contiguous.")
    _chk(q_bias.is_contiguous(), "q_bias tensor must be contiguous.")
    K, N = input.size(2), output.size(2)
    _chk(group_size > 0 and K % group_size == 0, "group_size must divide input_feat.")
    ng = K // group_size
    _chk(q_weight.dim() == 3 and q_weight.size(0) == K // 32 and q_weight.size(1) == bitwidth and q_weight.size(2) == N,
         f"q_weight tensor must be of shape (input_feat / 32, bitwidth, output_feat). Expected ({K // 32}, {bitwidth}, "
         f"{N}), got ({', '.join(str(s) for s in q_weight.shape)}).")
    _chk(alpha.dim() == 3 and alpha.size(0) == ng and alpha.size(1) == bitwidth and alpha.size(2) == N,
         f"alpha tensor must be of shape (num_groups, bitwidth, output_feat). Expected ({ng}, {bitwidth}, {N}), "
         f"got ({', '.join(str(s) for s in alpha.shape)}).")
    _chk(q_bias.dim() == 2 and q_bias.size(0) == ng and q_bias.size(1) == N,
         "q_bias tensor must be of shape (num_groups, output_feat).")
    _chk(input.dtype == torch.float16 and q_weight.dtype == torch.int32, "only float16 / int32 are supported.")
    with _dev_guard(q_weight):
        ws = _lutgemm_workspace(input.device, K)
        if ws is not None:  # tables built once per call into the scratch buffer, then the GEMV (bit-identical, ~4x faster)
            rc = _lib.lib().gq_lutgemm_gemv_ws(input.data_ptr(), output.data_ptr(), q_weight.data_ptr(), alpha.data_ptr(),
                                               q_bias.data_ptr(), N, K, bitwidth, group_size, ws.data_ptr(), ws.numel(),
                                               _lib.current_stream_ptr())
        else:
            rc = _lib.lib().gq_lutgemm_gemv(input.data_ptr(), output.data_ptr(), q_weight.data_ptr(), alpha.data_ptr(),
                                            q_bias.data_ptr(), N, K, bitwidth, group_size, _lib.current_stream_ptr())
    _lib.check(rc, "lutgemm_gemv")


_LG_WS = {}
_LG_WS_RETIRED = []


def _lutgemm_workspace(device, K):
    """per-device scratch for the sign-sum tables (64 bytes per input feature); the library never allocates.  None while a
    graph is being captured and no buffer exists yet (the single-kernel form is used for that call)."""
    import os
    if os.environ.get("GQ_LUTGEMM_WS", "1") == "0":
        return None
    # one buffer per (device, stream): the tables kernel and the GEMV that reads them are ordered by the stream only
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(device).cuda_stream))
    ws = _LG_WS.get(key)
    if ws is None or ws.numel() < K * 64:
        if torch.cuda.is_current_stream_capturing():
            return None
        if ws is not None:
            _LG_WS_RETIRED.append(ws)  # a captured graph may still replay launches that point at the smaller buffer
        ws = torch.empty(max(K, 16384) * 64, dtype=torch.uint8, device=device)
        _LG_WS[key] = ws
    return ws
