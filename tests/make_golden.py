"""Generate golden input/output vectors by IMPORTING the reference's own Python.

Runs only in the authoring container (needs /root/reference); the resulting
.npz files under tests/golden/ are data (seeded inputs + the reference's
outputs) and are what travels to the GPU box.  No reference source is copied.

    python tests/make_golden.py            # (re)writes tests/golden/*.npz

Sources of truth used:
  * any_precision/quantization/pack.py   pack_single_weight / unpack_single_weight
    (imported by file path with a no-op `numba.njit` stub, numba is absent here)
  * any_precision/quantization/finetune_utils.py   _dequantize_weight
  * inference/lib/utils/kernel_decompress.py       decode_compressed
  * inference/lib/codebook/bitshift.py             quantlut_sym / bitshift_codebook
  * inference/lib/utils/matmul_had.py              matmul_hadU / matmul_hadUt / get_hadK
  * inference/sqllm_llama_convert_fuse.py, inference/qtip_convert_no_fuse.py   the checkpoint converters, RUN as the scripts
    they are (subprocess, on a seeded tiny checkpoint written to a temporary directory)
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("GQ_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _stub_numba():
    m = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    m.njit = njit
    sys.modules["numba"] = m


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_ap():
    import torch
    _stub_numba()
    pack = _load(os.path.join(REF, "any_precision/quantization/pack.py"), "ref_pack")
    fu = _load(os.path.join(REF, "any_precision/quantization/finetune_utils.py"), "ref_finetune_utils")
    cases = [(2, 8, 96), (2, 8, 1024), (3, 8, 1056), (4, 4, 4096), (2, 4, 4096), (3, 4, 4096), (8, 4, 2080),
             (4, 4, 11008), (2, 4, 14336), (3, 4, 5120), (5, 4, 1120), (6, 4, 2048), (7, 4, 4128)]
    for bits, N, K in cases:
        rng = np.random.default_rng(1000 * bits + K)
        codes = rng.integers(0, 1 << bits, size=(N, 1, K), dtype=np.uint8)
        qweight = pack.pack_single_weight(torch.from_numpy(codes), bits)  # np.int32 [bits,N,K/32]
        qweight = np.asarray(qweight)
        back = pack.unpack_single_weight(torch.from_numpy(qweight.copy()), bits).numpy()
        assert (back == codes).all()
        lut = np.sort(rng.normal(0, 0.02, size=(N, 1, 1 << bits)).astype(np.float16), axis=-1)
        W = fu._dequantize_weight(torch.from_numpy(codes), torch.from_numpy(lut)).numpy()  # fp16 [N,K]
        x = rng.normal(0, 1, size=(K, )).astype(np.float16)
        y64 = W.astype(np.float64) @ x.astype(np.float64)
        np.savez_compressed(os.path.join(OUT, f"ap_b{bits}_N{N}_K{K}.npz"), bits=bits, codes=codes[:, 0, :],
                            qweight=qweight, lut=lut[:, 0, :], W=W, x=x, y64=y64)
        print("ap", bits, N, K, qweight.shape)


def gen_qtip():
    os.environ["TORCHDYNAMO_DISABLE"] = "1"
    import torch
    sys.path.insert(0, os.path.join(REF, "inference"))
    from lib.codebook import bitshift
    from lib.utils.kernel_decompress import decode_compressed
    for R, (m, k) in [(2, (64, 96)), (3, (64, 96)), (4, (64, 96)), (2, (256, 256)), (3, (32, 512)), (4, (128, 64))]:
        # the reference test recipe: qtip/qtip-kernels/test_decompress_matvec.py:251-272, seed 42 (:305)
        torch.manual_seed(42)
        compressed = torch.randint(torch.iinfo(torch.int32).min, torch.iinfo(torch.int32).max, (R * m * k // 32, ),
                                   dtype=torch.int32)
        tlut = torch.clamp(torch.randn(512, 2) / 16, -1, 1).to(torch.float16)
        x = torch.clamp(torch.randn(k, 1) / 16, -1, 1).to(torch.float16)
        cb = bitshift.bitshift_codebook(L=16, K=R, V=2, tlut_bits=9, decode_mode="quantlut_sym", tlut=tlut)
        lut_expanded = cb.lut.T.contiguous()  # [65536, 2] fp16
        W = decode_compressed(16, 9, R, 1, m, k, compressed.view(torch.int16), lut_expanded)  # [m,k] fp16
        y = (W.float() @ x.float()).double().numpy()
        y64 = W.double().numpy() @ x.double().numpy()
        np.savez_compressed(os.path.join(OUT, f"qtip_R{R}_m{m}_k{k}.npz"), R=R, m=m, k=k,
                            compressed=compressed.numpy(), tlut=tlut.numpy(), x=x.numpy()[:, 0], W=W.numpy(),
                            y64=y64[:, 0], lut_expanded_head=lut_expanded[:64].numpy())
        print("qtip", R, m, k)
    # quantlut_sym expansion itself (bitshift.py:72-80): full 65536x2 table for one seeded tlut
    torch.manual_seed(7)
    tlut = (torch.randn(512, 2) / 16).to(torch.float16)
    full = bitshift.quantlut_sym(tlut, 16, 9)
    np.savez_compressed(os.path.join(OUT, "qtip_quantlut_sym.npz"), tlut=tlut.numpy(), expanded=full.numpy())


def gen_hadamard():
    os.environ["TORCHDYNAMO_DISABLE"] = "1"
    import torch
    sys.path.insert(0, os.path.join(REF, "inference"))
    from lib.utils import matmul_had
    for n in (64, 224, 688, 1024, 4096, 11008, 14336):
        torch.manual_seed(n)
        X = torch.randn(2, n, dtype=torch.float32)
        Y = matmul_had.matmul_hadU(X)
        Yt = matmul_had.matmul_hadUt(X)
        hadK, Kf = matmul_had.get_hadK(n)
        np.savez_compressed(os.path.join(OUT, f"had_n{n}.npz"), n=n, X=X.numpy(), Y=Y.numpy(), Yt=Yt.numpy(), K=Kf,
                            hadK=(hadK.numpy().astype(np.int8) if hadK is not None else np.zeros((0, 0), np.int8)))
        print("had", n, Kf)


def gen_convert():
    """the reference's own converter scripts run on seeded tiny checkpoints (ap_helpers.convert_input_state_dict:
    32 layers because sqllm_llama_convert_fuse.py:62-69 accepts only "Llama-2-*" directory names).  Stored: for EVERY tensor of the
    input and of the script's output its dtype / shape / sha256 (the input is regenerated from its seed by the tests and checked
    against these digests first), and layers 0 and 31 + the non-layer tensors of the output in full."""
    import json
    import subprocess
    import tempfile
    import torch
    from safetensors.torch import save_file
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ap_helpers import convert_input_state_dict, qtip_convert_input_state_dict, tensor_digest

    def as_np(t):
        return t.view(torch.int16).numpy() if t.dtype == torch.bfloat16 else t.numpy()

    for bitwidth in (2, 3):
        with tempfile.TemporaryDirectory() as tmp:
            d = os.path.join(tmp, "Llama-2-7b")
            os.makedirs(d)
            sd = convert_input_state_dict()
            torch.save(sd, os.path.join(d, "pytorch_model.bin"))
            subprocess.check_call([sys.executable, os.path.join(REF, "inference/sqllm_llama_convert_fuse.py"), "--ckpt_dir", d, "--bitwidth", str(bitwidth)])
            out = torch.load(os.path.join(d, "converted_pytorch_model.bin"), weights_only=True)
        full = {k: as_np(v.contiguous()) for k, v in out.items() if not k.startswith("layers.") or k.startswith("layers.0.") or k.startswith("layers.31.")}
        meta = dict(bitwidth=bitwidth, input={k: tensor_digest(v) for k, v in sd.items()}, output={k: tensor_digest(v) for k, v in out.items()})
        np.savez_compressed(os.path.join(OUT, f"convert_ap_fuse_b{bitwidth}.npz"), meta=json.dumps(meta), **{"full::" + k: v for k, v in full.items()})
        print("convert ap", bitwidth, len(sd), "->", len(out), "tensors")
    with tempfile.TemporaryDirectory() as tmp:
        sd = qtip_convert_input_state_dict()
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(tmp, "model.safetensors"))
        subprocess.check_call([sys.executable, os.path.join(REF, "inference/qtip_convert_no_fuse.py"), "--ckpt_dir", tmp])
        out = torch.load(os.path.join(tmp, "converted_pytorch_model.bin"), weights_only=True)
    meta = dict(input={k: tensor_digest(v) for k, v in sd.items()}, output={k: tensor_digest(v) for k, v in out.items()})
    np.savez_compressed(os.path.join(OUT, "convert_qtip_no_fuse.npz"), meta=json.dumps(meta))
    print("convert qtip", len(sd), "->", len(out), "tensors")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["ap", "qtip", "had", "convert"]
    if "convert" in which:
        gen_convert()
    if "ap" in which:
        gen_ap()
    if "qtip" in which:
        gen_qtip()
    if "had" in which:
        gen_hadamard()
