"""Loader for HF-layout GuidedQuant / Any-Precision checkpoints: a directory with
`config.json` (a Llama config plus the `anyprec` section, any_precision/modules/AnyPrecisionForCausalLM.py:43-47) and the
weights as `pytorch_model.bin` or (sharded) safetensors with the HF keys `model.layers.{i}.self_attn.q_proj.{qweight,lut{b}}`
(any_precision/quantization/pack.py:112-123) -> the fused gpt-fast `Transformer` of this package, ready for the HIP
decode path.  Equivalent to running inference/sqllm_llama_convert_fuse.py and then inference/generate.py::load_model,
without the intermediate file and without the "Llama-2-*" directory-name restriction (any layer count / GQA geometry).
Host code only; plain torch.load / safetensors and manual device placement (no accelerate dispatch)."""
import glob
import json
import os

import torch

from .APLinear import APLinear
from .convert import convert_anyprec_fuse
from .model import ModelArgs, Transformer


def model_args_from_hf_config(cfg: dict) -> ModelArgs:
    """Llama `config.json` -> ModelArgs (inference/model.py:27-51 field meanings)."""
    name = os.path.basename(str(cfg.get("_name_or_path") or cfg.get("model_type", "llama")).rstrip("/")) or "llama"
    if "llama" not in name.lower():  # (the block layout is Llama's whatever the checkpoint directory is called)
        name = "llama-" + name
    return ModelArgs(block_size=int(cfg.get("max_position_embeddings", 8192)), vocab_size=int(cfg["vocab_size"]),
                     n_layer=int(cfg["num_hidden_layers"]), n_head=int(cfg["num_attention_heads"]), dim=int(cfg["hidden_size"]),
                     intermediate_size=int(cfg["intermediate_size"]),
                     n_local_heads=int(cfg.get("num_key_value_heads", cfg["num_attention_heads"])),
                     rope_base=float(cfg.get("rope_theta", 10000.0)), norm_eps=float(cfg.get("rms_norm_eps", 1e-5)),
                     rope_scaling=cfg.get("rope_scaling"), model_name=os.path.basename(str(name)))


def read_hf_state_dict(path: str) -> dict:
    """pytorch_model.bin, model.safetensors or sharded model-*-of-*.safetensors under `path`."""
    p = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(p):
        return torch.load(p, map_location="cpu", weights_only=True)
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no pytorch_model.bin / *.safetensors under {path}")
    from safetensors.torch import load_file
    sd = {}
    for f in files:
        sd.update(load_file(f))
    return sd


def supported_precisions(cfg: dict):
    ap = cfg.get("anyprec")
    if not ap:
        raise ValueError("config.json has no 'anyprec' section (not an Any-Precision checkpoint)")
    return list(range(int(ap["seed_precision"]), int(ap["parent_precision"]) + 1))


def load_anyprec_hf(path: str, bitwidth: int = None, device="cuda", dtype=torch.float16) -> Transformer:
    """Build the fused decode model from an HF-layout Any-Precision checkpoint directory at `bitwidth`
    (default: the seed precision, the only one GuidedQuant checkpoints carry, scripts/run_lnq.sh)."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    bits = supported_precisions(cfg)
    if bitwidth is None:
        bitwidth = bits[0]
    if bitwidth not in bits:
        raise ValueError(f"bitwidth {bitwidth} not in the checkpoint's precisions {bits}")
    return anyprec_state_dict_to_transformer(read_hf_state_dict(path), cfg, bitwidth, device, dtype)


def anyprec_state_dict_to_transformer(sd: dict, cfg: dict, bitwidth: int, device="cuda", dtype=torch.float16) -> Transformer:
    """HF-keyed Any-Precision tensors (host or device) + the config dict -> the fused decode model at `bitwidth`"""
    args = model_args_from_hf_config(cfg)
    sd = {k: v for k, v in sd.items() if "rotary_emb" not in k}
    if "lm_head.weight" not in sd and cfg.get("tie_word_embeddings", False):
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]  # tied embeddings (Llama-3.2-1B)
    fused = convert_anyprec_fuse(sd, bitwidth, n_layer=args.n_layer)
    on_dev = all(v.device.type != "cpu" for v in fused.values())
    model = Transformer(dtype, args, linear_class=APLinear, linear_kwargs=dict(bitwidth=bitwidth, device=device if on_dev else "cpu"),
                        fuse_linears=True)
    if on_dev:
        model = model.to(device=device, dtype=dtype)
    model.load_state_dict(fused, strict=True)
    return model.to(device=device, dtype=dtype).eval()
