"""AnyPrecisionForCausalLM -- the HF-path entry of the reference (any_precision/modules/AnyPrecisionForCausalLM.py:28-210) on
MI355X: an HF causal LM whose decoder linears are `AnyPrecisionLinear` modules (multi-precision parent tensor + one LUT per
precision), with `from_quantized(path, precisions=...)`, `forward(..., precision=b)`, `generate(..., precision=b)`,
`set_precision`, `prune_precisions` -- the harness the reference's only published number (130 tok/s, README.md:95-97,
inference_example.py:34-77) runs on.

Same public surface; the construction is this package's own:
  * the skeleton comes from `AutoModelForCausalLM.from_config` on the meta device; which modules to swap is read from the
    checkpoint's `config.anyprec["arch_config"]` (`model_name`, `layers_name`, `module_names`, pack.py:190-195) -- or, when a
    checkpoint lacks it, every nn.Linear inside the decoder layers -- instead of the reference's analyzer package;
  * weights are read with plain torch / safetensors (hf_loader.read_hf_state_dict) and placed on ONE device: one process
    per GPU is this framework's model (a 70B 2-bit model is 19 GB of a 288 GB HBM), there is no accelerate `device_map`;
  * a rows == 1 call runs the HIP LUT-GEMV (plugin::anyprec_gemv), more rows the dequant + matmul branch, host tensors the
    CPU twins (AnyPrecisionLinear.py).
`native_decoder(bitwidth)` additionally returns the fused gpt-fast `Transformer` of the same checkpoint (hf_loader), whose
decode step is the 5-launches-per-layer hipGraph path bench.py measures.
"""
import gc
import json
import os
from typing import List, Optional

import torch
import torch.nn as nn

from .AnyPrecisionLinear import AnyPrecisionLinear
from .hf_loader import read_hf_state_dict


def replace_module_by_name(layer, module_name, new_module):
    levels = module_name.split('.')
    module = layer
    for level in levels[:-1]:
        module = getattr(module, level) if not level.isdigit() else module[int(level)]
    setattr(module, levels[-1], new_module)


def _get_by_path(root, dotted):
    for name in [p for p in dotted.split('.') if p]:
        root = getattr(root, name) if not name.isdigit() else root[int(name)]
    return root


class AnyPrecisionForCausalLM(nn.Module):

    def __init__(self, model_path, config, precisions: Optional[List[int]] = None, torch_dtype=torch.float16, fuse_layers=False,
                 trust_remote_code=True, local_dir=None, device=None, random_init_seed: Optional[int] = None):
        super().__init__()
        from transformers import AutoModelForCausalLM
        if torch_dtype != torch.float16:
            raise RuntimeError('Only float16 is supported for now.')
        self.config = config
        self.model_path = model_path
        ap = config.anyprec if hasattr(config, "anyprec") else None
        if not ap:
            raise ValueError("config has no 'anyprec' section (not an Any-Precision checkpoint)")
        self.supported_bits = list(range(ap['seed_precision'], ap['parent_precision'] + 1))
        if precisions is None:
            self.precisions = self.supported_bits
        else:
            assert len(precisions) == len(set(precisions)), "Precisions must be unique"
            assert all(bit in self.supported_bits for bit in precisions), \
                f"Supported bits {precisions} must be a subset of model supported bits {self.supported_bits}"
            self.precisions = precisions
        self.precision = max(self.precisions)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)

        with torch.device("meta"):
            self.model = AutoModelForCausalLM.from_config(config=config, torch_dtype=torch_dtype, trust_remote_code=trust_remote_code)
        self.ap_linears = []
        self._load_quantized_modules()
        self._native_cache = {}
        if random_init_seed is not None:
            self._random_weights(random_init_seed)
        else:
            if not os.path.exists(model_path):
                raise FileNotFoundError(f"{model_path}: not a local checkpoint directory (there is no hub download here; "
                                        "the reference calls snapshot_download)")
            self._load_weights(model_path)
        self.tie_weights()
        if fuse_layers:
            self.fuse_layers()
        self.prune_precisions()

    # -- construction ------------------------------------------------------------------------------------------------
    def get_model_layers(self):
        arch = (self.config.anyprec.get('arch_config') or {})
        module = _get_by_path(self.model, arch.get('model_name', 'model'))
        return getattr(module, arch.get('layers_name', 'layers'))

    @property
    def layer_type(self):
        for layer in self.get_model_layers():
            return layer.__class__.__name__
        return None

    def _load_quantized_modules(self):
        arch = (self.config.anyprec.get('arch_config') or {})
        names = arch.get('module_names')
        for layer in self.get_model_layers():
            targets = {}
            if names:
                for n in names:
                    targets[n] = _get_by_path(layer, n)
            else:
                targets = {n: m for n, m in layer.named_modules() if isinstance(m, nn.Linear)}
            for name, module in targets.items():
                lin = AnyPrecisionLinear(module.in_features, module.out_features, self.supported_bits, bias=module.bias is not None,
                                         precisions=self.precisions, device="meta", dtype=torch.float16)
                lin.output = torch.zeros((1, 1, module.out_features), dtype=torch.float16, device=self.device)
                self.ap_linears.append(lin)
                replace_module_by_name(layer, name, lin)

    def _materialise_meta_buffers(self):
        # non-persistent buffers (the rotary embedding's inv_freq) have no data in a checkpoint and are still on the meta
        # device: a fresh instance of the owning module, built from its config on the CPU, supplies them
        for mod in list(self.model.modules()):
            meta = [b for b, buf in mod._buffers.items() if buf is not None and buf.is_meta]
            if not meta:
                continue
            if not hasattr(mod, "config"):
                raise RuntimeError(f"buffers {meta} of {type(mod).__name__} have no data in the checkpoint")
            with torch.device("cpu"):
                fresh = type(mod)(mod.config)
            for b in meta:
                mod._buffers[b] = fresh._buffers[b]

    def _random_weights(self, seed, lut_std=0.02):
        """synthetic weights of the checkpoint format, generated on the device (benchmarks / tests without a checkpoint on disk: the
        `--random_init` of inference/generate.py:222-245 for the HF-path harness)"""
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        sd = {}
        for name, t in list(self.model.state_dict().items()):
            shape = tuple(t.shape)
            if name.endswith("qweight"):
                v = torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=self.device, generator=g)
            elif ".lut" in name:
                v = (torch.randn(shape, device=self.device, generator=g) * lut_std).sort(dim=1).values.half().contiguous()
            elif name.endswith("norm.weight") or "layernorm" in name:
                v = torch.ones(shape, dtype=torch.float16, device=self.device)
            else:
                v = (torch.randn(shape, device=self.device, generator=g) * 0.02).half()
            sd[name] = v
        self.model.load_state_dict(sd, strict=False, assign=True)
        self._materialise_meta_buffers()
        self.model.to(self.device)
        for lin in self.ap_linears:
            lin.output = lin.output.to(self.device)

    def _load_weights(self, path):
        sd = read_hf_state_dict(path)
        if "lm_head.weight" not in sd and getattr(self.config, "tie_word_embeddings", False):
            emb = [k for k in sd if k.endswith("embed_tokens.weight")]
            if emb:
                sd["lm_head.weight"] = sd[emb[0]]
        cast = {k: (v.to(torch.float16) if v.is_floating_point() else v) for k, v in sd.items()}
        missing, unexpected = self.model.load_state_dict(cast, strict=False, assign=True)
        missing = [k for k in missing if "rotary_emb.inv_freq" not in k]
        if missing:
            raise RuntimeError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:4]}")
        self._materialise_meta_buffers()
        self.model.to(self.device)
        for lin in self.ap_linears:
            lin.output = lin.output.to(self.device)
        del sd, cast
        gc.collect()

    # -- the reference's surface ---------------------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        prev_precision = self.precision
        if 'precision' in kwargs:
            self.set_precision(kwargs.pop('precision'))
        results = self.model.forward(*args, **kwargs)
        self.set_precision(prev_precision)
        return results

    def generate(self, *args, **kwargs):
        """HF `generate` on the module tree (inference_example.py:34-77) -- or, with native=True and a plain bs=1 request
        (input_ids, max_new_tokens, do_sample / temperature / top_k), the same checkpoint's fused decode model behind the same
        call: prompt through the HIP prompt pass, every new token one replay of the captured 5-launches-per-layer graph.  Returns
        the [1, prompt + new] token tensor like HF does."""
        prev_precision = self.precision
        if 'precision' in kwargs:
            self.set_precision(kwargs.pop('precision'))
        try:
            if kwargs.pop('native', False):
                return self._generate_native(*args, **kwargs)
            with torch.inference_mode():
                return self.model.generate(*args, **kwargs)
        finally:
            self.set_precision(prev_precision)

    def _generate_native(self, input_ids=None, max_new_tokens=100, do_sample=False, temperature=1.0, top_k=32, **unused):
        from . import generate as gen
        ids = (input_ids if input_ids is not None else unused.pop("inputs")).to(self.device)
        if ids.dim() != 2 or ids.shape[0] != 1:
            raise ValueError("native=True serves one sequence (bs = 1)")
        unsupported = [k for k in unused if k not in ("cache_implementation", "pad_token_id", "eos_token_id", "attention_mask", "use_cache")]
        if unsupported:
            raise ValueError(f"native=True does not take {unsupported}")
        dec = self.native_decoder(self.precision)
        T = ids.shape[1]
        dec.setup_caches(1, T + max_new_tokens)
        temp = float(temperature) if do_sample else 0.0
        key = (self.precision, dec.max_seq_length, temp, int(top_k))
        graph = self._native_cache.get(("graph",) + key)
        if graph is None:
            graph = gen.DecodeGraph(dec, self.device, native_sampling=True, temperature=temp, top_k=int(top_k))
            self._native_cache = {k: v for k, v in self._native_cache.items() if k[0] != "graph"}
            self._native_cache[("graph",) + key] = graph
        with torch.inference_mode():
            seq = gen.generate(dec, ids.view(-1).to(torch.int32), max_new_tokens, use_graph=True, graph=graph, temperature=temp, top_k=int(top_k))
        return seq.to(ids.dtype)

    @staticmethod
    def _load_config(model_path, trust_remote_code=True):
        from transformers import AutoConfig
        return AutoConfig.from_pretrained(model_path, trust_remote_code=trust_remote_code)

    @classmethod
    def from_config_random(cls, config, precisions=None, device=None, seed=0, torch_dtype=torch.float16):
        """the module tree of `config` (a transformers config with the `anyprec` section) with synthetic weights, no checkpoint"""
        return cls(model_path=None, config=config, precisions=precisions, torch_dtype=torch_dtype, device=device, random_init_seed=seed)

    @classmethod
    def from_quantized(cls, quant_model_path, trust_remote_code=True, fuse_layers=False, precisions=None, local_dir=None,
                       torch_dtype=torch.float16, device=None):
        config = cls._load_config(quant_model_path, trust_remote_code)
        return cls(model_path=quant_model_path, precisions=precisions, config=config, fuse_layers=fuse_layers,
                   trust_remote_code=trust_remote_code, local_dir=local_dir, torch_dtype=torch_dtype, device=device)

    def prune_precisions(self):
        for ap_linear in self.ap_linears:
            ap_linear.prune_precisions()
        gc.collect()

    def set_precision(self, precision):
        for ap_linear in self.ap_linears:
            ap_linear.set_precision(precision)
        self.precision = precision

    def tie_weights(self):
        if hasattr(self.model, "tie_weights"):
            self.model.tie_weights()

    def fuse_layers(self):
        # the reference leaves this a TODO (AnyPrecisionForCausalLM.py:192-196); the fused decode model is `native_decoder`
        raise NotImplementedError("layer fusion inside the HF module tree is not implemented (as in the reference); use "
                                  "native_decoder(bitwidth) for the fused QKV / Up-Gate decode path")

    def native_decoder(self, bitwidth: Optional[int] = None):
        """the same checkpoint as the fused gpt-fast `Transformer` (fused QKV / Up-Gate Any-Precision linears at one precision)
        whose bs=1 decode step runs as the captured 5-launches-per-layer HIP graph"""
        from .hf_loader import anyprec_state_dict_to_transformer, load_anyprec_hf
        bitwidth = bitwidth or min(self.precisions)
        dec = self._native_cache.get(("decoder", bitwidth))
        if dec is None:
            if self.model_path is not None and os.path.exists(str(self.model_path)):
                dec = load_anyprec_hf(self.model_path, bitwidth=bitwidth, device=self.device)
            else:  # no checkpoint on disk (from_config_random): from the module tree's own tensors
                cfg = self.config.to_dict() if hasattr(self.config, "to_dict") else dict(self.config)
                dec = anyprec_state_dict_to_transformer(dict(self.model.state_dict()), cfg, bitwidth, self.device)
            self._native_cache[("decoder", bitwidth)] = dec
        return dec
