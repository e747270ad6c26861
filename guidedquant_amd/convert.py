"""Checkpoint converters: HF-layout GuidedQuant checkpoints -> the gpt-fast layout the
decode harness loads.  State-dict transformations only (host code).

  * `convert_anyprec_fuse(state_dict, bitwidth, n_layer)`  ==  inference/sqllm_llama_convert_fuse.py:35-118:
    strip "model.", rename (embed_tokens->tok_embeddings, self_attn->attention, o_proj->wo, mlp->feed_forward,
    down_proj->w2, lm_head->output), keep only `lut{bitwidth}` as `lut`, cast bf16 / LUTs to fp16, keep the first
    `bitwidth` bit-planes of every qweight (any-precision prefix property), then fuse q/k/v -> wqkv and gate/up -> w1w3
    by concatenating qweight along dim 1 (rows) and lut along dim 0.
    Generalised: the reference only accepts "Llama-2-*" directory names (:62-69); here the layer count is an argument or
    inferred from the keys, so Llama-3 checkpoints convert too.
  * `convert_qtip_no_fuse(state_dict)`  ==  inference/qtip_convert_no_fuse.py:9-46: key renames only (wq/wk/wv/wo,
    w1/w3/w2), no fusion.
  * CLI: python -m guidedquant_amd.convert --ckpt_dir D --bitwidth 2 [--backend ap|qtip]  (writes
    D/converted_pytorch_model.bin like the reference scripts).
"""
import argparse
import os
import re

import torch

_AP_REPLACEMENTS = {
    'embed_tokens': 'tok_embeddings',
    'self_attn': 'attention',
    'o_proj': 'wo',
    'mlp': 'feed_forward',
    'down_proj': 'w2',
    'lm_head': 'output',
    'lookup_table': 'lut',
}

_QTIP_REPLACEMENTS = {
    'embed_tokens': 'tok_embeddings',
    'self_attn': 'attention',
    'q_proj': 'wq',
    'k_proj': 'wk',
    'v_proj': 'wv',
    'o_proj': 'wo',
    'mlp': 'feed_forward',
    'gate_proj': 'w1',
    'up_proj': 'w3',
    'down_proj': 'w2',
    'lm_head': 'output',
}


def _rename(key, table):
    new_key = key.replace('model.', '')
    for old, new in table.items():
        new_key = new_key.replace(old, new)
    return new_key


def infer_n_layer(state_dict):
    idx = [int(m.group(1)) for k in state_dict for m in [re.search(r'layers\.(\d+)\.', k)] if m]
    if not idx:
        raise ValueError("no 'layers.N.' keys in the checkpoint")
    return max(idx) + 1


def convert_anyprec_fuse(state_dict, bitwidth, n_layer=None):
    new_dict = {}
    for key, value in state_dict.items():
        new_key = _rename(key, _AP_REPLACEMENTS)
        if 'lut' in new_key:
            if f'lut{bitwidth}' in new_key:
                new_key = re.sub(r'(?<=lut)[2-8]', '', new_key)
            else:
                continue
        new_dict[new_key] = value
    for key in list(new_dict.keys()):
        v = new_dict[key]
        if v.dtype == torch.bfloat16:
            v = v.half()
        if key.endswith('.lut'):
            v = v.half()
        if 'qweight' in key:
            v = v.contiguous()[:bitwidth, :, :].contiguous()
        new_dict[key] = v
    if n_layer is None:
        n_layer = infer_n_layer(new_dict)
    for i in range(n_layer):
        a, f = f'layers.{i}.attention.', f'layers.{i}.feed_forward.'
        for suffix, dim in (('qweight', 1), ('lut', 0)):
            new_dict[a + 'wqkv.' + suffix] = torch.cat(
                (new_dict.pop(a + 'q_proj.' + suffix), new_dict.pop(a + 'k_proj.' + suffix), new_dict.pop(a + 'v_proj.' + suffix)),
                dim=dim).contiguous()
            new_dict[f + 'w1w3.' + suffix] = torch.cat(
                (new_dict.pop(f + 'gate_proj.' + suffix), new_dict.pop(f + 'up_proj.' + suffix)), dim=dim).contiguous()
    return new_dict


def convert_qtip_no_fuse(state_dict):
    return {_rename(k, _QTIP_REPLACEMENTS): v for k, v in state_dict.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ckpt_dir', type=str, required=True)
    ap.add_argument('--bitwidth', type=int, default=2)
    ap.add_argument('--backend', choices=['ap', 'qtip'], default='ap')
    args = ap.parse_args()
    if args.backend == 'ap':
        ckpt = torch.load(os.path.join(args.ckpt_dir, "pytorch_model.bin"), weights_only=True)
        out = convert_anyprec_fuse(ckpt, args.bitwidth)
    else:
        from safetensors.torch import load_file
        out = convert_qtip_no_fuse(load_file(os.path.join(args.ckpt_dir, "model.safetensors")))
    torch.save(out, os.path.join(args.ckpt_dir, "converted_pytorch_model.bin"))


if __name__ == "__main__":
    main()
