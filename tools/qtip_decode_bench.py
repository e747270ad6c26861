"""QTIP-backend decode throughput on a Llama-2-7b-like model with a power-of-two MLP width (the real 11008 needs the
reference's order-172 Hadamard table, which is not vendored: GQ_HADAMARD_TABLES).  Random-init weights, captured graph."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import model as gm
from guidedquant_amd.generate import load_model, benchmark_decode

inter = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
gm.transformer_configs["llama2-7b-pow2"] = dict(model_name="Llama-2-7b-pow2", block_size=4096, n_layer=layers, n_head=32, n_local_heads=32,
                                                dim=4096, intermediate_size=inter, vocab_size=32000, rope_base=10000)
m = load_model("llama2-7b-pow2", "cuda:0", "qtip", 2, random_init=True)
r = benchmark_decode(m, torch.device("cuda:0"), num_samples=2, max_new_tokens=50)
print(json.dumps({"model": "Llama-2-7b-pow2", "intermediate": inter, "layers": layers, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}}))
