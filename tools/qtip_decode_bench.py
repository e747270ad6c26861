"""QTIP-backend decode throughput on a Llama-2-7b-shaped model: `qtip_decode_bench.py 11008` is the real MLP width (needs the
reference's order-172 Hadamard table: GQ_HADAMARD_TABLES, or the copy in the parity fixtures), `8192` a power-of-two
stand-in that needs no table.  Random-init weights, captured graph."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import model as gm
from guidedquant_amd.generate import load_model, benchmark_decode

inter = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if inter & (inter - 1) and not os.environ.get("GQ_HADAMARD_TABLES"):
    # measurement convenience only: the order-172 / order-28 tables are the caller's data (GQ_HADAMARD_TABLES); the parity
    # fixtures under tests/golden carry a copy (generated from the reference by tests/make_golden.py)
    import numpy as np, tempfile
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    tabs = {}
    for f in ("had_n11008.npz", "had_n14336.npz"):
        g = np.load(os.path.join(gold, f))
        tabs["had%d" % int(g["K"])] = g["hadK"]
    path = os.path.join(tempfile.mkdtemp(), "tables.npz")
    np.savez(path, **tabs)
    os.environ["GQ_HADAMARD_TABLES"] = path
gm.transformer_configs["llama2-7b-pow2"] = dict(model_name="Llama-2-7b" if inter == 11008 else "Llama-2-7b-pow2", block_size=4096, n_layer=layers, n_head=32, n_local_heads=32,
                                                dim=4096, intermediate_size=inter, vocab_size=32000, rope_base=10000)
m = load_model("llama2-7b-pow2", "cuda:0", "qtip", 2, random_init=True)
r = benchmark_decode(m, torch.device("cuda:0"), num_samples=2, max_new_tokens=50,
                     native_sampling=os.environ.get("GQ_TORCH_SAMPLING", "0") == "0")  # fused HIP sampler, as bench.py
print(json.dumps({"model": "Llama-2-7b" if inter == 11008 else "Llama-2-7b-pow2", "native": m._native_kind(), "intermediate": inter, "layers": layers, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}}))
