"""Authoring-container script (like tests/make_golden.py): export the reference's non-power-of-two Hadamard factor
matrices -- literal constant tables, ~95k lines of inference/lib/utils/matmul_had.py:133-95000 (get_had12 .. get_had172;
the public Sloane-library matrices QuIP# / QTIP checkpoints were rotated with) -- into ONE small data file:

    guidedquant_amd/data/hadamard_factors.npz      key "had{K}": uint8, the K x K sign matrix bit-packed row-major
                                                   (np.packbits of entry == +1), 12 orders, ~20 KB

Data, not code: a checkpoint quantized with the reference is only decodable with exactly these matrices (a different
Hadamard matrix of the same order is a different rotation).  `guidedquant_amd.qtip.get_hadK` unpacks them; nothing of
the reference's source text is stored.  Needs /root/reference (not present on the GPU box; the .npz travels)."""
import os
import sys
import types

import numpy as np

REF = "/root/reference/inference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "guidedquant_amd", "data", "hadamard_factors.npz")


def main():
    sys.path.insert(0, REF)
    # matmul_had.py does `from lib import utils` at import time; only its table functions are needed here
    import torch  # noqa: F401
    lib = types.ModuleType("lib")
    lib.utils = types.ModuleType("lib.utils")
    sys.modules.setdefault("lib", lib)
    sys.modules.setdefault("lib.utils", lib.utils)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_matmul_had", os.path.join(REF, "lib", "utils", "matmul_had.py"))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except Exception as e:  # torch.library registrations of the module may fail without the CUDA extension: the tables are defined first
        print("note:", type(e).__name__, e)
    out = {}
    for K in (12, 20, 28, 36, 52, 60, 108, 116, 124, 140, 156, 172):
        h = getattr(mod, f"get_had{K}")().numpy()
        assert h.shape == (K, K) and set(np.unique(h).tolist()) <= {-1.0, 1.0}
        assert np.array_equal(h @ h.T, K * np.eye(K)), K  # it is a Hadamard matrix
        out[f"had{K}"] = np.packbits((h > 0).reshape(-1))
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes", sorted(out))


if __name__ == "__main__":
    main()
