"""Authoring-container script: golden vectors of the LNQ inner loops from the reference's own
functions -- `objective_function`, `update_P` (the coordinate-descent assignment update) and `update_C` (the per-row least
squares centroid update) of any_precision/quantization/layerwise_quantize.py:15-204 -- executed on the CPU.

The module is loaded from its file under /root/reference with (a) its two imports that are irrelevant to these functions
replaced by empty stand-ins (`any_precision.analyzer.analyzer.ModelAnalyzer`, whose package does not import under the installed
transformers, and the tqdm progress-bar helper) and (b) `torch.device("cuda")` answered with the CPU device: the reference
hard-codes the device string and this container has no GPU.  The arithmetic that runs is the reference's.  Only inputs and
outputs are stored (tests/golden/lnq_*.npz)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/any_precision/quantization/layerwise_quantize.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _TorchProxy:
    """forwards to torch; torch.device("cuda") -> the CPU device"""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def device(*a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            return torch.device("cpu")
        return torch.device(*a, **k)


def load_reference():
    class _Bar:
        def update(self, n):
            pass

        def close(self):
            pass

    pkg = types.ModuleType("any_precision")
    pkg.__path__ = []
    q = types.ModuleType("any_precision.quantization")
    q.__path__ = []
    an = types.ModuleType("any_precision.analyzer")
    an.__path__ = []
    ana = types.ModuleType("any_precision.analyzer.analyzer")
    ana.ModelAnalyzer = object
    utils = types.ModuleType("any_precision.quantization.utils")
    utils.get_progress_bar = lambda total, desc: _Bar()
    for name, m in (("any_precision", pkg), ("any_precision.quantization", q), ("any_precision.analyzer", an),
                    ("any_precision.analyzer.analyzer", ana), ("any_precision.quantization.utils", utils)):
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("any_precision.quantization.layerwise_quantize", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.torch = _TorchProxy()
    return mod


def make_case(rng, N, d, n_cluster, num_groups):
    W = rng.normal(0, 0.02, (N, d)).astype(np.float32)
    X = rng.normal(0, 1, (num_groups, 4 * d, d)).astype(np.float32) * (1 + 3 * (rng.random((1, 1, d)) < 0.03))
    H = np.einsum("gsi,gsj->gij", X, X).astype(np.float32) / (4 * d)
    H += 1e-3 * np.eye(d, dtype=np.float32)[None] * np.trace(H, axis1=1, axis2=2)[:, None, None] / d
    # centroids: per-row quantiles of the weights; labels: nearest centroid (a SqueezeLLM-like seed)
    qs = (np.arange(n_cluster) + 0.5) / n_cluster
    C = np.quantile(W, qs, axis=1).T.astype(np.float32)
    labels = np.abs(W[:, :, None] - C[:, None, :]).argmin(-1).astype(np.int8)
    return W, H, labels, C


def main():
    mod = load_reference()
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)
    for tag, (N, d, n_cluster, num_groups, cycles) in {"b2_n128_d256": (128, 256, 4, 1, 2), "b3_n128_d384_g2": (128, 384, 8, 2, 1),
                                                         "b4_n64_d128": (64, 128, 16, 1, 3)}.items():
        W, H, labels, C = make_case(rng, N, d, n_cluster, num_groups)
        Wt, Ht = torch.tensor(W), torch.tensor(H)
        lt, Ct = torch.tensor(labels), torch.tensor(C)
        obj0 = float(mod.objective_function(Wt, Ht, lt, Ct))
        newl = mod.update_P(Wt, Ht, lt, Ct, cd_cycles=cycles, verbose=False)
        obj1 = float(mod.objective_function(Wt, Ht, newl, Ct))
        newC = mod.update_C(Wt, Ht, newl, Ct, 0)
        obj2 = float(mod.objective_function(Wt, Ht, newl, newC))
        np.savez_compressed(os.path.join(OUT, f"lnq_{tag}.npz"), W=W, H=H, labels=labels, C=C, cd_cycles=cycles, obj0=obj0,
                            labels_P=newl.cpu().numpy().astype(np.int8), obj1=obj1, C_new=newC.cpu().numpy().astype(np.float32), obj2=obj2)
        print(tag, "objective", obj0, "->", obj1, "->", obj2, "changed %.2f%%" % (100 * float((newl.cpu().numpy() != labels).mean())))


if __name__ == "__main__":
    main()
