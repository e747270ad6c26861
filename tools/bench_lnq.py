"""LNQ assignment update (update_P) on the GPU: the fused block kernel path of guidedquant_amd.lnq against a per-column torch loop
(the structure of the reference's update_P: ~6 small launches per input column), one CD cycle on a 4096 x 4096 2-bit layer."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd import lnq  # noqa: E402


def per_column_loop(W, Hn, What, C, assign):
    """plain torch statement of one cycle, column by column (what the fused kernel replaces)"""
    N, d = W.shape
    B = (What - W) @ torch.tril(Hn[0], diagonal=-1)
    for st in range(0, d, 128):
        end = min(st + 128, d)
        for j in range(st, end):
            sol = W[:, j] - B[:, j]
            arg = (sol[:, None] - C).abs().argmin(dim=1)
            val = torch.gather(C, 1, arg[:, None])[:, 0]
            assign[:, j] = arg
            What[:, j] = val
            if j + 1 < end:
                B[:, j + 1:end] += (val - W[:, j])[:, None] * Hn[0, j, j + 1:end][None, :]
        if end < d:
            B[:, end:] += (What[:, st:end] - W[:, st:end]) @ Hn[0, st:end, end:]
    return assign


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    N = d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ncl = 4
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    W = torch.randn(N, d, device=dev, generator=g) * 0.02
    X = torch.randn(2 * d, d, device=dev, generator=g)
    H = (X.T @ X / (2 * d) + 1e-2 * torch.eye(d, device=dev))[None].contiguous()
    qs = torch.tensor([(i + 0.5) / ncl for i in range(ncl)], device=dev)
    C = torch.quantile(W, qs, dim=1).T.contiguous()
    labels = (W[:, :, None] - C[:, None, :]).abs().argmin(-1)
    o0 = float(lnq.objective_function(W, H, labels, C))
    lnq.update_P(W, H, labels, C, 1, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    new = lnq.update_P(W, H, labels, C, 1, verbose=False)
    torch.cuda.synchronize()
    t_fused = time.perf_counter() - t0
    Hn = (H / torch.diagonal(H, dim1=1, dim2=2)[:, None, :]).contiguous()
    What = torch.gather(C, 1, labels)
    t0 = time.perf_counter()
    ref = per_column_loop(W, Hn, What.clone(), C, labels.clone())
    torch.cuda.synchronize()
    t_loop = time.perf_counter() - t0
    print(json.dumps({"N": N, "d": d, "n_cluster": ncl, "cycles": 1, "fused_update_P_s": round(t_fused, 4), "per_column_torch_loop_s": round(t_loop, 3),
                      "speedup": round(t_loop / t_fused, 1), "objective_before": o0, "objective_fused": float(lnq.objective_function(W, H, new, C)),
                      "objective_loop": float(lnq.objective_function(W, H, ref, C)), "assignments_equal_frac": float((new == ref).float().mean())}))
