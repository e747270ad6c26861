"""LNQ inner loops on MI355X -- the two alternating updates of the layer-wise non-uniform
quantization objective  sum_i (w_hat_i - w_i) H (w_hat_i - w_i)^T  of the reference's calibration
(any_precision/quantization/layerwise_quantize.py): same functions, argument order and results; the formulations are this
package's own.

  objective_function(W, H, labels, C)   :15-49   mean over rows of the quadratic form
  update_P(W, H, labels, C, cd_cycles)  :51-127  coordinate descent on the assignments, column by column (Gauss-Seidel)
  update_C(W, H, labels, C, iteration)  :129-204 per-row least squares for the centroids
  train_least_squares(...)              :206-288 the alternating loop with early stopping

update_P: the sequential 128-column inner loop (:93-118, ~6 torch launches per column in the reference) is ONE HIP launch per
block (`gq_lnq_cd_block`, csrc/lnq.hip); the two GEMM-shaped updates around it are library GEMMs.  update_C: the reference forms
A_i = L^T P_i ([d x n_cluster] per row, L = chol(H)) and calls lstsq on the damped system; here the same damped normal equations
(P_i^T H P_i + lambda I) c_i = P_i^T H w_i are assembled from n_cluster masked GEMMs against H (no Cholesky, no [N, d, n_cluster]
intermediate) and solved in float64 -- equal up to the conditioning of lstsq vs normal equations (tests: 1e-3 of the centroid scale,
1e-4 of the objective).  Everything runs on the device of W (CPU tensors take a torch restatement of the inner loop, used by the
CPU test-suite against the reference-generated goldens).  The calibration pipeline around these loops (Hessian capture, seeding,
layer streaming) stays with the reference, as the north star prescribes.
"""
import logging
from typing import Tuple

import numpy as np
import torch

from . import _lib

CD_BLOCK = 128  # layerwise_quantize.py:91


def _w_hat(C: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    return torch.gather(C, 1, labels.long())


@torch.no_grad()
def objective_function(W: torch.Tensor, H: torch.Tensor, labels: torch.Tensor, C: torch.Tensor) -> torch.Tensor:
    dev = W.device
    delta = _w_hat(C.to(dev), labels.to(dev)) - W
    G = H.shape[0]
    dg = delta.reshape(G, -1, delta.shape[-1])
    # einsum('nij,njk,nik->i') of the reference: summed over the groups n, then the mean over the rows i of a group
    return (torch.bmm(dg, H.to(dev)) * dg).sum(-1).sum(0).mean()


def _cd_block(W, B, Hn, C, assign, What, group_rows, st, end):
    """the sequential inner loop over the columns [st, end) -- one HIP launch on the GPU, a torch restatement on the CPU"""
    N, d = W.shape
    if W.is_cuda:
        with torch.cuda.device(W.device):
            rc = _lib.lib().gq_lnq_cd_block(W.data_ptr(), B.data_ptr(), Hn.data_ptr(), C.data_ptr(), assign.data_ptr(), What.data_ptr(), N, d,
                                            C.shape[1], group_rows, st, end, _lib.current_stream_ptr())
        if rc != _lib.GQ_ENOTSUP:
            _lib.check(rc, "gq_lnq_cd_block")
            return
        # shapes the kernel does not serve (rows per Hessian group not a multiple of 32, more than 16 centroids): the same loop
        # column by column with torch ops ON THE SAME DEVICE (the reference's own formulation, layerwise_quantize.py:93-118)
    G = Hn.shape[0]
    Bb = B[:, st:end].clone()
    Hrow = Hn[:, st:end, st:end].repeat_interleave(group_rows, dim=0) if G > 1 else None
    for j in range(end - st):
        w = W[:, st + j]
        sol = w - Bb[:, j]
        arg = (sol[:, None] - C).abs().argmin(dim=1)
        val = torch.gather(C, 1, arg[:, None])[:, 0]
        assign[:, st + j] = arg.to(assign.dtype)
        What[:, st + j] = val
        if j + 1 < end - st:
            h = Hn[0, st + j, st + j + 1:end][None, :] if G == 1 else Hrow[:, j, j + 1:]
            Bb[:, j + 1:] = Bb[:, j + 1:] + (val - w)[:, None] * h


@torch.no_grad()
def update_P(W: torch.Tensor, H: torch.Tensor, labels: torch.Tensor, C: torch.Tensor, cd_cycles: int, verbose: bool = True) -> torch.Tensor:
    dev = W.device
    W = W.float().contiguous()
    C = C.to(dev).float().contiguous()
    N, d = W.shape
    G = H.shape[0]
    assert N % G == 0
    group_rows = N // G
    prev = labels.to(dev).long()
    assign = prev.to(torch.uint8).contiguous()
    What = _w_hat(C, prev).contiguous()
    # column k of H divided by H[k][k]: Hn[a][k] = H[a][k] / H[k][k]
    Hn = (H.to(dev).float() / torch.diagonal(H.to(dev).float(), dim1=1, dim2=2)[:, None, :]).contiguous()
    lower = torch.tril(Hn, diagonal=-1)
    Wg, Whg = W.view(G, group_rows, d), What.view(G, group_rows, d)
    for _ in range(cd_cycles):
        B = torch.bmm(Whg - Wg, lower).reshape(N, d).contiguous()  # terms of the columns behind each column (old assignments)
        for st in range(0, d, CD_BLOCK):
            end = min(st + CD_BLOCK, d)
            _cd_block(W, B, Hn, C, assign, What, group_rows, st, end)
            if end < d:  # the block's new differences reach the columns ahead
                Bg = B.view(G, group_rows, d)
                Bg[:, :, end:] += torch.bmm(Whg[:, :, st:end] - Wg[:, :, st:end], Hn[:, st:end, end:])
    out = assign.long()
    if verbose:
        logging.info(f"Percentage of assignments changed: {100.0 * float((out != prev).float().mean()):.2f}%")
    return out


@torch.no_grad()
def update_C(W: torch.Tensor, H: torch.Tensor, labels: torch.Tensor, C: torch.Tensor, iteration: int = 0, lambda_reg: float = 1e-7) -> torch.Tensor:
    dev = W.device
    W = W.float()
    N, d = W.shape
    ncl = C.shape[-1]
    G = H.shape[0]
    group_rows = N // G
    lab = labels.to(dev).long()
    Hd = H.to(dev).float()
    gram = torch.empty(N, ncl, ncl, dtype=torch.float64, device=dev)
    rhs = torch.empty(N, ncl, dtype=torch.float64, device=dev)
    HW = torch.bmm(W.view(G, group_rows, d), Hd).reshape(N, d)  # rows w_i H (H symmetric)
    masks = [(lab == a).float() for a in range(ncl)]
    for a in range(ncl):
        Ma = torch.bmm(masks[a].view(G, group_rows, d), Hd).reshape(N, d)  # row i: sum of the rows of H with label a
        for b in range(ncl):
            gram[:, a, b] = (Ma * masks[b]).sum(-1, dtype=torch.float64)  # d-long reductions accumulate in fp64
        rhs[:, a] = (HW * masks[a]).sum(-1, dtype=torch.float64)
    gram += lambda_reg * torch.eye(ncl, dtype=torch.float64, device=dev)
    sol = torch.linalg.solve(gram, rhs.unsqueeze(-1)).squeeze(-1)
    if torch.isnan(sol).any():
        raise RuntimeError("NaN in the centroid update")
    return sol.float().cpu()


def train_least_squares(W: np.ndarray, init_labels: np.ndarray, init_centroids: np.ndarray, H: np.ndarray, num_iterations: int = 3,
                        cd_cycles: int = 4, device=None) -> Tuple[np.ndarray, np.ndarray, dict]:
    dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    labels = torch.tensor(init_labels, dtype=torch.int8)
    C = torch.tensor(init_centroids, dtype=torch.float32)
    Wt = torch.tensor(W, dtype=torch.float32, device=dev)
    Ht = torch.tensor(H, dtype=torch.float32, device=dev)
    for i in range(Ht.shape[0]):  # damp H until it is positive definite (:222-239)
        avg = torch.diagonal(Ht[i]).mean()
        damp, prev = 1e-5, 0.0
        while True:
            if torch.linalg.cholesky_ex(Ht[i]).info.item() == 0:
                break
            Ht[i].diagonal().add_((damp - prev) * avg)
            prev, damp = damp, damp * 10
            if damp > 1.0:
                raise RuntimeError("Hessian is not positive definite even with dampening 1e0")
    best = float(objective_function(Wt, Ht, labels, C))
    best_labels, best_C = labels.clone(), C.clone()
    log = {"objective": [best], "iteration": [0]}
    for it in range(num_iterations):
        if it > 0:
            labels = update_P(Wt, Ht, labels, C, cd_cycles=cd_cycles, verbose=False).to(torch.int8).cpu()
        log["objective"].append(float(objective_function(Wt, Ht, labels, C)))
        log["iteration"].append(it + 1)
        C = update_C(Wt, Ht, labels, C, it)
        cur = float(objective_function(Wt, Ht, labels, C))
        log["objective"].append(cur)
        log["iteration"].append(it + 1)
        if cur < best:
            best, best_labels, best_C = cur, labels.clone(), C.clone()
        else:
            labels, C = best_labels, best_C
            break
    return best_labels.numpy(), best_C.numpy().astype(np.float32), log
