"""GPU: the LNQ coordinate-descent block kernel (gq_lnq_cd_block, csrc/lnq.hip) bit for bit against the oracle's restatement of
any_precision/quantization/layerwise_quantize.py:93-118, and update_P / update_C / objective on the GPU against the
reference-generated goldens."""
import numpy as np
import pytest

from conftest import golden_files

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,d,ncl,G,blocks", [(128, 256, 4, 1, ((0, 128), (128, 256))), (256, 198, 8, 2, ((0, 128), (128, 198))),
                                              (96, 128, 16, 3, ((0, 128), )), (4096, 512, 4, 1, ((256, 384), )), (64, 40, 2, 2, ((0, 40), ))])
def test_cd_block_kernel_bit_exact(oracle, N, d, ncl, G, blocks):
    from guidedquant_amd import lnq
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(N + d + ncl)
    W = rng.normal(0, 0.02, (N, d)).astype(np.float32)
    B = rng.normal(0, 0.004, (N, d)).astype(np.float32)
    Hn = rng.normal(0, 0.05, (G, d, d)).astype(np.float32)
    C = np.sort(rng.normal(0, 0.02, (N, ncl)).astype(np.float32), axis=1)
    C[::7, 1] = C[::7, 0]  # exact ties between centroids: the lowest index wins
    Wt, Bt, Ht, Ct = (torch.tensor(a, device=dev) for a in (W, B, Hn, C))
    for st, end in blocks:
        assign = torch.full((N, d), 255, dtype=torch.uint8, device=dev)
        What = torch.full((N, d), float("nan"), device=dev)
        lnq._cd_block(Wt, Bt, Ht, Ct, assign, What, N // G, st, end)
        torch.cuda.synchronize()
        a_ref, w_ref = oracle.lnq_cd_block_np(W, B, Hn, C, N // G, st, end)
        assert np.array_equal(assign.cpu().numpy()[:, st:end], a_ref)
        assert np.array_equal(What.cpu().numpy()[:, st:end].view(np.uint32), w_ref.view(np.uint32))
        assert bool((assign[:, :st] == 255).all()) and bool((assign[:, end:] == 255).all())  # nothing outside the block is written
        assert torch.equal(Bt.cpu(), torch.tensor(B))  # B is read-only for the kernel


@pytest.mark.parametrize("path", golden_files("lnq_"))
def test_lnq_updates_on_gpu_match_the_reference(path):
    from guidedquant_amd import lnq
    dev = torch.device("cuda:0")
    g = np.load(path)
    W, H = torch.tensor(g["W"], device=dev), torch.tensor(g["H"], device=dev)
    labels, C = torch.tensor(g["labels"]), torch.tensor(g["C"])
    assert float(lnq.objective_function(W, H, labels, C)) == pytest.approx(float(g["obj0"]), rel=1e-5)
    newl = lnq.update_P(W, H, labels, C, cd_cycles=int(g["cd_cycles"]), verbose=False)
    want = torch.tensor(g["labels_P"]).long()
    # GPU GEMMs sum in another order than the CPU run that made the golden: near-ties of the argmin may flip
    assert float((newl.cpu() == want).float().mean()) >= 0.995
    assert float(lnq.objective_function(W, H, newl, C)) == pytest.approx(float(g["obj1"]), rel=1e-3)
    newC = lnq.update_C(W, H, want, C, 0)
    assert float((newC - torch.tensor(g["C_new"])).abs().max()) <= 1e-4 * float(np.abs(g["C_new"]).max())
    assert float(lnq.objective_function(W, H, want, newC)) == pytest.approx(float(g["obj2"]), rel=1e-4)


def test_update_P_gpu_equals_cpu_path_at_scale():
    """size-independent property on a layer-sized slice: the fused-kernel path (GPU) and the torch restatement (CPU) reach the same
    objective, and the objective never increases over the cycles"""
    from guidedquant_amd import lnq
    rng = np.random.default_rng(0)
    N, d, ncl = 512, 1024, 4
    W = rng.normal(0, 0.02, (N, d)).astype(np.float32)
    X = rng.normal(0, 1, (2 * d, d)).astype(np.float32)
    H = (X.T @ X / (2 * d) + 1e-2 * np.eye(d)).astype(np.float32)[None]
    C = np.quantile(W, (np.arange(ncl) + 0.5) / ncl, axis=1).T.astype(np.float32)
    labels = np.abs(W[:, :, None] - C[:, None, :]).argmin(-1).astype(np.int8)
    Wc, Hc, lc, Cc = torch.tensor(W), torch.tensor(H), torch.tensor(labels), torch.tensor(C)
    o0 = float(lnq.objective_function(Wc, Hc, lc, Cc))
    cpu = lnq.update_P(Wc, Hc, lc, Cc, 2, verbose=False)
    gpu = lnq.update_P(Wc.cuda(), Hc.cuda(), lc, Cc, 2, verbose=False)
    oc, og = float(lnq.objective_function(Wc, Hc, cpu, Cc)), float(lnq.objective_function(Wc, Hc, gpu.cpu(), Cc))
    assert oc < o0 and og < o0 and og == pytest.approx(oc, rel=2e-3)
    assert float((cpu == gpu.cpu()).float().mean()) >= 0.99
