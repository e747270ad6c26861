"""CPU: the LNQ inner loops (guidedquant_amd/lnq.py: objective_function, update_P, update_C -- SURVEY.md section 8 f-4) against
golden vectors generated from the reference's own functions (tests/make_golden_lnq.py runs
any_precision/quantization/layerwise_quantize.py on the CPU), and the oracle's restatement of the inner loop against the
product's CPU path."""
import numpy as np
import pytest

from conftest import golden_files

torch = pytest.importorskip("torch")


def _load(path):
    g = np.load(path)
    return g, torch.tensor(g["W"]), torch.tensor(g["H"]), torch.tensor(g["labels"]), torch.tensor(g["C"])


@pytest.mark.parametrize("path", golden_files("lnq_"))
def test_objective_and_updates_match_the_reference(path):
    from guidedquant_amd import lnq
    g, W, H, labels, C = _load(path)
    assert float(lnq.objective_function(W, H, labels, C)) == pytest.approx(float(g["obj0"]), rel=1e-5)
    newl = lnq.update_P(W, H, labels, C, cd_cycles=int(g["cd_cycles"]), verbose=False)
    want = torch.tensor(g["labels_P"]).long()
    # the coordinate descent is sequential and discontinuous (argmin): a last-bit difference of a GEMM flips a near-tie and the
    # flip propagates, so equality is asked of the objective, agreement of the assignments
    assert float((newl == want).float().mean()) >= 0.999  # measured: identical on all three fixtures
    assert float(lnq.objective_function(W, H, newl, C)) == pytest.approx(float(g["obj1"]), rel=1e-5)
    assert float(lnq.objective_function(W, H, newl, C)) < float(g["obj0"])
    # centroid update on the REFERENCE's assignments: same least-squares solution
    newC = lnq.update_C(W, H, want, C, 0)
    scale = float(np.abs(g["C_new"]).max())
    assert float((newC - torch.tensor(g["C_new"])).abs().max()) <= 1e-5 * scale  # measured 4-6e-7
    assert float(lnq.objective_function(W, H, want, newC)) == pytest.approx(float(g["obj2"]), rel=1e-4)


def test_inner_block_cpu_path_equals_oracle(oracle):
    """the product's CPU restatement of the inner loop and the oracle's numpy one, bit for bit (2 groups, a 70-column tail block)"""
    from guidedquant_amd import lnq
    rng = np.random.default_rng(3)
    N, d, ncl, G = 64, 198, 8, 2
    W = rng.normal(0, 0.02, (N, d)).astype(np.float32)
    B = rng.normal(0, 0.004, (N, d)).astype(np.float32)
    Hn = rng.normal(0, 0.05, (G, d, d)).astype(np.float32)
    C = np.sort(rng.normal(0, 0.02, (N, ncl)).astype(np.float32), axis=1)
    for st, end in ((0, 128), (128, 198)):
        a_ref, w_ref = oracle.lnq_cd_block_np(W, B, Hn, C, N // G, st, end)
        assign = torch.zeros(N, d, dtype=torch.uint8)
        What = torch.zeros(N, d)
        lnq._cd_block(torch.tensor(W), torch.tensor(B), torch.tensor(Hn), torch.tensor(C), assign, What, N // G, st, end)
        assert np.array_equal(assign.numpy()[:, st:end], a_ref)
        assert np.array_equal(What.numpy()[:, st:end].view(np.uint32), w_ref.view(np.uint32))


def test_train_least_squares_improves_and_stops(tmp_path):
    from guidedquant_amd import lnq
    g = np.load(golden_files("lnq_b2")[0])
    labels, C, log = lnq.train_least_squares(g["W"], g["labels"], g["C"], g["H"], num_iterations=3, cd_cycles=2, device="cpu")
    assert labels.shape == g["labels"].shape and C.shape == g["C"].shape and C.dtype == np.float32
    assert log["objective"][-1] <= log["objective"][0] and min(log["objective"]) < 0.8 * log["objective"][0]
