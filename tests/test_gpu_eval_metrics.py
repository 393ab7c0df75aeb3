"""acm_eval_metrics (ABI 28): the per-epoch evaluation of the reference's loops -- accuracy on every index set and the NLL on the
validation set from eval-mode logits (ACM-Geometric/train.py:138-140 + data_utils.py:153-168; ACM-Pytorch/train.py:112-139) -- as
one launch, against the torch ops it replaces: argmax ties, unlabeled rows (-1), every class count the models use, more than one
block of partials, bit-identical reruns, and train.EvalStep with and without it (eager and captured)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(out, labels, sets, loss_set):
    accs = [float((out.argmax(1)[idx] == labels[idx]).double().mean()) for idx in sets]
    o = F.log_softmax(out.double(), 1)
    return accs, float(F.nll_loss(o[sets[loss_set]], labels[sets[loss_set]]))


@pytest.mark.parametrize("n,c,k", [(2708, 7, 3), (168114, 2, 3), (5201, 5, 2), (300, 40, 1), (70000, 64, 8), (5, 3, 2)])
def test_metrics_match_the_torch_ops(n, c, k):
    from acm_gnn_amd import functional as AF
    g = torch.Generator().manual_seed(n + c)
    out = torch.randn(n, c, generator=g).to(DEV)
    out[::7] = out[::7].round()                            # ties between classes: the first maximum must win
    if n > 10:
        out[3] = 0.0
    labels = torch.randint(0, c, (n,), generator=g).to(DEV)
    perm = torch.randperm(n, generator=g).to(DEV)
    cut = [int(n * q / (k + 1)) for q in range(k + 1)]
    sets = [perm[cut[q]:max(cut[q + 1], cut[q] + 1)] for q in range(k)]
    labels[perm[cut[k]:]] = -1                              # rows outside every set: unlabeled (data_utils.rand_train_test_idx)
    w = torch.zeros(k, n, device=DEV)
    for q, idx in enumerate(sets):
        w[q, idx] = 1.0 / idx.numel()
    loss_set = k - 1
    res = AF.eval_metrics(out, labels, w, loss_set).cpu().double().numpy()
    accs, nll = _reference(out, labels.clamp_min(0), sets, loss_set)
    np.testing.assert_allclose(res[:k], accs, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(res[k], nll, rtol=5e-6)
    # same bits on every run, also with the workspace reused (the arrival counter resets itself)
    bufs = AF.eval_metrics_buffers(n, k, torch.device(DEV))
    a = AF.eval_metrics(out, labels, w, loss_set, bufs).clone()
    for _ in range(3):
        assert torch.equal(AF.eval_metrics(out, labels, w, loss_set, bufs), a)
    # a strided logits matrix (a column slice of a wider buffer)
    wide = torch.zeros(n, c + 3, device=DEV)
    wide[:, :c] = out
    assert torch.equal(AF.eval_metrics(wide[:, :c], labels, w, loss_set), a)


def test_bad_arguments_are_refused():
    from acm_gnn_amd import functional as AF
    out, y = torch.randn(10, 3, device=DEV), torch.zeros(10, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError, match="acm_eval_metrics"):
        AF.eval_metrics(out, y, torch.zeros(9, 10, device=DEV), 0)           # more than eight index sets
    with pytest.raises(RuntimeError, match="acm_eval_metrics"):
        AF.eval_metrics(torch.randn(10, 65, device=DEV), y, torch.zeros(1, 10, device=DEV), 0)
    with pytest.raises(ValueError):
        AF.eval_metrics(out, y, torch.zeros(2, 9, device=DEV), 0)


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_eval_step_with_and_without_the_fused_metrics(use_graph):
    from acm_gnn_amd import GCN, data as D, train as T
    from acm_gnn_amd.distributed import make_sharded_operators
    adj, x_np, y_np, (tr, va, te), _ = D.synthetic_dataset("tiny", seed=3)
    low, deg = D.build_filters(adj)
    ops = make_sharded_operators(low, deg, torch.device(DEV))
    x, y = torch.from_numpy(D.row_normalize_features(x_np)).to(DEV), torch.from_numpy(y_np).to(DEV)
    sets = tuple(torch.from_numpy(s).to(DEV) for s in (tr, va, te))
    torch.manual_seed(0)
    model = GCN(x.shape[1], 64, int(y.max()) + 1, 2, y.shape[0], 0.3, "acmgcnp", 0).to(DEV)
    a = T.EvalStep(model, x, ops, y, sets, use_graph=use_graph)
    b = T.EvalStep(model, x, ops, y, sets, use_graph=use_graph, fused_metrics=False)
    (oa, acc_a, la), (ob, acc_b, lb) = a(), b()
    assert a._metrics is not None and b._metrics is None
    torch.testing.assert_close(oa, ob, rtol=1e-5, atol=1e-6)       # (the second pass reuses the layer's cached P: another kernel form)
    np.testing.assert_allclose(acc_a, acc_b, rtol=1e-6, atol=2.0 / 64)
    np.testing.assert_allclose(la, lb, rtol=1e-4)
    # the metrics of THE SAME logits, both ways
    ref_acc = [float((oa.argmax(1)[s] == y[s]).double().mean()) for s in sets]
    np.testing.assert_allclose(acc_a, ref_acc, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(la, float(F.nll_loss(F.log_softmax(oa.double(), 1)[sets[1]], y[sets[1]])), rtol=5e-6)
    (_, acc_a2, la2) = a()                                  # (eager: this pass reuses the cached P -- logits to rounding)
    np.testing.assert_allclose(acc_a2, acc_a, atol=2.0 / 64)
    np.testing.assert_allclose(la2, la, rtol=1e-5)
    (_, acc_a3, la3) = a()
    assert acc_a3 == acc_a2 and la3 == la2                  # same kernels, same bits
