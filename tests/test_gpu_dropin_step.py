"""The launcher's binding of ``utils.train_model`` (acm_gnn_amd.dropin.install_fused_train_step, round 6) on the GPU: the fused
small-graph step behind the reference's function signature against the reference's own loop body -- restated here, the
checkout does not travel (ACM-Pytorch/utils.py:547-574: model.train(), zero_grad, forward on the loader's dense adj_low / sparse
adj_high, log_softmax + NLLLoss on the training rows, accuracy, backward, optimizer.step()) -- on the Cora structure with its
real features.  The reference loop is fed the masks the kernels draw (tests/replay.PhiloxDropout in place of F.dropout) and
torch.optim.Adam, so the two runs are the same experiment: same losses, same training accuracies, same parameters."""
import os
import sys
import types

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from conftest import GOLDEN, load_npz
from replay import PhiloxDropout

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference_train_model(model, optimizer, adj_low, adj_high, adj_low_unnormalized, features, labels, idx_train, criterion, dataset_name):
    model.train()
    optimizer.zero_grad()
    output = F.log_softmax(model(features, adj_low, adj_high, adj_low_unnormalized), dim=1)
    loss_train = criterion(output[idx_train], labels[idx_train])
    acc_train = (output[idx_train].max(1)[1] == labels[idx_train]).double().sum() / labels[idx_train].shape[0]
    loss_train.backward()
    optimizer.step()
    return 100 * acc_train.item(), loss_train.item()


@pytest.mark.parametrize("mask_index", [False, True], ids=["index", "bool_mask"])
def test_bound_train_model_is_the_same_experiment(monkeypatch, mask_index):
    from acm_gnn_amd import GCN, dropin, functional as AF, layers
    from acm_gnn_amd.graph import clear_cache
    g = load_npz(os.path.join(GOLDEN, "graph_cora.npz"))
    n = int(g["n"])
    x = torch.from_numpy(sp.csr_matrix((g["feat_vals"], g["feat_indices"], g["feat_indptr"]), shape=(n, int(g["feat_dim"]))).toarray().astype(np.float32))
    a = sp.csr_matrix((np.ones(len(g["adj_un_indices"]), np.float32), g["adj_un_indices"], g["adj_un_indptr"]), shape=(n, n))
    a_un = torch.from_numpy(a.toarray())
    rowsum = (torch.eye(n) + a_un).sum(1)
    adj_low = torch.mm(torch.diag(torch.pow(rowsum, -1)), torch.eye(n) + a_un).to(DEV)
    adj_high = (torch.eye(n, device=DEV) - adj_low).to_sparse()
    xd, yd = x.to(DEV), torch.from_numpy(g["labels"]).to(DEV)
    tr_mask = torch.from_numpy(g["train_mask"].astype(bool))
    idx = tr_mask.to(DEV) if mask_index else tr_mask.nonzero().view(-1).to(DEV)
    crit = torch.nn.NLLLoss()
    seed, p = 0x5EED0ACC12345, 0.6
    monkeypatch.setattr(layers, "_default_device", lambda: torch.device("cpu"))

    def fresh(adam):
        clear_cache()
        torch.manual_seed(3)
        m = GCN(x.shape[1], 64, 7, 1, n, p, "acmgcn", 0, variant=False, attn_layernorm=False).to(DEV)
        return m, adam(m.parameters(), lr=0.01, weight_decay=5e-5)

    # (a) the reference's function with the kernels' masks injected, torch's Adam
    model_a, opt_a = fresh(torch.optim.Adam)
    drop = PhiloxDropout(seed, x)
    monkeypatch.setattr(F, "dropout", drop)
    ref_hist = []
    for _ in range(8):
        drop.next_epoch()
        ref_hist.append(_reference_train_model(model_a, opt_a, adj_low, adj_high, None, xd, yd, idx, crit, "cora"))
    monkeypatch.undo()
    monkeypatch.setattr(layers, "_default_device", lambda: torch.device("cpu"))
    # (b) the launcher's bindings: torch.optim.Adam -> FusedAdam, utils.train_model -> the fused small-graph step
    utils = types.ModuleType("utils")
    utils.train_model = _reference_train_model
    monkeypatch.setitem(sys.modules, "utils", utils)
    before = dropin.install_fused_optimizers()
    try:
        assert dropin.install_fused_train_step() is _reference_train_model
        model_b, opt_b = fresh(torch.optim.Adam)
        assert type(opt_b).__name__ == "FusedAdam"
        model_b.dropout_state = AF.DropoutState(torch.device(DEV), seed=seed)      # (the binding keeps a state it finds)
        model_b.fused_dropout = True
        got = [utils.train_model(model_b, opt_b, adj_low, adj_high, None, xd, yd, idx, crit, "cora") for _ in range(8)]
    finally:
        torch.optim.Adam, torch.optim.AdamW = before
    assert int(model_b.dropout_state.step.item()) == 8
    np.testing.assert_allclose([h[1] for h in got], [h[1] for h in ref_hist], rtol=2e-5)
    np.testing.assert_allclose([h[0] for h in got], [h[0] for h in ref_hist], atol=100.0 / int(tr_mask.sum()) + 1e-9)    # (one row at most)
    for (k, pa), (_, pb) in zip(model_a.state_dict().items(), model_b.state_dict().items()):
        torch.testing.assert_close(pb, pa, rtol=1e-3, atol=2e-5, msg=k)
    # outside the envelope the reference's own function runs, untouched (here: another criterion)
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        dropin._WARNED.clear()
        model_c, opt_c = fresh(torch.optim.Adam)
        r = utils.train_model(model_c, opt_c, adj_low, adj_high, None, xd, yd, idx, torch.nn.NLLLoss(reduction="sum"), "cora")
    assert any("train_model stays" in str(w.message) for w in caught) and np.isfinite(r[1])
