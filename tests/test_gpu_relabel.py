"""In-operator degree relabelling (graph.relabel_by_degree) on the MI355X: results in the caller's numbering are
unchanged -- at model level (one translation in, one out), at layer level (the drop-in route) and through the captured
training step, which moves its static inputs into the operator's numbering once."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from oracle import acm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, seed):
    rng = np.random.default_rng(seed)
    a = sp.random(n, n, density=0.03, random_state=rng, format="csr")
    a = ((a + a.T) > 0).astype(np.float64).tolil()
    a[7, :] = 1.0
    a[:, 7] = 1.0                                     # a hub in the middle of the id range
    a[3, 3] = 1.0
    return sp.csr_matrix(a)


@pytest.mark.parametrize("model_type,variant,s,f_in", [("acmgcnp", 0, 0, 7), ("acmgcnp", 0, 1, 7), ("acmgcnp", 1, 1, 7),
                                                       ("acmgcnpp", 1, 0, 40), ("acmsgc", 0, 0, 12)])
def test_relabelled_operators_give_the_same_results(model_type, variant, s, f_in, monkeypatch, tune):
    from acm_gnn_amd import GCN, graph, train as T
    n = 600
    low, high, un = (t.to(DEV) for t in O.filters_linkx(_graph(n, 4)))
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(n, f_in, generator=g).to(DEV), torch.randint(0, 3, (n,), generator=g).to(DEV)
    idx = torch.arange(0, n, 2, device=DEV)
    w = T.row_weights(idx, n)
    res = {}
    for mode in ("0", "1"):
        tune(relabel={"auto": -1}.get(mode, None) if mode == "auto" else int(mode))
        graph.clear_cache()
        ops = graph.operators_for(low, high, un if s else None)
        assert (ops.perm is not None) == (mode == "1")
        torch.manual_seed(2)
        model = GCN(f_in, 64, 3, 2, n, 0.0, model_type, s, variant=bool(variant), attn_layernorm=True).to(DEV)
        out = model(x, low, high, un)
        F.nll_loss(F.log_softmax(out, 1)[idx], y[idx]).backward()
        out = out.detach()          # drop the autograd graph: a live AccumulateGrad node pins the stream it was created on
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        att = model.gcns[0].att_high.clone()
        with torch.no_grad():
            layer_out = model.gcns[0](x, low, high, un if s else None)
        model.zero_grad(set_to_none=True)
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        losses = []
        for use_graph in (False, True):
            step = T.TrainStep(model, opt, x, low, y, w, high, un, use_graph=use_graph, fused_dropout=False)
            assert step._permuted == (mode == "1")
            losses.append(float(step()))
        res[mode] = (out, grads, att, layer_out, losses, {k: p.grad.clone() for k, p in model.named_parameters()
                                                                  if p.grad is not None})
    a, b = res["1"], res["0"]
    scale = max(1.0, float(b[0].abs().max()))
    assert float((a[0] - b[0]).abs().max()) < 2e-5 * scale
    assert float((a[2] - b[2]).abs().max()) < 2e-5 and float((a[3] - b[3]).abs().max()) < 2e-5 * max(1.0, float(b[3].abs().max()))
    for k, v in b[1].items():
        assert float((a[1][k] - v).abs().max()) < 1e-4 * max(1.0, float(v.abs().max())), k
    for k, v in b[5].items():
        assert float((a[5][k] - v).abs().max()) < 1e-4 * max(1.0, float(v.abs().max())), ("train step", k)
    assert max(abs(p - q) for p, q in zip(a[4], b[4])) < 2e-6 and abs(a[4][0] - a[4][1]) < 2e-6
