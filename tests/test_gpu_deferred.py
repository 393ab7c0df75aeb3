"""Deferred second phases on the GPU (acm_reduce_list_t / acm_reduce_flush): bit-identical to the immediate form,
for every op that has one, in eager and graph-replayed training steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _setup(model_type, s, hidden, f_in=None, dataset="tiny", seed=3, dropout=0.0):
    from acm_gnn_amd import GCN, data as D, train as T
    from oracle import acm_oracle as O
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset(dataset, seed=seed)
    ops = tuple(m.to(DEV) for m in O.filters_linkx(adj))          # (A_low, A_high, A) as the reference passes them
    x = torch.from_numpy(D.row_normalize_features(x_np)).to(DEV)
    if f_in:
        x = torch.cat([x] * (f_in // x.shape[1] + 1), 1)[:, :f_in].contiguous()
    y = torch.from_numpy(y_np).to(DEV)
    w = T.row_weights(torch.from_numpy(tr), x.shape[0]).to(DEV)
    torch.manual_seed(0)
    model = GCN(x.shape[1], hidden, int(y.max()) + 1, 2, x.shape[0], dropout, model_type, s).to(DEV)
    return model, ops, x, y, w


@pytest.mark.parametrize("model_type,s,hidden,f_in", [("acmgcn", 0, 64, None), ("acmgcnp", 0, 64, None),
                                                      ("acmgcnp", 1, 64, None), ("acmgcnp", 0, 16, 40),
                                                      ("acmgcnpp", 1, 32, 100)])
def test_deferred_equals_immediate_bitwise(model_type, s, hidden, f_in):
    from acm_gnn_amd import functional as AF
    model, ops, x, y, w = _setup(model_type, s, hidden, f_in)
    model(x, *ops)                  # (no input dropout: the first pass leaves P = A_low X for every later pass over this x)
    out = model(x, *ops)
    loss0, dz = AF.nll_loss_and_grad(out, y, w)
    out.backward(dz)
    want = {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    model.zero_grad(set_to_none=True)
    with AF.deferred_reductions() as pending:
        out = model(x, *ops)
        loss, dz = AF.nll_loss_and_grad(out, y, w)
        out.backward(dz)
        assert pending.pending >= 3
        assert pending.all_adopted([loss] + [p.grad for p in model.parameters()])
        pending.flush()
    assert float(loss) == float(loss0) and np.isfinite(float(loss))
    got = {k: v.grad for k, v in model.named_parameters() if v.grad is not None}
    assert got.keys() == want.keys()
    # Under a deferral list the two-class output layer of the three-channel models leaves its projection backward to the
    # hidden layer's kernel (acm_conv_agg_bwd_t.proj_*: another summation order for that layer's dW and for what flows on);
    # everything else is the same launches with the second phases postponed: bit-identical
    lazy = model_type != "acmgcnpp" and hidden == 64 and int(y.max()) + 1 <= 2
    for k in want:
        if lazy:
            torch.testing.assert_close(got[k], want[k], rtol=2e-4, atol=2e-5 * float(want[k].abs().max()), msg=lambda m, k=k: f"{k}: {m}")
        else:
            assert torch.equal(got[k], want[k]), k


@pytest.mark.parametrize("use_graph", [False, True])
def test_train_step_trajectory_is_the_same_with_and_without_deferral(use_graph, monkeypatch):
    from acm_gnn_amd import FusedAdamW, train as T
    params = {}
    for defer in (True, False):
        model, ops, x, y, w = _setup("acmgcnp", 1, 64, dropout=0.3)
        opt = FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
        if not defer:
            monkeypatch.setattr(T.TrainStep, "_forward_backward", _immediate)
        model.dropout_state = None
        step = T.TrainStep(model, opt, x, ops[0], y, w, ops[1], ops[2], use_graph=use_graph)
        losses = [float(step()) for _ in range(6)]
        assert step._defer == defer or not defer
        params[defer] = (losses, [p.detach().clone() for p in model.parameters()])
        monkeypatch.undo()
    # (the deferred step leaves the output layer's projection backward to the hidden layer's kernel -- proj_*, only under a
    #  deferral list since round 4 -- so the two trajectories agree to rounding, not bit for bit)
    np.testing.assert_allclose(params[True][0], params[False][0], rtol=2e-5)
    for a, b in zip(params[True][1], params[False][1]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-5)


def _immediate(self, for_optimizer=False):
    from acm_gnn_amd import functional as AF
    out = self.model(self.x, self.adj, self.adj_high, self.adj_un)
    loss, dz = AF.nll_loss_and_grad(out, self.labels, self.weights)
    out.backward(dz)
    return loss


@pytest.mark.parametrize("s,ln", [(0, True), (0, False), (1, True)])
def test_saved_head_statistics_equal_the_recomputation_bitwise(s, ln, monkeypatch):
    """acm_conv_agg_fwd hands its row statistics (mean | rstd | sigmoid | alpha) to acm_conv_agg_bwd; without them the
    backward recomputes the same numbers with the same code, so every gradient agrees bit for bit."""
    from acm_gnn_amd import GraphConvolution, functional as AF, data as D
    from oracle import acm_oracle as O
    adj, x_np, _, _, _ = D.synthetic_dataset("tiny", seed=5)
    low, high, un = (m.to(DEV) for m in O.filters_linkx(adj))
    n = adj.shape[0]
    torch.manual_seed(1)
    layer = GraphConvolution(7, 64, n, "acmgcnp", structure_info=s, attn_layernorm=ln).to(DEV)
    x = torch.from_numpy(x_np[:, :7].astype(np.float32)).to(DEV).requires_grad_(True)
    gout = torch.randn(n, 64, device=DEV)
    grads = []
    for use_stats in (True, False):
        if not use_stats:
            real = AF._lib.ConvAggBwd

            class NoStats(real):                       # same struct; the stats pointer is dropped before the call
                def __setattr__(self, k, v):
                    if k not in ("head_stats", "ld_head_stats"):
                        super().__setattr__(k, v)
            monkeypatch.setattr(AF._lib, "ConvAggBwd", NoStats)
        layer.zero_grad(set_to_none=True)
        x.grad = None
        out = layer(x, low, high, un if s else None)
        out.backward(gout)
        grads.append({k: v.grad.clone() for k, v in layer.named_parameters() if v.grad is not None} | {"x": x.grad.clone()})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) >= 8
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


@pytest.mark.parametrize("model_type,hidden,classes_graph", [("acmgcn", 64, "tiny"), ("acmgcnp", 64, "tiny"), ("acmgcnp", 16, "tiny")])
def test_fused_loss_tail_equals_the_three_calls(model_type, hidden, classes_graph):
    """acm_conv_fwd_tail (row phase + masked NLL + K3 in one kernel) against acm_conv_fwd, acm_nll_loss and
    acm_conv_bwd_local: logits, dz and the G tables bit for bit; the loss and the parameter-gradient sums up to the
    different (fixed) order of their block partials."""
    from acm_gnn_amd import functional as AF
    model, ops, x, y, w = _setup(model_type, 0, hidden, dataset=classes_graph)
    model(x, *ops)                  # (no input dropout: the first pass leaves P = A_low X for every later pass over this x)
    out0 = model(x, *ops)
    loss0, dz0 = AF.nll_loss_and_grad(out0, y, w)
    out0.backward(dz0)
    want = {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    model.zero_grad(set_to_none=True)
    with AF.fused_loss_tail(y, w) as tail:
        out = model(x, *ops)
    assert tail.matches(out), "the output layer did not take the request"
    assert torch.equal(out, out0) and torch.equal(tail.dz, dz0)
    np.testing.assert_allclose(float(tail.loss), float(loss0), rtol=2e-6)
    out.backward(tail.dz)
    got = {k: v.grad for k, v in model.named_parameters() if v.grad is not None}
    assert got.keys() == want.keys()
    for k in want:
        scale = float(want[k].abs().max()) + 1e-12
        assert float((got[k] - want[k]).abs().max()) <= 3e-6 * scale, k
    # a different gradient: the layer's own K3 runs again
    model.zero_grad(set_to_none=True)
    with AF.fused_loss_tail(y, w) as tail:
        out = model(x, *ops)
    out.backward(tail.dz * 3.0)
    for k, v in model.named_parameters():
        if v.grad is not None:
            scale = float(want[k].abs().max()) + 1e-12
            assert float((v.grad - 3.0 * want[k]).abs().max()) <= 1e-5 * scale, k
