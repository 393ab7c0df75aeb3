"""Counter-based dropout on the MI355X: the kernels' masks equal the numpy restatement of
Philox4x32-7 bit for bit; layers with the in-register post-dropout equal the oracle fed that mask,
forward and backward, in every lane layout; a replayed hipGraph draws fresh masks."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import fake_lib
from oracle import acm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _factors(state, p, tag, n, c, row_offset=0):
    d = state.spec(p, tag, row_offset)
    step = int(state.step.item())

    class D:                                   # what fake_lib.dropout_factors reads, with the step by value
        pass
    dd = D()
    dd.p, dd.tag, dd.seed, dd.row_offset = d.p, d.tag, d.seed, d.row_offset
    host = np.array([step], np.int64)
    dd.step = host.ctypes.data
    out = fake_lib.dropout_factors(dd, n, c)
    del host
    return out


@pytest.mark.parametrize("n,c,pad,p,off", [(1000, 7, 8, 0.1, 0), (513, 64, None, 0.5, 0), (300, 130, None, 0.9, 12345),
                                           (4099, 1, None, 0.25, 0), (64, 16, 16, 0.0, 0), (10, 300, 304, 0.7, 2 ** 33)])
def test_dropout_kernel_equals_numpy_philox(n, c, pad, p, off):
    from acm_gnn_amd import functional as AF
    st = AF.DropoutState(DEV, seed=0xDEADBEEF12345678)
    st.step.fill_(3 + 2 ** 35)
    x = torch.randn(n, c, generator=torch.Generator().manual_seed(1))
    out = AF.dropout(x.to(DEV), p, st, tag=5, pad_to=pad, row_offset=off).cpu().numpy()
    f = _factors(st, p, 5, n, c, off).astype(np.float32) if p > 0 else np.ones((n, c), np.float32)
    assert out.shape == (n, pad or c)
    assert np.array_equal(out[:, :c], x.numpy() * f)
    if pad and pad > c:
        assert np.all(out[:, c:] == 0)
    if p > 0:
        assert abs((f > 0).mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / (n * c)) + 1e-3


CASES = [("acmgcnp", 0, 0, True, 7, 64, False, True),      # aggregate-first, grouped layout
         ("acmgcnp", 0, 1, True, 7, 64, False, True),      # + structure channel
         ("acmgcnp", 0, 0, True, 12, 24, False, True),
         # the sixteen-rows-per-wave kernels behind a fused ReLU + dropout (out-mask form): f_pad 4 / 8 / 16, three and four channels
         ("acmgcnp", 0, 0, True, 3, 64, False, True), ("acmgcnp", 0, 1, True, 4, 64, False, True),
         ("acmgcn", 0, 0, False, 12, 64, False, True), ("acmgcnp", 0, 1, True, 16, 64, False, True),
         ("acmgcnp", 0, 1, False, 8, 64, False, True),
         ("acmgcnp", 1, 1, True, 30, 64, True, False),     # literal, grouped backward (16 < F <= 64)
         ("acmgcn", 0, 0, False, 30, 5, True, False),      # literal, packed layout / row-parallel narrow forward
         ("acmgcnp", 0, 0, True, 30, 2, True, False),
         ("acmgcnp", 1, 1, True, 20, 100, True, False),    # wide layouts
         ("acmgcnp", 0, 0, True, 20, 130, True, False)]


@pytest.mark.parametrize("model_type,variant,s,ln,f_in,f_out,x_grad,agg", CASES)
def test_layer_with_in_register_dropout_matches_oracle(model_type, variant, s, ln, f_in, f_out, x_grad, agg, monkeypatch, tune):
    from acm_gnn_amd import GraphConvolution, functional as AF
    from acm_gnn_amd.graph import clear_cache
    tune(agg_first=int(bool(agg)))
    clear_cache()
    rng = np.random.default_rng(3)
    n, p = 500, 0.35
    a = sp.random(n, n, density=0.03, random_state=rng, format="csr")
    a = sp.csr_matrix(((a + a.T) > 0).astype(np.float64))
    low, high, un = O.filters_linkx(a)
    torch.manual_seed(4)
    layer = GraphConvolution(f_in, f_out, n, model_type, variant=variant, structure_info=s, attn_layernorm=ln)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    g = torch.Generator().manual_seed(5)
    x, gout = torch.randn(n, f_in, generator=g), torch.randn(n, f_out, generator=g)
    st = AF.DropoutState(DEV, seed=77)
    st.step.fill_(9)
    mask = torch.from_numpy(_factors(st, p, 1, n, f_out)).float()
    xr = x.clone().requires_grad_(x_grad)
    ref = O.layer_forward(params, xr, low, high, un if s else None, model_type=model_type, variant=variant,
                          structure_info=s, attn_layernorm=ln)
    ref = torch.relu(ref) * mask
    ref.backward(gout)
    layer = layer.to(DEV)
    xd = x.to(DEV).requires_grad_(x_grad)
    out = layer(xd, low.to(DEV), high.to(DEV), un.to(DEV) if s else None, post_relu=True, post_drop=(p, 1, st))
    out.backward(gout.to(DEV))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))
    assert abs(float((out == 0).float().mean()) - float((ref == 0).float().mean())) < 1e-6
    for k, prm in layer.named_parameters():
        rg = params[k].grad
        if rg is None:
            assert prm.grad is None, k
            continue
        tol = 1e-4 * max(1.0, float(rg.abs().max()))
        assert float((prm.grad.cpu() - rg).abs().max()) < tol, k
    if x_grad:
        assert float((xd.grad.cpu() - xr.grad).abs().max()) < 1e-4 * max(1.0, float(xr.grad.abs().max()))


def test_graph_replay_draws_fresh_masks_and_fused_adam_advances():
    from acm_gnn_amd import GCN, FusedAdamW, data as D, train as T
    from acm_gnn_amd.graph import CsrGraph, FilterOperators
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=1)
    low, _ = D.build_filters(adj)
    ops = FilterOperators(CsrGraph.from_scipy(low, DEV))
    x, y = torch.from_numpy(D.row_normalize_features(x_np)).to(DEV), torch.from_numpy(y_np).to(DEV)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), x.shape[0])
    runs = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        model = GCN(7, 64, 2, 2, x.shape[0], 0.5, "acmgcnp", 0).to(DEV)
        opt = FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
        if use_graph:
            state = {k: v.clone() for k, v in model.state_dict().items()}
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=use_graph)
        assert model.fused_dropout and opt.also_advance is model.dropout_state.step
        if use_graph:                       # the capture warm-up advanced weights, moments and the counter: rewind
            model.load_state_dict(state)
            for stt in opt.state.values():
                for v in stt.values():
                    v.zero_()
            model.dropout_state.step.zero_()
        runs.append([float(step()) for _ in range(6)])
        assert int(model.dropout_state.step.item()) == 6
        seen = set()
        for _ in range(3):                  # hidden activations differ from step to step: the mask moved
            step()
            seen.add(float(model.gcns[0].att_low.sum().item()))
        assert len(seen) == 3
    np.testing.assert_allclose(runs[1], runs[0], rtol=2e-5)      # same seed, same counters: same trajectory
    assert len(set(runs[0])) == 6
