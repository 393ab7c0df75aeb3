"""Degenerate shapes through the full layer on the MI355X: a single node, no edges at all (A_low = I), one
feature, one class, a star (one hub row longer than any chunk), rows that are exactly chunk-sized, and an
empty graph handle.  Each case is checked against the oracle, forward and backward, in both execution forms
where they apply."""
import numpy as np
import pytest
from conftest import tune_now as tune
import scipy.sparse as sp
import torch

from oracle import acm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(adj, f_in, f_out, model_type="acmgcnp", s=0, variant=0, x_grad=True, monkeypatch=None, agg=False):
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import clear_cache
    tune(agg_first=int(bool(agg)))
    clear_cache()
    n = adj.shape[0]
    low, high, un = O.filters_linkx(sp.csr_matrix(adj))
    torch.manual_seed(1)
    layer = GraphConvolution(f_in, f_out, n, model_type, variant=variant, structure_info=s, attn_layernorm=True)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    g = torch.Generator().manual_seed(2)
    x, gout = torch.randn(n, f_in, generator=g), torch.randn(n, f_out, generator=g)
    xr = x.clone().requires_grad_(x_grad)
    ref = O.layer_forward(params, xr, low, high, un if s else None, model_type=model_type, variant=variant,
                          structure_info=s, attn_layernorm=True)
    ref.backward(gout)
    layer = layer.to(DEV)
    xd = x.to(DEV).requires_grad_(x_grad)
    out = layer(xd, low.to(DEV), high.to(DEV), un.to(DEV) if s else None)
    out.backward(gout.to(DEV))
    assert torch.isfinite(out).all()
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 3e-5 * max(1.0, float(ref.detach().abs().max()))
    for k, p in layer.named_parameters():
        rg = params[k].grad
        if rg is None:
            assert p.grad is None, k
            continue
        assert float((p.grad.cpu() - rg).abs().max()) < 1e-4 * max(1.0, float(rg.abs().max())), k
    if x_grad:
        assert float((xd.grad.cpu() - xr.grad).abs().max()) < 1e-4 * max(1.0, float(xr.grad.abs().max()))


def test_single_node(monkeypatch):
    _check(np.zeros((1, 1)), 3, 4, monkeypatch=monkeypatch)
    _check(np.zeros((1, 1)), 3, 4, s=1, monkeypatch=monkeypatch)
    _check(np.zeros((1, 1)), 3, 64, x_grad=False, monkeypatch=monkeypatch, agg=True)


def test_no_edges_identity_filter(monkeypatch):
    adj = np.zeros((37, 37))
    for f_out in (1, 2, 5, 64, 70):
        _check(adj, 6, f_out, monkeypatch=monkeypatch)
    _check(adj, 6, 64, s=1, x_grad=False, monkeypatch=monkeypatch, agg=True)


def test_one_feature_one_class(monkeypatch):
    rng = np.random.default_rng(0)
    a = (rng.random((50, 50)) < 0.1).astype(float)
    a = np.maximum(a, a.T)
    _check(a, 1, 1, monkeypatch=monkeypatch)
    _check(a, 1, 1, model_type="acmgcn", variant=1, monkeypatch=monkeypatch)
    _check(a, 1, 2, x_grad=False, monkeypatch=monkeypatch, agg=True)


@pytest.mark.parametrize("chunk", [0, 256, 1024])
@pytest.mark.parametrize("n", [300, 1025, 3000])
def test_star_and_chunk_boundaries(n, chunk, monkeypatch, tune):
    """Node 0 is connected to everybody (a row of n entries, split into ceil(n / chunk) work items); a few rows have
    exactly chunk - 1 / chunk / chunk + 1 entries after the +I, for the automatic chunk of a small graph (128) and
    for the chunks larger graphs get (ACM_CHUNK pins them here)."""
    if chunk:
        tune(chunk=int(chunk))
    a = np.zeros((n, n))
    a[0, 1:] = 1
    a[1:, 0] = 1
    for r, deg in ((3, 126), (4, 127), (5, 128), (6, 254), (7, 255), (8, 256)):
        cols = np.arange(10, 10 + deg)
        a[r, cols] = 1
        a[cols, r] = 1
    for f_out, s in ((2, 0), (5, 1), (64, 0), (64, 1)):
        _check(a, 9, f_out, s=s, monkeypatch=monkeypatch)
    _check(a, 7, 64, s=1, x_grad=False, monkeypatch=monkeypatch, agg=True)


def test_empty_handle_and_zero_rows():
    from acm_gnn_amd import functional as AF
    from acm_gnn_amd.graph import CsrGraph
    ip = torch.zeros(1, dtype=torch.int32, device=DEV)
    g = CsrGraph.from_csr(ip, torch.zeros(0, dtype=torch.int32, device=DEV), torch.zeros(0, device=DEV), 5)
    assert g.n_rows == 0 and g.nnz == 0
    out = AF.spmm(g, torch.randn(5, 3, device=DEV))
    assert out.shape == (0, 3)
    ip = torch.zeros(8, dtype=torch.int32, device=DEV)                 # 7 rows, no entries
    g = CsrGraph.from_csr(ip, torch.zeros(0, dtype=torch.int32, device=DEV), None, 7)
    out = AF.spmm(g, torch.randn(7, 4, device=DEV))
    assert out.shape == (7, 4) and float(out.abs().max()) == 0.0
    assert AF.gemm(torch.zeros(0, 4, device=DEV), torch.zeros(4, 3, device=DEV)).shape == (0, 3)
    assert float(AF.gemm(torch.zeros(5, 0, device=DEV), torch.zeros(0, 3, device=DEV)).abs().max()) == 0.0
