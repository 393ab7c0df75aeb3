"""The streamed aggregate-first forward (acm_csr_build_streams + agg_stream_kernel) against the CSR-walking fused kernel
and the oracle: same operator, same inputs, the two kernels must agree to fp32 summation-order noise; long rows (pieces
combined by the last arriver, several launches in a row so that the self-resetting arrival counters are exercised),
empty rows, row counts that are not multiples of four, a wave count larger than the number of slices."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import acm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, seed, hub_degree=0, empty_rows=0, mean_deg=12):
    rng = np.random.default_rng(seed)
    deg = np.minimum((rng.pareto(1.3, n) * mean_deg / 3 + 1).astype(np.int64), n - 1)
    if hub_degree:
        deg[rng.integers(0, n, 3)] = min(hub_degree, n - 1)
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, rows.size)
    a = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.float64).tolil()
    a.setdiag(0)
    a = a.tocsr()
    if empty_rows:                                        # isolated nodes: after A + I they keep only the self loop
        iso = rng.choice(n, empty_rows, replace=False)
        keep = np.ones(n, bool)
        keep[iso] = False
        d = sp.diags(keep.astype(np.float64))
        a = (d @ a @ d).tocsr()
    a.eliminate_zeros()
    return a


def _run_layer(adj, f_in, f_out, seed, lmax, n_waves, streams, repeats=1, variant=0):
    from acm_gnn_amd import GraphConvolution, functional as AF
    from acm_gnn_amd.graph import clear_cache, operators_for
    os.environ["ACM_STREAMS"] = "1" if streams else "0"
    os.environ["ACM_RELABEL"] = "0"
    try:
        clear_cache()
        n = adj.shape[0]
        low, high, _ = O.filters_linkx(adj)
        torch.manual_seed(seed)
        layer = GraphConvolution(f_in, f_out, n, "acmgcnp", variant=variant, structure_info=0, attn_layernorm=True)
        params = {k: v.detach().cpu().clone() for k, v in layer.named_parameters()}
        layer = layer.to(DEV)
        x = torch.randn(n, f_in, generator=torch.Generator().manual_seed(seed + 1))
        lowd, highd = low.to(DEV), high.to(DEV)
        ops = operators_for(lowd, highd, None)
        if streams:
            assert ops.low.build_streams(n_waves=n_waves, lmax=lmax)
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        outs = []
        for _ in range(repeats):
            outs.append(layer(x.to(DEV), lowd, highd).detach().cpu())
        AF.set_kernel_timer(None)
        assert any(k.startswith("conv_agg_fwd") for k in timer.events), sorted(timer.events)
        info = (ops.low.stream_steps, ops.low.stream_waves, ops.low.stream_long_rows)
        ref = O.layer_forward({k: v.clone() for k, v in params.items()}, x, low, high, None, model_type="acmgcnp",
                              variant=variant, structure_info=0, attn_layernorm=True)
        return outs, ref.detach(), info
    finally:
        os.environ.pop("ACM_STREAMS", None)
        os.environ.pop("ACM_RELABEL", None)
        clear_cache()


@pytest.mark.parametrize("n,hub,empty,lmax,n_waves,f_out", [
    (403, 0, 0, 0, 0, 64),            # short rows only, n % 4 != 0
    (1501, 1400, 7, 64, 0, 64),       # hubs cut into ~22 pieces each, isolated nodes
    (1501, 1400, 7, 32, 8, 64),       # every row longer than 32 is cut; only 8 waves (many slices per wave)
    (2002, 900, 0, 128, 4096, 24),    # more waves than slices; F < 64 (column guards)
])
def test_stream_kernel_matches_csr_kernel_and_oracle(n, hub, empty, lmax, n_waves, f_out):
    adj = _graph(n, seed=n, hub_degree=hub, empty_rows=empty)
    got_s, ref, info = _run_layer(adj, 7, f_out, 3, lmax, n_waves, streams=True, repeats=3)
    got_c, _, info_c = _run_layer(adj, 7, f_out, 3, lmax, n_waves, streams=False)
    assert info[0] > 0 and info_c[0] == 0
    if hub and lmax:
        assert info[2] >= 3
    scale = max(1.0, float(ref.abs().max()))
    assert float((got_c[0] - ref).abs().max()) < 2e-5 * scale
    for o in got_s:
        assert float((o - ref).abs().max()) < 2e-5 * scale
        assert torch.equal(o, got_s[0])                       # arrival order does not change the sums
    assert float((got_s[0] - got_c[0]).abs().max()) < 1e-5 * scale


def test_stream_kernel_backward_matches_oracle():
    """The forward saves P = A_low X and the head statistics for the row-local backward: gradients through the streamed
    kernel against autograd through the oracle."""
    from acm_gnn_amd import GraphConvolution
    from acm_gnn_amd.graph import clear_cache
    adj = _graph(1203, seed=5, hub_degree=700)
    n = adj.shape[0]
    low, high, _ = O.filters_linkx(adj)
    os.environ["ACM_STREAMS"] = "1"
    os.environ["ACM_STREAM_LMAX"] = "64"
    try:
        clear_cache()
        torch.manual_seed(0)
        layer = GraphConvolution(7, 64, n, "acmgcnp", variant=0, structure_info=0, attn_layernorm=True)
        params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
        layer = layer.to(DEV)
        x = torch.randn(n, 7)
        gout = torch.randn(n, 64)
        xd = x.to(DEV).requires_grad_(True)
        out = layer(xd, low.to(DEV), high.to(DEV))
        out.backward(gout.to(DEV))
        xr = x.clone().requires_grad_(True)
        ref = O.layer_forward(params, xr, low, high, None, model_type="acmgcnp", variant=0, structure_info=0, attn_layernorm=True)
        ref.backward(gout)
        assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        assert float((xd.grad.cpu() - xr.grad).abs().max()) < 2e-4 * max(1.0, float(xr.grad.abs().max()))
        for k, prm in layer.named_parameters():
            if prm.grad is None:
                continue
            g = params[k].grad
            assert float((prm.grad.cpu() - g).abs().max()) < 3e-4 * max(1.0, float(g.abs().max())), k
    finally:
        os.environ.pop("ACM_STREAMS", None)
        os.environ.pop("ACM_STREAM_LMAX", None)
        clear_cache()
