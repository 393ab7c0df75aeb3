"""The per-wave id streams (acm_csr_build_streams) as their one consumer walks them: the gather waves of the pipelined
first-layer backward (acm_conv_agg_bwd_t.next_agg; acm_stream_device.h: stream_gather_role) compute the NEXT step's
P = D^-1 (P dropout(x)) -- checked against a float64 scipy product of the same table on stream layouts with long rows cut
into pieces (combined by the last arriver; several steps in a row so that the self-resetting arrival counters are
exercised), isolated nodes, row counts that are not multiples of four and very few waves (many slices per wave)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, seed, hub_degree=0, empty_rows=0, mean_deg=40):
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


@pytest.mark.parametrize("n,hub,empty,lmax,n_waves", [
    (2003, 0, 0, 0, 0),               # short rows only, n % 4 != 0, the default layout
    (2501, 1400, 7, 64, 512),         # hubs cut into ~22 pieces each, isolated nodes
    (2501, 1400, 7, 32, 8),           # every row longer than 32 is cut; only 8 waves (many slices per wave)
    (4002, 900, 0, 128, 1000),        # as many waves as the workspace of the backward allows
])
def test_carried_gather_over_the_id_streams_matches_scipy(n, hub, empty, lmax, n_waves, tune):
    from acm_gnn_amd import GCN, FusedAdamW, data as D, functional as AF, train as T
    from acm_gnn_amd.distributed import make_sharded_operators
    tune(pipeline=1024, relabel=0)
    adj = _graph(n, seed=n, hub_degree=hub, empty_rows=empty)
    low, deg = D.build_filters(adj)
    ops = make_sharded_operators(low, deg, torch.device(DEV), relabel=False)
    assert ops.implicit
    if lmax or n_waves:                                   # (idempotent: the pipeline's own build_streams call keeps this layout)
        assert ops.low.build_streams(n_waves=n_waves, lmax=lmax)
        if hub and lmax:
            assert ops.low.stream_long_rows >= 3
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 7, generator=g).to(DEV)
    y = torch.randint(0, 2, (n,), generator=g).to(DEV)
    w = T.row_weights(torch.arange(0, n, 3, device=DEV), n)
    torch.manual_seed(0)
    model = GCN(7, 64, 2, 2, n, 0.25, "acmgcnp", 0, variant=False, attn_layernorm=True).to(DEV)
    model.dropout_state = AF.DropoutState(torch.device(DEV), seed=7)
    step = T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.01), x, ops, y, w, use_graph=False)
    assert step.pipe is not None and ops.low.stream_steps > 0
    pat = sp.csr_matrix((np.ones(len(ops.low.arrays()[1])), ops.low.arrays()[1].cpu().numpy(), ops.low.arrays()[0].cpu().numpy()),
                        shape=(n, n))
    rs = ops.row_scale.cpu().numpy().astype(np.float64)
    timer = AF.KernelTimer()
    AF.set_kernel_timer(timer)
    try:
        for _ in range(4):                                # arrival counters reset themselves between launches
            loss = float(step())
            assert np.isfinite(loss)
            table = step.pipe.table().cpu().numpy().astype(np.float64)
            want = rs[:, None] * (pat @ table)
            got = step.pipe.agg().cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6 * max(1.0, np.abs(want).max()))
    finally:
        AF.set_kernel_timer(None)
    assert any("+gather" in k for k in timer.events), sorted(timer.events)
