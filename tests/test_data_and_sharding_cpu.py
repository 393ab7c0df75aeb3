"""Data preparation and the row-shard plan (host side), plus the 2-rank gloo run of the
sharded layer with the kernels replaced by the test double."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import ROOT
from oracle import acm_oracle as O


def test_synthetic_graph_shape_and_determinism():
    from acm_gnn_amd import data as D
    a1, x1, y1, s1, n1 = D.synthetic_dataset("tiny", seed=3)
    a2, x2, y2, s2, n2 = D.synthetic_dataset("tiny", seed=3)
    n, e, f_in, c = D.SHAPES["tiny"]
    assert a1.shape == (n, n) and a1.nnz == 2 * e and n1 == n and x1.shape == (n, f_in)
    assert (a1 != a1.T).nnz == 0 and a1.diagonal().sum() == 0 and set(np.unique(a1.data)) == {1.0}
    assert (a1 != a2).nnz == 0 and np.array_equal(x1, x2) and np.array_equal(y1, y2)
    assert sorted(np.concatenate(s1).tolist()) == list(range(n))
    deg = np.diff(a1.indptr)
    assert deg.max() > 8 * deg.mean()                                   # heavy tail
    a3, *_ = D.synthetic_dataset("tiny", seed=3, pad_to=7)
    assert a3.shape[0] % 7 == 0 and a3[n:].nnz == 0


def test_build_filters_matches_oracle_linkx_dialect():
    from acm_gnn_amd import data as D
    adj, x, *_ = D.synthetic_dataset("tiny", seed=1)
    adj = adj.tolil()
    adj[5, 5] = 1.0                                                    # raw self-loop (quirk Q5)
    adj = adj.tocsr()
    low, deg = D.build_filters(adj)
    ref_low, ref_high, _ = O.filters_linkx(adj)
    ip, ix, v = O.coo_to_csr_arrays(ref_low)
    assert np.array_equal(low.indptr, ip) and np.array_equal(low.indices, ix) and np.array_equal(low.data, v)
    assert deg[5] == adj[5].sum() + 1
    np.testing.assert_allclose(D.row_normalize_features(x), np.asarray(O.row_normalize_sp(sp.csr_matrix(x)).todense()),
                               rtol=1e-6)


def test_degree_order_is_an_isomorphism():
    from acm_gnn_amd import data as D
    adj, x, y, splits, _ = D.synthetic_dataset("tiny", seed=2)
    perm = D.degree_order(adj)
    a2, x2, y2, s2 = D.permute_dataset(adj, x, y, splits, perm)
    d = np.diff(a2.indptr)
    assert np.all(np.diff(d) <= 0)
    i, j = a2.nonzero()
    assert np.all(np.asarray(adj[perm[i], perm[j]]).ravel() == 1) and a2.nnz == adj.nnz
    assert np.array_equal(x2, x[perm]) and np.array_equal(y2, y[perm])
    assert np.array_equal(np.sort(perm[s2[0]]), splits[0])


def test_shard_plan_rows_of_a_and_a_transpose():
    from acm_gnn_amd import data as D, distributed as DD
    adj, *_ = D.synthetic_dataset("tiny", seed=4, pad_to=4)
    low, deg = D.build_filters(adj)
    n = low.shape[0]
    rows, rows_t = [], []
    for r in range(4):
        lo, lt, dg, off = DD.shard_filter_arrays(low, deg, 4, r)
        assert off == r * n // 4 and lo.shape == (n // 4, n) and lt.shape == (n // 4, n)
        assert np.array_equal(dg, deg[off:off + n // 4])
        rows.append(lo)
        rows_t.append(lt)
    assert (sp.vstack(rows) != low).nnz == 0 and (sp.vstack(rows_t) != low.T.tocsr()).nnz == 0
    with pytest.raises(ValueError):
        DD.shard_bounds(10, 4, 0)
    idx = np.array([0, 3, n // 4, n - 1])
    assert DD.local_index(idx, 4, 0, n).tolist() == [0, 3] and DD.local_index(idx, 4, 3, n).tolist() == [n // 4 - 1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cfg, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import fake_lib

    class MP:                                     # minimal monkeypatch for the child process
        def setattr(self, obj, name, val):
            setattr(obj, name, val)

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake_lib.install(MP())
        os.environ["ACM_IMPLICIT"] = str(cfg.get("implicit", 1))
        import torch.nn.functional as F
        from acm_gnn_amd import GCN, data as D, distributed as DD
        adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=5, pad_to=world)
        low, deg = D.build_filters(adj)
        n = adj.shape[0]
        ops = DD.make_sharded_operators(low, deg, "cpu", with_structure=bool(cfg["s"]))
        assert ops.sharded and ops.n_local == n // world and ops.implicit == bool(cfg.get("implicit", 1))
        ops.hops = cfg.get("hops", 1)
        b, e = DD.shard_bounds(n, world, rank)
        torch.manual_seed(0)
        pdrop = cfg.get("dropout", 0.0)
        full = GCN(7, 16, 2, 2, n, pdrop, cfg["model"], cfg["s"], variant=bool(cfg["variant"]), attn_layernorm=True)
        model = GCN(7, 16, 2, 2, e - b, pdrop, cfg["model"], cfg["s"], variant=bool(cfg["variant"]), attn_layernorm=True)
        if pdrop:                                       # counter-based dropout: every rank draws the global mask
            from acm_gnn_amd import functional as AF
            model.fused_dropout, model.dropout_state = True, AF.DropoutState("cpu", seed=7)
            if cfg.get("x_full"):
                ops.x_full = torch.from_numpy(x_np)
        sd = full.state_dict()
        for k in list(sd):
            if k.endswith(".struc_low"):
                sd[k] = sd[k][b:e].clone()
        model.load_state_dict(sd)
        x = torch.from_numpy(x_np[b:e])
        y = torch.from_numpy(y_np[b:e])
        idx = torch.from_numpy(DD.local_index(tr, world, rank, n))
        out = model(x, ops)
        loss = F.nll_loss(F.log_softmax(out, 1)[idx], y[idx], reduction="sum") / len(tr)
        loss.backward()
        tot = loss.detach().clone()
        dist.all_reduce(tot)
        grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
        ret.put((rank, out.detach().numpy().copy(), float(tot), grads))    # numpy: pickled by value
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [dict(model="acmgcnp", s=0, variant=0), dict(model="acmgcnp", s=1, variant=1),
                                 dict(model="acmgcn", s=0, variant=1), dict(model="acmgcnp", s=1, variant=0),
                                 dict(model="acmgcnp", s=1, variant=0, implicit=0),
                                 dict(model="acmgcnp", s=1, variant=1, implicit=0),
                                 dict(model="acmgcnp", s=0, variant=0, dropout=0.5),
                                 dict(model="acmgcnp", s=1, variant=0, dropout=0.5, x_full=1),
                                 dict(model="acmgcn", s=0, variant=1, dropout=0.5, x_full=1),
                                 dict(model="acmsgc", s=0, variant=0, hops=3),
                                 dict(model="acmsgc", s=0, variant=0, hops=2, implicit=0)],
                         ids=["agg+literal", "struct-acmii", "acmii", "struct-agg", "struct-agg-explicit",
                              "struct-acmii-explicit", "dropout", "dropout-xfull-struct", "dropout-xfull-acmii",
                              "sgc-3hop", "sgc-2hop-explicit"])
def test_two_rank_row_shard_equals_single_process(cfg, monkeypatch):
    """world_size = 2 over gloo: the sharded forward/backward (halo all-gathers + parameter-gradient
    all-reduce issued by functional.AcmConvFunction) must reproduce the 1-process result."""
    import torch.multiprocessing as mp
    import torch.nn.functional as F
    import fake_lib
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, ret)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    results, deadline = [], time.time() + 300
    while len(results) < world:
        try:
            results.append(ret.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                for p in procs:
                    p.terminate()
                pytest.fail(f"sharded workers failed (exit codes {[p.exitcode for p in procs]})")
    results.sort(key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference through the same host stack
    fake_lib.install(monkeypatch)
    monkeypatch.setenv("ACM_IMPLICIT", "0")                   # the single-process reference keeps explicit values
    from acm_gnn_amd import GCN, data as D, distributed as DD
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=5, pad_to=world)
    low, deg = D.build_filters(adj)
    n = adj.shape[0]
    ops = DD.make_sharded_operators(low, deg, "cpu", with_structure=bool(cfg["s"]))
    assert not ops.sharded and not ops.implicit
    ops.hops = cfg.get("hops", 1)
    torch.manual_seed(0)
    full = GCN(7, 16, 2, 2, n, cfg.get("dropout", 0.0), cfg["model"], cfg["s"], variant=bool(cfg["variant"]),
               attn_layernorm=True)
    if cfg.get("dropout"):
        from acm_gnn_amd import functional as AF
        full.fused_dropout, full.dropout_state = True, AF.DropoutState("cpu", seed=7)
    out = full(torch.from_numpy(x_np), ops)
    idx = torch.from_numpy(tr)
    loss = F.nll_loss(F.log_softmax(out, 1)[idx], torch.from_numpy(y_np)[idx], reduction="sum") / len(tr)
    loss.backward()
    got = torch.from_numpy(np.concatenate([r[1] for r in results]))
    torch.testing.assert_close(got, out.detach(), rtol=1e-5, atol=1e-6)
    assert abs(results[0][2] - loss.item()) < 1e-6
    half = n // world
    for k, p in full.named_parameters():
        if p.grad is None:
            continue
        for rank, _, _, grads in results:
            ref = p.grad[rank * half:(rank + 1) * half] if k.endswith(".struc_low") else p.grad
            torch.testing.assert_close(torch.from_numpy(grads[k]), ref, rtol=1e-4, atol=1e-6,
                                       msg=lambda m, k=k: f"{k}: {m}")
