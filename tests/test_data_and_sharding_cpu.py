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
    plan = DD.equal_rows_plan(n, 4)
    assert plan.uniform and plan.n_gathered == n and plan.n_max == n // 4
    rows, rows_t = [], []
    for r in range(4):
        lo, lt, dg, off = DD.shard_filter_arrays(low, deg, plan, r)
        assert off == r * n // 4 and lo.shape == (n // 4, n) and lt.shape == (n // 4, n)
        assert np.array_equal(dg, deg[off:off + n // 4])
        rows.append(lo)
        rows_t.append(lt)
    assert (sp.vstack(rows) != low).nnz == 0 and (sp.vstack(rows_t) != low.T.tocsr()).nnz == 0
    with pytest.raises(ValueError):
        DD.shard_bounds(10, 4, 0)
    idx = np.array([0, 3, n // 4, n - 1])
    assert DD.local_index(idx, 4, 0, n).tolist() == [0, 3] and DD.local_index(idx, 4, 3, n).tolist() == [n // 4 - 1]
    assert DD.local_index(idx, plan, 3).tolist() == [n // 4 - 1]


def _powerlaw(n=24_000, e=480_000, seed=3):
    from acm_gnn_amd import data as D
    adj = D.chung_lu_graph(n, e, 6_000, seed=seed)
    perm = D.degree_order(adj)
    return adj, adj[perm][:, perm].tocsr()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_work_balanced_shard_plan(world):
    """acm_shard_plan (the real host routine of libacm_hip.so: no device needed) on a degree-ordered power-law graph,
    the case where equal rows fail (rank 0 gets the hubs): pure nnz balance to 5 %, the row-cost variant balances
    nnz + row_cost * rows, cuts are monotone, the numpy test double computes the same plan."""
    import ctypes as C
    import fake_lib
    from acm_gnn_amd import data as D, distributed as DD
    _, adj = _powerlaw()
    low, _ = D.build_filters(adj)
    n = low.shape[0]
    eq = DD.equal_rows_plan(n - n % world, world)
    ip_eq = low.indptr[: n - n % world + 1]
    assert eq.imbalance(ip_eq)[0] > 1.5                                 # what round 1 shipped: >= 1.5x on rank 0
    plan = DD.shard_plan(low.indptr, world, row_cost=0)
    assert plan.world == world and plan.n_global == n and np.all(np.diff(plan.bounds) > 0)
    nnz_ratio, _ = plan.imbalance(low.indptr)
    assert nnz_ratio <= 1.05, nnz_ratio
    plan_w = DD.shard_plan(low.indptr, world, row_cost=64)
    assert plan_w.imbalance(low.indptr, 64)[1] <= 1.05
    assert not plan.uniform and plan.n_gathered == world * plan.n_max
    fake = fake_lib.FakeLib()
    for rc in (0, 64):
        ip = np.ascontiguousarray(low.indptr, dtype=np.int64)
        out = np.zeros(world + 1, np.int64)
        assert fake.acm_shard_plan(n, ip.ctypes.data_as(C.c_void_p), world, rc, out.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(out, DD.shard_plan(low.indptr, world, rc).bounds)
    # padded halo numbering: a bijection onto the used rows of the gathered table, monotone inside a block
    ids = plan.padded_ids(np.arange(n))
    assert len(np.unique(ids)) == n and ids.max() < plan.n_gathered
    for r in range(world):
        b, e = plan.rows(r)
        assert np.array_equal(ids[b:e], r * plan.n_max + np.arange(e - b))
    padded = plan.pad_rows(np.arange(n, dtype=np.float32)[:, None])
    assert padded.shape == (plan.n_gathered, 1) and np.array_equal(padded[ids, 0], np.arange(n))


def test_shard_plan_degenerate_inputs():
    from acm_gnn_amd import distributed as DD
    one_hub = np.array([0, 1000, 1001, 1002, 1003], dtype=np.int64)     # one row heavier than a share
    plan = DD.shard_plan(one_hub, 4, row_cost=0)
    assert plan.bounds[0] == 0 and plan.bounds[-1] == 4 and np.all(np.diff(plan.bounds) >= 0)
    assert (0, 1) in [plan.rows(r) for r in range(4)]                   # the hub row has a block of its own
    empty = DD.shard_plan(np.zeros(1, np.int64), 3, row_cost=5)
    assert empty.bounds.tolist() == [0, 0, 0, 0]
    assert DD.shard_plan(np.arange(9, dtype=np.int64), 1).bounds.tolist() == [0, 8]
    with pytest.raises(RuntimeError):
        DD.shard_plan(np.arange(1, 10, dtype=np.int64), 2)              # indptr[0] != 0


def test_interleaved_degree_order_balances_equal_blocks():
    """Dealing the degree ranking to the ranks like cards makes EQUAL contiguous blocks balanced in rows and nnz at
    once (what bench.py does for --node-order degree on several GPUs)."""
    from acm_gnn_amd import data as D, distributed as DD
    _, adj = _powerlaw()
    n = adj.shape[0]
    for world in (2, 4, 8):
        perm = DD.interleave_order(n, world)
        assert sorted(perm.tolist()) == list(range(n))
        a2 = adj[perm][:, perm].tocsr()
        low, _ = D.build_filters(a2)
        plan = DD.equal_rows_plan(n, world)
        nnz_ratio, _ = plan.imbalance(low.indptr)
        assert nnz_ratio <= 1.02, (world, nnz_ratio)
        for r in range(world):                                          # every block is itself sorted by degree
            b, e = plan.rows(r)
            assert np.all(np.diff(np.diff(a2.indptr)[b:e]) <= 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _lib_now():
    from acm_gnn_amd import _lib
    return _lib.load()


def _case_dataset(cfg, world):
    """(adj, x, y, train idx) of a sharding case: the tiny graph (7 features), or -- ``wide=F_in`` -- a random graph with 8 192
    rows per rank, mean degree 24 and F_in dense features: the regime of the wide aggregate-first layer (functional._AcmAggWide)."""
    from acm_gnn_amd import data as D
    if not cfg.get("wide"):
        adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=5, pad_to=world)
        return adj, x_np, y_np, tr
    import scipy.sparse as sp
    n, rng = 8192 * world, np.random.default_rng(11)
    m = n * 12
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    adj = sp.csr_matrix((np.ones(m, np.float32), (r, c)), shape=(n, n))
    adj = ((adj + adj.T) > 0).astype(np.float32).tocsr()
    adj.setdiag(0)
    adj.eliminate_zeros()
    x_np = rng.standard_normal((n, int(cfg["wide"]))).astype(np.float32)
    y_np = rng.integers(0, 2, n).astype(np.int64)
    return adj, x_np, y_np, np.sort(rng.permutation(n)[: n // 2])


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
        from acm_gnn_amd import tuning
        tuning.HOST.implicit = int(cfg.get("implicit", 1))          # (a child process of its own: nothing to restore)
        import torch.nn.functional as F
        from acm_gnn_amd import GCN, data as D, distributed as DD
        adj, x_np, y_np, tr = _case_dataset(cfg, world)
        f_in = x_np.shape[1]
        if cfg.get("plan") == "work":                 # hubs first: the case equal blocks cannot balance
            adj, x_np, y_np, (tr, _, _) = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), D.degree_order(adj))
        low, deg = D.build_filters(adj)
        n = adj.shape[0]
        # "rows": equal blocks; "work": the nnz-balanced plan of acm_shard_plan (blocks of different lengths: padded halo)
        plan = DD.equal_rows_plan(n, world) if cfg.get("plan", "rows") == "rows" else DD.shard_plan(low.indptr, world, 8)
        ops = DD.make_sharded_operators(low, deg, "cpu", with_structure=bool(cfg["s"]), plan=plan)
        b, e = plan.rows(rank)
        assert ops.sharded and ops.n_local == e - b and ops.implicit == bool(cfg.get("implicit", 1))
        assert ops.low.n_cols == plan.n_gathered and ops.uniform == plan.uniform
        ops.hops = cfg.get("hops", 1)
        torch.manual_seed(0)
        pdrop = cfg.get("dropout", 0.0)
        hid = cfg.get("hid", 16)
        full = GCN(f_in, hid, 2, 2, n, pdrop, cfg["model"], cfg["s"], variant=bool(cfg["variant"]), attn_layernorm=True)
        model = GCN(f_in, hid, 2, 2, e - b, pdrop, cfg["model"], cfg["s"], variant=bool(cfg["variant"]), attn_layernorm=True)
        if pdrop:                                       # counter-based dropout: every rank draws the global mask
            from acm_gnn_amd import functional as AF
            model.fused_dropout, model.dropout_state = True, AF.DropoutState("cpu", seed=7)
            if cfg.get("x_full"):
                ops.x_full = torch.from_numpy(x_np)       # honoured with equal blocks only
        sd = full.state_dict()
        for k in list(sd):
            if k.endswith(".struc_low"):
                sd[k] = sd[k][b:e].clone()
        model.load_state_dict(sd)
        x = torch.from_numpy(x_np[b:e])
        y = torch.from_numpy(y_np[b:e])
        idx = torch.from_numpy(DD.local_index(tr, plan, rank))
        calls = []
        if cfg.get("wide"):
            fake = _lib_now()
            for name in ("acm_conv_aggw_fwd", "acm_conv_bwd_spmm"):
                fake.__dict__[name] = (lambda o, nm: lambda *a: (calls.append(nm), o(*a))[1])(getattr(fake, name), name)
        out = model(x, ops)
        loss = F.nll_loss(F.log_softmax(out, 1)[idx], y[idx], reduction="sum") / len(tr)
        loss.backward()
        if cfg.get("wide"):                             # the first layer ran in the wide aggregate-first form on every rank: one
            assert calls.count("acm_conv_aggw_fwd") == 1 and calls.count("acm_conv_bwd_spmm") == 1, calls     # transposed gather left
        tot = loss.detach().clone()
        dist.all_reduce(tot)
        grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
        ret.put((rank, out.detach().numpy().copy(), float(tot), grads, (b, e)))    # numpy: pickled by value
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
                                 dict(model="acmsgc", s=0, variant=0, hops=2, implicit=0),
                                 dict(model="acmgcnp", s=1, variant=0, plan="work"),
                                 dict(model="acmgcnp", s=1, variant=1, plan="work", implicit=0),
                                 dict(model="acmgcnp", s=0, variant=0, dropout=0.5, x_full=1, plan="work"),
                                 dict(model="acmgcnp", s=1, variant=1, world=4),
                                 dict(model="acmgcnp", s=1, variant=0, world=4, plan="work", dropout=0.5),
                                 dict(model="acmsgc", s=0, variant=0, hops=3, world=4, plan="work"),
                                 dict(model="acmgcnpp", s=0, variant=0, dropout=0.5),
                                 dict(model="acmgcnpp", s=1, variant=1, world=4, plan="work"),
                                 dict(model="acmgcnp", s=0, variant=1, hid=64, dropout=0.5, x_full=1),
                                 dict(model="acmgcnp", s=1, variant=1, hid=64, plan="work"),
                                 dict(model="acmgcnp", s=1, variant=0, world=8, plan="work", dropout=0.5),
                                 dict(model="acmsgc", s=0, variant=0, hops=3, world=8, plan="work"),
                                 dict(model="acmgcnp", s=0, variant=0, world=8, dropout=0.5, x_full=1),
                                 dict(model="acmgcnp", s=0, variant=0, hid=64, wide=40, dropout=0.3),
                                 dict(model="acmgcn", s=0, variant=0, hid=64, wide=65, plan="work")],
                         ids=["agg+literal", "struct-acmii", "acmii", "struct-agg", "struct-agg-explicit",
                              "struct-acmii-explicit", "dropout", "dropout-xfull-struct", "dropout-xfull-acmii",
                              "sgc-3hop", "sgc-2hop-explicit", "work-plan-struct-agg", "work-plan-acmii-explicit",
                              "work-plan-dropout", "4-ranks-struct-acmii", "4-ranks-work-plan-dropout",
                              "4-ranks-work-plan-sgc-3hop", "acmgcnpp-dropout", "4-ranks-work-plan-acmgcnpp",
                              "acmii-recompute-dropout-xfull", "acmii-recompute-struct-work-plan",
                              "8-ranks-work-plan-struct-dropout", "8-ranks-work-plan-sgc-3hop", "8-ranks-equal-blocks-xfull",
                              "wide-aggregate-first-dropout", "wide-aggregate-first-work-plan"])
def test_row_shard_equals_single_process(cfg, monkeypatch, tune):
    """world_size = 2, 4 and 8 (the node's GPU count) over gloo: the sharded forward/backward (halo all-gathers + parameter-gradient
    all-reduce issued by functional.AcmConvFunction) must reproduce the 1-process result -- with equal blocks and
    with the work-balanced plan (blocks of different lengths, padded halo numbering)."""
    import torch.multiprocessing as mp
    import torch.nn.functional as F
    import fake_lib
    world, port = cfg.get("world", 2), _free_port()
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
    tune(implicit=0)                   # the single-process reference keeps explicit values
    from acm_gnn_amd import GCN, data as D, distributed as DD
    adj, x_np, y_np, tr = _case_dataset(cfg, world)
    if cfg.get("plan") == "work":
        adj, x_np, y_np, (tr, _, _) = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), D.degree_order(adj))
    low, deg = D.build_filters(adj)
    n = adj.shape[0]
    if cfg.get("wide"):
        tune(agg_first=0)                # ... and the literal project-then-gather form
    ops = DD.make_sharded_operators(low, deg, "cpu", with_structure=bool(cfg["s"]))
    assert not ops.sharded and not ops.implicit
    if cfg.get("plan") == "work":
        assert len({r[4][1] - r[4][0] for r in results}) > 1               # the blocks really differ in length
    ops.hops = cfg.get("hops", 1)
    torch.manual_seed(0)
    full = GCN(x_np.shape[1], cfg.get("hid", 16), 2, 2, n, cfg.get("dropout", 0.0), cfg["model"], cfg["s"], variant=bool(cfg["variant"]),
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
    for k, p in full.named_parameters():
        if p.grad is None:
            continue
        for rank, _, _, grads, (b, e) in results:
            ref = p.grad[b:e] if k.endswith(".struc_low") else p.grad
            torch.testing.assert_close(torch.from_numpy(grads[k]), ref, rtol=1e-4, atol=1e-6,
                                       msg=lambda m, k=k: f"{k}: {m}")


def _train_dataset(cfg, world):
    """("tiny" synthetic dataset) or, for the input pipeline's regime (12 < nnz / n <= 160 in every row block), a random
    graph of mean degree ~20."""
    from acm_gnn_amd import data as D
    if cfg.get("graph") != "dense":
        adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=5, pad_to=world)
        return adj, x_np, y_np, tr
    import scipy.sparse as sp
    n = 96 * world
    rng = np.random.default_rng(17)
    m = n * 10
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    adj = sp.csr_matrix((np.ones(m, np.float32), (r, c)), shape=(n, n))
    adj = ((adj + adj.T) > 0).astype(np.float32).tocsr()
    adj.setdiag(0)
    adj.eliminate_zeros()
    x_np = rng.standard_normal((n, 7)).astype(np.float32)
    y_np = rng.integers(0, 2, n)
    return adj, x_np, y_np, np.sort(rng.permutation(n)[: n // 2])


def _train_worker(rank, world, port, cfg, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import fake_lib

    class MP:
        def setattr(self, obj, name, val):
            setattr(obj, name, val)

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake_lib.install(MP())
        import acm_gnn_amd
        from acm_gnn_amd import GCN, data as D, distributed as DD, functional as AF, train as T
        if cfg.get("pipeline"):
            acm_gnn_amd.tuning.HOST.pipeline = 32                    # (a child process of its own: nothing to restore)
        adj, x_np, y_np, tr = _train_dataset(cfg, world)
        low, deg = D.build_filters(adj)
        n = adj.shape[0]
        plan = DD.equal_rows_plan(n, world) if cfg["plan"] == "rows" else DD.shard_plan(low.indptr, world, 8)
        ops = DD.make_sharded_operators(low, deg, "cpu", with_structure=bool(cfg["s"]), plan=plan)
        b, e = plan.rows(rank)
        if cfg.get("x_full"):
            ops.x_full = torch.from_numpy(x_np)
        hid = cfg.get("hid", 16)
        torch.manual_seed(0)
        full = GCN(7, hid, 2, 2, n, cfg["dropout"], cfg["model"], cfg["s"], variant=bool(cfg["variant"]), attn_layernorm=True)
        model = GCN(7, hid, 2, 2, e - b, cfg["dropout"], cfg["model"], cfg["s"], variant=bool(cfg["variant"]), attn_layernorm=True)
        sd = full.state_dict()
        for k in list(sd):
            if k.endswith(".struc_low"):
                sd[k] = sd[k][b:e].clone()
        model.load_state_dict(sd)
        model.dropout_state = AF.DropoutState("cpu", seed=7)
        opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.02, weight_decay=1e-3)
        w = T.row_weights(torch.from_numpy(DD.local_index(tr, plan, rank)), e - b, n_train_total=len(tr))
        carried = []
        if cfg.get("pipeline"):
            fake, bwd = acm_gnn_amd._lib.load(), acm_gnn_amd._lib.load().acm_conv_agg_bwd
            fake.acm_conv_agg_bwd = lambda nn, qq, *a: (carried.append((bool(qq._obj.next_agg), bool(qq._obj.proj_dz))), bwd(nn, qq, *a))[1]
        step = T.TrainStep(model, opt, torch.from_numpy(x_np[b:e]), ops, torch.from_numpy(y_np[b:e]), w, fused_dropout=True)
        losses = [float(step()) for _ in range(3)]
        assert step._defer                                  # the deferred path stayed on
        if cfg.get("pipeline"):                             # every backward carried the next step's gather (of THIS rank's
            assert step.pipe is not None and carried == [(True, True)] * 3, carried    # rows) and the output layer's projection
            assert step.pipe.table().shape[0] == n and step.pipe.agg().shape[0] == e - b
        params = {k: p.detach().numpy().copy() for k, p in model.named_parameters()}
        ret.put((rank, losses, params, (b, e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [dict(model="acmgcnp", s=0, variant=0, dropout=0.3, plan="rows"),
                                 dict(model="acmgcnp", s=1, variant=1, dropout=0.0, plan="work"),
                                 dict(model="acmgcnpp", s=0, variant=0, dropout=0.3, plan="rows"),
                                 dict(model="acmgcnp", s=0, variant=0, dropout=0.3, plan="rows", graph="dense", hid=64, x_full=1,
                                      pipeline=1),
                                 dict(model="acmgcnp", s=0, variant=0, dropout=0.3, plan="rows", graph="dense", hid=64, x_full=1,
                                      pipeline=1, world=4)],
                         ids=["agg-dropout", "struct-acmii-work-plan", "acmgcnpp", "input-pipeline", "4-ranks-input-pipeline"])
def test_sharded_train_step_equals_single_process(cfg, monkeypatch):
    """train.TrainStep on two gloo ranks: the second phases of every partial sum run as ONE deferred launch per step and
    the row-shard gradient sums are all-reduced right after it (DeferredReductions.allreduce), before the optimizer --
    three steps must leave every rank with the parameters of the single-process run."""
    import queue
    import time
    import torch.multiprocessing as mp
    import fake_lib
    world, port = cfg.get("world", 2), _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, cfg, ret)) for r in range(world)]
    for p in procs:
        p.start()
    results, deadline = [], time.time() + 300
    while len(results) < world:
        try:
            results.append(ret.get(timeout=2))
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() > deadline:
                for p in procs:
                    p.terminate()
                pytest.fail(f"sharded train workers failed (exit codes {[p.exitcode for p in procs]})")
    for p in procs:
        p.join(60)
    results.sort(key=lambda t: t[0])
    fake_lib.install(monkeypatch)
    import acm_gnn_amd
    from acm_gnn_amd import GCN, data as D, distributed as DD, functional as AF, train as T
    adj, x_np, y_np, tr = _train_dataset(cfg, world)
    low, deg = D.build_filters(adj)
    n = adj.shape[0]
    ops = DD.make_sharded_operators(low, deg, "cpu", with_structure=bool(cfg["s"]))
    torch.manual_seed(0)
    full = GCN(7, cfg.get("hid", 16), 2, 2, n, cfg["dropout"], cfg["model"], cfg["s"], variant=bool(cfg["variant"]), attn_layernorm=True)
    full.dropout_state = AF.DropoutState("cpu", seed=7)
    opt = acm_gnn_amd.FusedAdamW(full.parameters(), lr=0.02, weight_decay=1e-3)
    w = T.row_weights(torch.from_numpy(tr), n)
    step = T.TrainStep(full, opt, torch.from_numpy(x_np), ops, torch.from_numpy(y_np), w, fused_dropout=True, pipeline_input=False)
    ref_losses = [float(step()) for _ in range(3)]
    for rank, losses, params, (b, e) in results:
        for k, p in full.named_parameters():
            ref = p.detach()[b:e] if k.endswith(".struc_low") else p.detach()
            torch.testing.assert_close(torch.from_numpy(params[k]), ref, rtol=2e-4, atol=2e-6, msg=lambda m, k=k: f"{k}: {m}")
    # each rank reports the loss over its own rows: the ranks' losses add up to the single-process loss
    for i in range(3):
        assert abs(sum(r[1][i] for r in results) - ref_losses[i]) < 1e-5 * max(1.0, abs(ref_losses[i]))


def _rows_worker(rank, world, port, case, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import fake_lib

    class MP:
        def setattr(self, obj, name, val):
            setattr(obj, name, val)

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake_lib.install(MP())
        from acm_gnn_amd import data as D, distributed as DD
        adj, x_np, _, _, _ = D.synthetic_dataset("tiny", seed=5, pad_to=world)
        if case == "self-loops":                       # raw self-loops: the diagonal of I + A counts twice
            adj = adj.tolil()
            for i in (1, adj.shape[0] - 2):
                adj[i, i] = 1.0
            adj = adj.tocsr()
        low, deg = D.build_filters(adj)
        low = low.tocsr()
        low.sort_indices()
        if case == "asymmetric":                       # one edge without its mirror image, held by the LAST rank only
            low = low.tolil()
            i = low.shape[0] - 1
            j = next(c for c in range(low.shape[0]) if low[i, c] == 0 and c != i)
            low[i, j] = low[i, i]
            low = low.tocsr()
            low.sort_indices()
        n = low.shape[0]
        plan = DD.shard_plan(low.indptr, world, 8)
        b, e = plan.rows(rank)
        mine = low[b:e]                                # all this rank ever looks at
        form = DD.pattern_form_of_rows(mine.indptr, mine.indices, mine.data, b, n)
        if case == "asymmetric":
            ret.put((rank, form is None, None))
            return
        ops = DD.make_sharded_operators_from_rows(mine.indptr, mine.indices, mine.data, deg[b:e], plan, rank, "cpu",
                                                  with_structure=True)
        ref = DD.make_sharded_operators(low, deg, "cpu", with_structure=True, plan=plan)
        same = all(torch.equal(a, c) for a, c in zip(ops.low.arrays()[:2], ref.low.arrays()[:2]))
        same = same and torch.equal(ops.row_scale, ref.row_scale) and torch.equal(ops.deg, ref.deg) and ops.implicit
        # ... and the product: rows [b, e) of A_low @ X from the pattern, the row scale and the padded table
        from acm_gnn_amd import functional as AF
        table = torch.from_numpy(plan.pad_rows(x_np))
        got = AF.spmm(ops.low, table, row_scale=ops.row_scale).numpy()
        want = (low @ x_np)[b:e]
        ret.put((rank, bool(same), float(np.abs(got - want).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["plain", "self-loops", "asymmetric"])
def test_operators_built_from_each_ranks_own_rows(case):
    """distributed.make_sharded_operators_from_rows: each rank sees only its row block of A_low; the pattern-only form is
    decided row-locally plus a 16-byte-per-rank exchange of edge-hash sums for the symmetry of the pattern (an edge whose
    mirror image is missing on ANOTHER rank is noticed by every rank)."""
    import queue
    import time
    import torch.multiprocessing as mp
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, case, ret)) for r in range(world)]
    for p in procs:
        p.start()
    results, deadline = [], time.time() + 300
    while len(results) < world:
        try:
            results.append(ret.get(timeout=2))
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() > deadline:
                for p in procs:
                    p.terminate()
                pytest.fail(f"workers failed (exit codes {[p.exitcode for p in procs]})")
    for p in procs:
        p.join(60)
    for rank, flag, err in results:
        assert flag, (case, rank)
        if err is not None:
            assert err < 1e-5, (case, rank, err)
