"""Two ranks sharing the one MI355X of the test box (gloo moves the halos, the HIP kernels do the rest): the
row-sharded forward / backward with the real kernels must reproduce the single-process result.  (RCCL refuses two
ranks on one device; the collective layer is torch's either way -- this covers the combination "real kernels +
world_size 2" that neither the CPU gloo tests nor the single-rank GPU tests see.)"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _prepare(cfg, world):
    """(adj, x, y, train idx, low, deg, plan) of a case: the tiny graph, or -- BASELINE config 4's multi-GPU half as far
    as one device allows -- exactly bench.py's workload and plans (degree ranking dealt to the ranks like cards + equal
    blocks; random ids + acm_shard_plan)."""
    from acm_gnn_amd import data as D, distributed as DD
    if cfg.get("dataset", "tiny") == "tiny":
        adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=5, pad_to=world)
        if cfg.get("plan") == "work":                 # hubs first + the work-balanced plan: blocks of different lengths
            adj, x_np, y_np, (tr, _, _) = D.permute_dataset(adj, x_np, y_np, (tr, tr, tr), D.degree_order(adj))
        low, deg = D.build_filters(adj)
        n = adj.shape[0]
        plan = DD.shard_plan(low.indptr, world, 8) if cfg.get("plan") == "work" else DD.equal_rows_plan(n, world)
        return adj, x_np, y_np, tr, low, deg, plan
    order = "degree" if cfg["plan"] == "interleave" else "random"
    wl = D.bench_workload(cfg["dataset"], seed=0, node_order=order, pad_to=world)
    adj, x_np, y_np, (tr, va, te), low, deg = (wl[k] for k in ("adj", "x", "y", "splits", "low", "deg"))
    n = adj.shape[0]
    if cfg["plan"] == "interleave":
        adj, x_np, y_np, (tr, va, te) = D.permute_dataset(adj, x_np, y_np, (tr, va, te), DD.interleave_order(n, world))
        low, deg = D.build_filters(adj)
        plan = DD.equal_rows_plan(n, world)
    else:
        plan = DD.shard_plan(low.indptr, world)
    return adj, x_np, y_np, tr, low, deg, plan


def _build(cfg, n_local, n, dev, f_in=7, n_cls=2):
    from acm_gnn_amd import GCN, functional as AF
    torch.manual_seed(0)
    full = GCN(f_in, 64, n_cls, 2, n, cfg["dropout"], cfg["model"], cfg["s"], variant=bool(cfg["variant"]), attn_layernorm=True)
    model = GCN(f_in, 64, n_cls, 2, n_local, cfg["dropout"], cfg["model"], cfg["s"], variant=bool(cfg["variant"]),
                attn_layernorm=True)
    return full, model


def _worker(rank, world, port, cfg, ret):
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from acm_gnn_amd import data as D, distributed as DD, functional as AF
        adj, x_np, y_np, tr, low, deg, plan = _prepare(cfg, world)
        n = adj.shape[0]
        ops = DD.make_sharded_operators(low, deg, DEV, with_structure=bool(cfg["s"]), plan=plan)
        assert ops.sharded and ops.uniform == (cfg.get("plan") != "work")
        ops.hops = cfg.get("hops", 1)
        b, e = plan.rows(rank)
        full, model = _build(cfg, e - b, n, DEV, x_np.shape[1], int(y_np.max()) + 1)
        sd = full.state_dict()
        for k in list(sd):
            if k.endswith(".struc_low"):
                sd[k] = sd[k][b:e].clone()
        model.load_state_dict(sd)
        model = model.to(DEV)
        if cfg["dropout"]:
            model.fused_dropout, model.dropout_state = True, AF.DropoutState(DEV, seed=7)
            ops.x_full = torch.from_numpy(x_np).to(DEV)
        x = torch.from_numpy(x_np[b:e]).to(DEV)
        y = torch.from_numpy(y_np[b:e]).to(DEV)
        idx = torch.from_numpy(DD.local_index(tr, plan, rank)).to(DEV)
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        out = model(x, ops)
        loss = F.nll_loss(F.log_softmax(out, 1)[idx], y[idx], reduction="sum") / len(tr)
        loss.backward()
        torch.cuda.synchronize()
        AF.set_kernel_timer(None)
        if cfg["variant"] and x_np.shape[1] <= 8:
            # the ACMII first layer runs in the mask form on every rank: the table over the all-gathered input, and NO
            # all-gather of the 128-float gradient rows in its backward (the fp32 form's transposed products need one)
            used = set(k.split("/")[0] for k in timer.events)
            assert {"acmii_table", "conv_acmii_v_fwd", "conv_acmii_v_bwd"} <= used, sorted(used)
            assert not any(k.startswith("all_gather/") and k.endswith("x128") for k in timer.events), sorted(timer.events)
        if cfg.get("wide"):
            used = sorted(timer.events)
            assert any(k.startswith("conv_aggw/") for k in used), used
            assert [k for k in used if k.startswith("all_gather/") and k.endswith("x128")] == [f"all_gather/{plan.n_max}x128"], used
            assert sum(len(v) for k, v in timer.events.items() if k.startswith("all_gather/") and k.endswith("x128")) == 1
        grads = {k: p.grad.cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
        ret.put((rank, out.detach().cpu().numpy().copy(), grads, (b, e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [dict(model="acmgcnp", s=0, variant=0, dropout=0.0),
                                 dict(model="acmgcnp", s=1, variant=0, dropout=0.3),
                                 dict(model="acmgcn", s=0, variant=1, dropout=0.0),
                                 dict(model="acmgcnp", s=1, variant=1, dropout=0.3, plan="work"),
                                 dict(model="acmgcnp", s=0, variant=0, dropout=0.3, plan="work"),
                                 dict(model="acmgcnpp", s=0, variant=0, dropout=0.3),
                                 dict(model="acmgcnp", s=0, variant=0, dropout=0.1, dataset="twitch-gamer", plan="interleave"),
                                 dict(model="acmgcnp", s=0, variant=0, dropout=0.1, dataset="twitch-gamer", plan="work"),
                                 dict(model="acmsgc", s=0, variant=0, dropout=0.0, dataset="arxiv-year", plan="work", hops=3),
                                 # the wide aggregate-first first layer row-sharded: ONE 128-wide halo exchange (the dropped input)
                                 dict(model="acmgcnp", s=0, variant=0, dropout=0.2, dataset="arxiv-year", plan="interleave", wide=1),
                                 # EIGHT ranks on the one device: eight HIP contexts, the halos through gloo
                                 dict(model="acmgcnp", s=1, variant=0, dropout=0.3, world=8),
                                 dict(model="acmgcnp", s=0, variant=1, dropout=0.3, plan="work", world=8)],
                         ids=["agg", "struct-dropout", "acmii", "work-plan-struct-acmii", "work-plan-agg", "acmgcnpp",
                              "twitch-degree-interleaved", "twitch-random-work-plan", "arxiv-year-3hop-sgc-work-plan",
                              "arxiv-year-wide-aggregate-first",
                              "world8-struct-dropout", "world8-work-plan-acmii"])
def test_two_ranks_on_one_gpu_equal_single_process(cfg):
    import queue
    import time
    import torch.multiprocessing as mp
    import torch.nn.functional as F
    world, port = int(cfg.get("world", 2)), _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, ret)) for r in range(world)]
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
                pytest.fail(f"sharded GPU workers failed (exit codes {[p.exitcode for p in procs]})")
    for p in procs:
        p.join(60)
    results.sort(key=lambda t: t[0])
    from acm_gnn_amd import data as D, distributed as DD, functional as AF
    adj, x_np, y_np, tr, low, deg, plan = _prepare(cfg, world)
    n = adj.shape[0]
    big = cfg.get("dataset", "tiny") != "tiny"
    if big:                                            # what the JSON line's config.shard reports
        _, nnz_r, work_r = plan.work(low.indptr, DD.DEFAULT_ROW_COST)
        assert work_r.max() / work_r.mean() < 1.05 and nnz_r.max() / nnz_r.mean() < (1.05 if cfg["plan"] == "interleave" else 1.6)
    ops = DD.make_sharded_operators(low, deg, DEV, with_structure=bool(cfg["s"]))
    ops.hops = cfg.get("hops", 1)
    full, _ = _build(cfg, n, n, DEV, x_np.shape[1], int(y_np.max()) + 1)
    full = full.to(DEV)
    if cfg["dropout"]:
        full.fused_dropout, full.dropout_state = True, AF.DropoutState(DEV, seed=7)
    out = full(torch.from_numpy(x_np).to(DEV), ops)
    idx = torch.from_numpy(tr).to(DEV)
    loss = F.nll_loss(F.log_softmax(out, 1)[idx], torch.from_numpy(y_np).to(DEV)[idx], reduction="sum") / len(tr)
    loss.backward()
    got = np.concatenate([r[1] for r in results])
    ref_out = out.detach().cpu().numpy()
    if big:       # logits span 1e4 (row-normalised N(0,1) features): bounds relative to the range, as in test_gpu_fullsize
        assert float(np.abs(got - ref_out).max()) < 2e-6 * float(np.abs(ref_out).max())
    else:
        np.testing.assert_allclose(got, ref_out, rtol=1e-4, atol=1e-5)
    for k, p in full.named_parameters():
        if p.grad is None:
            continue
        for rank, _, grads, (b, e) in results:
            ref = p.grad[b:e] if k.endswith(".struc_low") else p.grad
            if big:
                assert float(np.abs(grads[k] - ref.cpu().numpy()).max()) < 3e-4 * float(ref.abs().max()) + 1e-6, k
            else:
                np.testing.assert_allclose(grads[k], ref.cpu().numpy(), rtol=1e-3,
                                           atol=1e-5 * max(1.0, float(ref.abs().max())), err_msg=k)


def _train_worker(rank, world, port, cfg, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import acm_gnn_amd
        from acm_gnn_amd import distributed as DD, functional as AF, train as T
        adj, x_np, y_np, tr, low, deg, plan = _prepare(cfg, world)
        n = adj.shape[0]
        ops = DD.make_sharded_operators(low, deg, DEV, plan=plan)
        b, e = plan.rows(rank)
        full, model = _build(cfg, e - b, n, DEV, x_np.shape[1], int(y_np.max()) + 1)
        sd = full.state_dict()
        for k in list(sd):
            if k.endswith(".struc_low"):
                sd[k] = sd[k][b:e].clone()
        model.load_state_dict(sd)
        model = model.to(DEV)
        model.dropout_state = AF.DropoutState(DEV, seed=7)
        ops.x_full = torch.from_numpy(x_np).to(DEV)
        opt = acm_gnn_amd.FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
        w = T.row_weights(torch.from_numpy(DD.local_index(tr, plan, rank)).to(DEV), e - b, n_train_total=len(tr))
        step = T.TrainStep(model, opt, torch.from_numpy(x_np[b:e]).to(DEV), ops, torch.from_numpy(y_np[b:e]).to(DEV), w,
                           fused_dropout=True)
        timer = AF.KernelTimer()
        AF.set_kernel_timer(timer)
        losses = [float(step()) for _ in range(3)]
        AF.set_kernel_timer(None)
        torch.cuda.synchronize()
        labels = sorted(timer.summary())
        pipe = step.pipe
        ret.put((rank, losses, {k: p.detach().cpu().numpy().copy() for k, p in model.named_parameters()}, labels,
                 None if pipe is None else (tuple(pipe.table().shape), tuple(pipe.agg().shape), bool(pipe.primed))))
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_the_input_pipeline_equal_the_plain_single_process_step():
    """bench.py's N > 1 configuration (degree ranking dealt like cards, equal blocks, replicated input) with
    train.TrainStep's input pipeline ON in both ranks: each rank's backward kernel carries the gather of ITS rows of the next
    step's P = A_low dropout(x) out of a locally drawn table of ALL nodes (no exchange), and also the output layer's
    projection backward.  Three optimizer steps against the single-process step with the pipeline off."""
    import queue
    import time
    import torch.multiprocessing as mp
    cfg = dict(model="acmgcnp", s=0, variant=0, dropout=0.1, dataset="twitch-gamer", plan="interleave")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, cfg, ret)) for r in range(world)]
    for p in procs:
        p.start()
    results, deadline = [], time.time() + 400
    while len(results) < world:
        try:
            results.append(ret.get(timeout=2))
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() > deadline:
                for p in procs:
                    p.terminate()
                pytest.fail(f"sharded GPU train workers failed (exit codes {[p.exitcode for p in procs]})")
    for p in procs:
        p.join(60)
    results.sort(key=lambda t: t[0])
    import acm_gnn_amd
    from acm_gnn_amd import distributed as DD, functional as AF, train as T
    adj, x_np, y_np, tr, low, deg, plan = _prepare(cfg, world)
    n = adj.shape[0]
    for rank, _, _, labels, pipe in results:
        b, e = plan.rows(rank)
        assert pipe == ((n, 8), (e - b, 8), True), pipe
        assert any(s.startswith("conv_agg_bwd+gather+proj") for s in labels), labels      # the combined kernel ran
        assert not any(s.startswith("proj_bwd") for s in labels), labels                  # no separate projection backward
        assert sorted(s for s in labels if s.startswith("all_gather")) == [f"all_gather/{plan.n_max}x4"], labels   # the output
        # layer's [Z_L | Z_H] and [G_L | G_H] halos only: nothing is exchanged for the first layer or its carried gather
    ops = DD.make_sharded_operators(low, deg, DEV)
    full, _ = _build(cfg, n, n, DEV, x_np.shape[1], int(y_np.max()) + 1)
    full = full.to(DEV)
    full.dropout_state = AF.DropoutState(DEV, seed=7)
    opt = acm_gnn_amd.FusedAdamW(full.parameters(), lr=0.01, weight_decay=1e-3)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), n)
    step = T.TrainStep(full, opt, torch.from_numpy(x_np).to(DEV), ops, torch.from_numpy(y_np).to(DEV), w, fused_dropout=True,
                       pipeline_input=False)
    ref_losses = [float(step()) for _ in range(3)]
    for i in range(3):
        assert abs(sum(r[1][i] for r in results) - ref_losses[i]) < 2e-5 * max(1.0, abs(ref_losses[i])), (i, ref_losses)
    for rank, _, params, _, _ in results:
        for k, p in full.named_parameters():
            if k.endswith(".struc_low"):
                continue                                   # unused without the structure channel
            ref = p.detach().cpu().numpy()
            assert float(np.abs(params[k] - ref).max()) < 2e-4 * float(np.abs(ref).max()) + 1e-6, k
