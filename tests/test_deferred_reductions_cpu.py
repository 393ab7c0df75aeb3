"""Deferred second phases (acm_reduce_list_t, include/acm_hip.h) through the numpy double of the C ABI, which poisons
the outputs of a deferred call with NaN until acm_reduce_flush: the host plumbing (functional.deferred_reductions,
train.TrainStep) must flush before anything reads a gradient, give the results of the immediate form, and fall back
to the immediate form when autograd did not adopt the kernels' output tensors."""
import numpy as np
import pytest
import torch

import fake_lib


def _setup(model_type="acmgcnp", s=0, hidden=16):
    from acm_gnn_amd import GCN, data as D, train as T
    from acm_gnn_amd.graph import CsrGraph, FilterOperators
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=3)
    low, _ = D.build_filters(adj)
    ops = FilterOperators(CsrGraph.from_scipy(low, "cpu"))
    x, y = torch.from_numpy(D.row_normalize_features(x_np)), torch.from_numpy(y_np)
    w = T.row_weights(torch.from_numpy(tr), x.shape[0])
    torch.manual_seed(0)
    model = GCN(x.shape[1], hidden, int(y.max()) + 1, 2, x.shape[0], 0.0, model_type, s)
    return model, ops, x, y, w


def _grads(model):
    return {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}


def test_outputs_are_undefined_until_the_flush_and_equal_the_immediate_form(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import functional as AF
    model, ops, x, y, w = _setup()
    out = model(x, ops)
    loss0, dz = AF.nll_loss_and_grad(out, y, w)
    out.backward(dz)
    want = _grads(model)
    model.zero_grad(set_to_none=True)
    with AF.deferred_reductions() as pending:
        out = model(x, ops)
        loss, dz = AF.nll_loss_and_grad(out, y, w)
        assert torch.isfinite(dz).all() and torch.isnan(loss)        # dz is first-phase output, the loss is not
        out.backward(dz)
        assert pending.pending >= 4                                  # loss, K3, acm_proj_bwd, acm_conv_agg_bwd
        poisoned = [k for k, v in model.named_parameters() if v.grad is not None and torch.isnan(v.grad).any()]
        assert len(poisoned) >= len(want) - 1                        # everything the reduced kernels produce
        assert pending.all_adopted([loss] + [p.grad for p in model.parameters()])
        pending.flush()
        assert pending.pending == 0
    assert float(loss) == float(loss0)
    got = _grads(model)
    assert got.keys() == want.keys()
    # (under the deferral list the output layer's projection backward rides the hidden layer's kernel -- acm_conv_agg_bwd_t.
    #  proj_*, round 4: only there -- so the two runs differ by the rounding of one more fused product)
    for k in want:
        torch.testing.assert_close(got[k], want[k], rtol=1e-5, atol=1e-6 * float(want[k].abs().max()), msg=lambda m, k=k: f"{k}: {m}")


def test_leaving_the_block_flushes_and_an_exception_discards(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import functional as AF
    model, ops, x, y, w = _setup()
    with AF.deferred_reductions():
        loss, _ = AF.nll_loss_and_grad(model(x, ops), y, w)
    assert torch.isfinite(loss)
    try:
        with AF.deferred_reductions() as pending:
            AF.nll_loss_and_grad(model(x, ops), y, w)
            raise KeyError("boom")
    except KeyError:
        pass
    assert pending.pending == 0 and AF._ambient().defer is None


def test_train_step_defers_and_matches_the_immediate_trajectory(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import FusedAdam, train as T
    traj = {}
    for defer in (True, False):
        model, ops, x, y, w = _setup()
        opt = FusedAdam(model.parameters(), lr=0.01, weight_decay=5e-4)
        step = T.TrainStep(model, opt, x, ops, y, w)
        step._defer = defer
        traj[defer] = [float(step()) for _ in range(4)]
        assert step._defer == defer
        assert all(torch.isfinite(p).all() for p in model.parameters())
    assert traj[True] == traj[False]


def test_train_step_leaves_the_flush_to_the_optimizer_launch(monkeypatch):
    """With this package's optimizer the step's deferral list travels to acm_adam_step (acm_adam_config_t.pending, ABI 23)
    unflushed: no acm_reduce_flush call of the host's, one flush inside every update -- and the same trajectory as with
    flush_in_optimizer=False.  A foreign optimizer, two parameter groups or a pending gradient all-reduce keep the host's
    own flush (nothing may read a gradient before it)."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import FusedAdam, FusedAdamW, functional as AF, train as T
    flushes = {"host": 0}
    real = AF.DeferredReductions.flush
    monkeypatch.setattr(AF.DeferredReductions, "flush", lambda self: (flushes.__setitem__("host", flushes["host"] + 1), real(self))[1])
    traj = {}
    for inside in (True, False):
        model, ops, x, y, w = _setup()
        opt = FusedAdamW(model.parameters(), lr=0.01, weight_decay=5e-4)
        step = T.TrainStep(model, opt, x, ops, y, w, flush_in_optimizer=inside)
        flushes["host"], fake.adam_flushed = 0, 0
        traj[inside] = [float(step()) for _ in range(4)]
        assert (flushes["host"], fake.adam_flushed) == ((0, 4) if inside else (4, 0))
        assert all(torch.isfinite(p).all() for p in model.parameters())
    assert traj[True] == traj[False]
    # two parameter groups: the optimizer flushes on the host before its first launch
    model, ops, x, y, w = _setup()
    ps = list(model.parameters())
    opt = FusedAdam([{"params": ps[:3]}, {"params": ps[3:], "lr": 0.02}], lr=0.01)
    step = T.TrainStep(model, opt, x, ops, y, w)
    flushes["host"], fake.adam_flushed = 0, 0
    assert np.isfinite(float(step())) and (flushes["host"], fake.adam_flushed) == (1, 0)
    # a foreign optimizer: the step flushes itself, as before
    model, ops, x, y, w = _setup()
    step = T.TrainStep(model, torch.optim.SGD(model.parameters(), lr=0.1), x, ops, y, w)
    flushes["host"] = 0
    assert np.isfinite(float(step())) and flushes["host"] == 1 and step._unflushed is None


def test_train_step_falls_back_when_a_gradient_was_accumulated_instead_of_adopted(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import train as T

    class KeepsGrads(torch.optim.SGD):                    # an optimizer that does not clear .grad: autograd then
        def zero_grad(self, set_to_none=True):            # ACCUMULATES into it, i.e. reads the new gradient at once
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is not None:
                        p.grad.zero_()

    model, ops, x, y, w = _setup()
    ref, *_ = _setup()
    ref.load_state_dict(model.state_dict())
    opt = KeepsGrads(model.parameters(), lr=0.1)
    step = T.TrainStep(model, opt, x, ops, y, w)
    first = float(step())                                 # .grad is None here: adopted
    assert step._defer
    second = float(step())                                # .grad exists: detected, redone without deferral
    assert not step._defer
    ref_step = T.TrainStep(ref, torch.optim.SGD(ref.parameters(), lr=0.1), x, ops, y, w)
    ref_step._defer = False
    assert [first, second] == [float(ref_step()), float(ref_step())]
    for a, b in zip(model.parameters(), ref.parameters()):
        assert torch.isfinite(a).all() and torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_fused_loss_tail_is_taken_by_the_output_layer_and_equals_the_three_calls(monkeypatch):
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import functional as AF
    calls = []
    real = fake.acm_conv_fwd_tail
    monkeypatch.setattr(fake, "acm_conv_fwd_tail", lambda *a: (calls.append(1), real(*a))[1])
    model, ops, x, y, w = _setup()
    model(x, ops)                                   # (no input dropout: the first pass leaves P = A_low X for every later one)
    out = model(x, ops)
    loss0, dz0 = AF.nll_loss_and_grad(out, y, w)
    out.backward(dz0)
    want = _grads(model)
    assert not calls
    model.zero_grad(set_to_none=True)
    with AF.fused_loss_tail(y, w) as tail:
        out = model(x, ops)
    assert calls == [1] and tail.matches(out)                    # only the output layer took the request
    assert torch.equal(tail.dz, dz0) and float(tail.loss) == float(loss0)
    out.backward(tail.dz)
    got = _grads(model)
    assert got.keys() == want.keys() and all(torch.equal(got[k], want[k]) for k in want)
    # any other gradient makes the layer run its row-local backward again
    model.zero_grad(set_to_none=True)
    with AF.fused_loss_tail(y, w) as tail:
        out = model(x, ops)
    out.backward(2.0 * tail.dz)
    for k, v in _grads(model).items():
        assert torch.allclose(v, 2.0 * want[k], rtol=1e-5, atol=1e-7), k
    # no request, or a layer that does not qualify: nothing changes
    with AF.fused_loss_tail(y, w) as tail:
        hidden = model.gcns[0](x, ops)
    assert tail.out is None and not tail.matches(hidden)


def test_train_step_with_the_wide_aggregate_first_layer_defers_its_weight_gradients_too(monkeypatch, tune):
    """functional._AcmAggWide under train.TrainStep: acm_conv_aggw_bwd leaves the head-parameter sums AND the three weight
    gradients (views of the layer's flat buffer) to the step's one flush -- the step stays deferred (every gradient adopted), and
    the trajectory equals the immediate one and the unfused form's (K3 + two transposed products)."""
    fake = fake_lib.install(monkeypatch)
    import scipy.sparse as sp
    from acm_gnn_amd import GCN, FusedAdamW, data as D, train as T
    from acm_gnn_amd.distributed import make_sharded_operators
    rng = np.random.default_rng(3)
    n = 8192
    m = n * 12
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    adj = sp.csr_matrix((np.ones(m, np.float32), (r, c)), shape=(n, n))
    adj = ((adj + adj.T) > 0).astype(np.float32).tocsr()
    adj.setdiag(0)
    adj.eliminate_zeros()
    low, deg = D.build_filters(adj)
    ops = make_sharded_operators(low, deg, torch.device("cpu"))
    x = torch.randn(n, 40, generator=torch.Generator().manual_seed(1))
    y = torch.randint(0, 3, (n,), generator=torch.Generator().manual_seed(2))
    w = T.row_weights(torch.arange(0, n, 2), n)
    calls = []
    orig = fake.acm_conv_aggw_bwd
    monkeypatch.setattr(fake, "acm_conv_aggw_bwd", lambda *a: (calls.append(1), orig(*a))[1])
    traj = {}
    for key, fused, defer in (("fused+deferred", 1, True), ("fused", 1, False), ("unfused+deferred", 0, True)):
        tune(aggw_fused=fused)
        torch.manual_seed(0)
        model = GCN(40, 64, 3, 2, n, 0.3, "acmgcnp", 0, variant=False, attn_layernorm=True)
        step = T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3), x, ops, y, w, small_step=False)
        step._defer = defer
        calls.clear()
        traj[key] = [float(step()) for _ in range(3)]
        assert step._defer == defer and len(calls) == (3 if fused else 0)
        assert all(torch.isfinite(p).all() for p in model.parameters())
    assert traj["fused+deferred"] == traj["fused"]
    np.testing.assert_allclose(traj["fused+deferred"], traj["unfused+deferred"], rtol=1e-5)


@pytest.mark.parametrize("cfg", [dict(model_type="acmgcnp", s=0, variant=0, hidden=64, dropout=0.3),
                                 dict(model_type="acmgcnp", s=0, variant=0, hidden=16, dropout=0.0),
                                 dict(model_type="acmgcn", s=0, variant=1, hidden=64, dropout=0.3),
                                 dict(model_type="acmgcnp", s=1, variant=0, hidden=64, dropout=0.3),
                                 dict(model_type="acmgcnp", s=1, variant=1, hidden=16, dropout=0.0),
                                 dict(model_type="acmgcnpp", s=0, variant=0, hidden=64, dropout=0.3),
                                 dict(model_type="acmgcnp", s=0, variant=0, hidden=64, dropout=0.3, csr=1),
                                 dict(model_type="acmgcnp", s=0, variant=0, hidden=64, dropout=0.3, torch_dropout=1)],
                         ids=["agg-first", "literal", "acmii", "structure", "structure-acmii", "residual", "csr-features", "mask-tensors"])
def test_train_step_on_its_own_tape_equals_the_autograd_step(cfg, monkeypatch):
    """train.TrainStep(tape=True): the step records this package's Functions itself (functional.Tape) and replays them backwards
    -- no autograd graph.  Same losses, same parameters and the same library calls as the autograd step (the fused loss tail,
    the lazy projection backward and the input pipeline included: the tape stands in for ``grad_fn`` and ``needs_input_grad``)."""
    fake = fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, FusedAdamW, data as D, functional as AF, train as T
    from acm_gnn_amd.graph import CsrGraph, FilterOperators, SparseFeatures
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=3)
    low, deg = D.build_filters(adj)
    from acm_gnn_amd.distributed import make_sharded_operators
    ops = make_sharded_operators(low, deg, torch.device("cpu"), with_structure=bool(cfg["s"]))
    x = torch.from_numpy(D.row_normalize_features(x_np))
    if cfg.get("csr"):
        import scipy.sparse as sp
        x = SparseFeatures.from_scipy(sp.csr_matrix(x.numpy()), "cpu")
    y = torch.from_numpy(y_np)
    w = T.row_weights(torch.from_numpy(tr), y.shape[0])
    runs, calls = {}, []
    for name in [k for k in dir(fake) if k.startswith("acm_") and not k.endswith("_bytes") and k not in ("acm_last_error",)]:
        orig = getattr(fake, name)
        if callable(orig):
            monkeypatch.setattr(fake, name, (lambda o, nm: lambda *a: (calls.append(nm), o(*a))[1])(orig, name))
    for tape in (False, True):
        calls.clear()
        torch.manual_seed(0)
        model = GCN(7, cfg["hidden"], int(y.max()) + 1, 2, y.shape[0], cfg["dropout"], cfg["model_type"], cfg["s"],
                    variant=bool(cfg["variant"]), attn_layernorm=True)
        step = T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.01, weight_decay=1e-3), x, ops, y, w, small_step=False,
                           tape=tape, fused_dropout=not cfg.get("torch_dropout"))
        torch.manual_seed(1)                                   # (F.dropout mask tensors of the mask-tensors case)
        losses = [float(step()) for _ in range(2)]
        calls.clear()                                          # (the first steps carry one-off work: item streams, a redone step)
        losses.append(float(step()))
        runs[tape] = (losses, {k: v.detach().clone() for k, v in model.named_parameters()}, list(calls), step._tape)
    assert runs[True][3] is True, "the step fell back to autograd"
    assert runs[True][0] == runs[False][0]
    for k in runs[False][1]:
        assert torch.equal(runs[True][1][k], runs[False][1][k]), k
    strip = lambda cs: [c for c in cs if c not in ("acm_tuning_get", "acm_csr_info")]
    assert strip(runs[True][2]) == strip(runs[False][2])


def test_a_broken_tape_redoes_the_step_on_the_masks_it_drew(monkeypatch):
    """ACM-GCN++ with F.dropout masks adds its residual branch with a torch operation (models.py:55-56), so the
    first taped step aborts (TapeBroken) AFTER the forward has drawn F.dropout masks and is redone on autograd.  The redo must
    draw the SAME masks -- the generators' states are put back -- so that tape=True (the default) trains exactly like
    tape=False on the same seeds; a mask-replay harness in place of F.dropout (whose state cannot be put back) makes such a
    model start on autograd instead."""
    fake_lib.install(monkeypatch)
    import torch.nn.functional as F
    from acm_gnn_amd import GCN, FusedAdamW, data as D, train as T
    from acm_gnn_amd.distributed import make_sharded_operators
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=3)
    low, deg = D.build_filters(adj)
    ops = make_sharded_operators(low, deg, torch.device("cpu"))
    x, y = torch.from_numpy(D.row_normalize_features(x_np)), torch.from_numpy(y_np)
    w = T.row_weights(torch.from_numpy(tr), y.shape[0])

    def run(tape, mt="acmgcnpp"):
        torch.manual_seed(0)
        model = GCN(7, 64, int(y.max()) + 1, 2, y.shape[0], 0.4, mt, 0, attn_layernorm=True)
        step = T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.01), x, ops, y, w, small_step=False, tape=tape, fused_dropout=False)
        torch.manual_seed(5)
        return [float(step()) for _ in range(3)], step

    la, sa = run(False)
    lb, sb = run(True)
    assert sb._tape is False, "the residual add is a torch operation: the tape must have broken"
    assert la == lb
    # a patched F.dropout: acmgcnpp never starts on the tape, the two-layer models keep it
    calls = []
    real = F.dropout
    monkeypatch.setattr(F, "dropout", lambda inp, p=0.5, training=True, inplace=False: (calls.append(1), real(inp, p, training))[1])
    _, s1 = run(True)
    assert s1._tape is False and len(calls) == 3 * 3            # three sites per forward, three forwards: none drawn twice
    calls.clear()
    _, s2 = run(True, "acmgcnp")
    assert s2._tape is True and s2._tape_proven


def test_a_taped_step_releases_what_it_saved(monkeypatch):
    """the records of a Tape hold the step's saved activations; nothing that outlives the step -- the layers'
    ``att`` tensors, the returned loss -- may keep them alive, and no reference cycle may be left for the cyclic collector:
    with the collector switched off, the bytes of live tensors stay constant from step to step."""
    import gc
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, FusedAdamW, data as D, functional as AF, train as T
    from acm_gnn_amd.distributed import make_sharded_operators
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=3)
    low, deg = D.build_filters(adj)
    ops = make_sharded_operators(low, deg, torch.device("cpu"))
    x, y = torch.from_numpy(D.row_normalize_features(x_np)), torch.from_numpy(y_np)
    w = T.row_weights(torch.from_numpy(tr), y.shape[0])
    torch.manual_seed(0)
    model = GCN(7, 64, int(y.max()) + 1, 2, y.shape[0], 0.3, "acmgcnp", 0, attn_layernorm=True)
    model.dropout_state = AF.DropoutState("cpu", seed=3)
    step = T.TrainStep(model, FusedAdamW(model.parameters(), lr=0.01), x, ops, y, w, small_step=False, tape=True, pipeline_input=False)

    def live_bytes():
        seen, total = set(), 0
        for o in gc.get_objects():
            if isinstance(o, torch.Tensor) and o.device.type == "cpu":
                st = o.untyped_storage()
                if st.data_ptr() not in seen:
                    seen.add(st.data_ptr())
                    total += st.nbytes()
        return total

    for _ in range(3):
        step()
    gc.collect()
    gc.disable()
    try:
        before = live_bytes()
        for _ in range(6):
            step()
        after = live_bytes()
    finally:
        gc.enable()
    assert step._tape is True
    assert after <= before + 4096, (before, after)
    assert not model.gcns[0].att_low.requires_grad and not hasattr(model.gcns[0].att_low, "_acm_tape_ctx")
