"""acm_adam_step on the MI355X against torch.optim.Adam / AdamW (single-tensor CPU implementation):
vector and scalar paths, unaligned views, more than one pack of 32 tensors, hipGraph replay."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _shapes():
    return [(7, 64), (64, 1), (3, 3), (64,), (1, 1), (2500,), (5000, 3), (168114, 8), (4099,)] + [(5, i + 1) for i in range(36)]


@pytest.mark.parametrize("decoupled,wd", [(False, 0.0), (False, 5e-4), (True, 1e-2)])
def test_fused_adam_matches_torch(decoupled, wd):
    from acm_gnn_amd import FusedAdam, FusedAdamW
    g = torch.Generator().manual_seed(0)
    init = [torch.randn(*s, generator=g) for s in _shapes()]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    # one parameter is a 4-byte-offset view into a larger buffer: the scalar path
    mine = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    kw = dict(lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    a = (FusedAdamW if decoupled else FusedAdam)(mine, **kw)
    b = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ref, foreach=False, **kw)
    assert len(mine) > 40            # more than one pack of acm_adam_step (PACK = 40)
    for it in range(12):
        for p, q in zip(mine, ref):
            gr = torch.randn(q.shape, generator=g) * (1.0 + it)
            q.grad = gr
            buf = torch.empty(gr.numel() + 1, device=DEV)
            view = buf[1:].view(gr.shape)                       # 4-byte aligned, not 16
            view.copy_(gr)
            p.grad = view if it % 2 else gr.to(DEV)
        a.step()
        b.step()
    for p, q in zip(mine, ref):
        want = q.detach().numpy()
        np.testing.assert_allclose(p.detach().cpu().numpy(), want, rtol=3e-6, atol=3e-6 * max(1.0, float(np.abs(want).max())))
        st = a.state[p]
        assert float(st["step"]) == 12.0
        wm = b.state[q]["exp_avg"].numpy()
        np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), wm, rtol=3e-6, atol=3e-6 * float(np.abs(wm).max()))
        wv = b.state[q]["exp_avg_sq"].numpy()
        np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), wv, rtol=3e-6, atol=3e-6 * float(np.abs(wv).max()))


def test_fused_adam_under_graph_capture():
    from acm_gnn_amd import FusedAdamW
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(300, 7, generator=g), torch.randn(64, generator=g)]
    grads = [torch.randn(300, 7, generator=g), torch.randn(64, generator=g)]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    mine = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    static_g = [t.to(DEV) for t in grads]
    a = FusedAdamW(mine, lr=0.02, weight_decay=1e-3)
    b = torch.optim.AdamW(ref, lr=0.02, weight_decay=1e-3, foreach=False)
    for p, gr in zip(mine, static_g):
        p.grad = gr
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a.step()                                                 # creates the state outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        a.step()
    for _ in range(6):
        graph.replay()
    torch.cuda.synchronize()
    for q, gr in zip(ref, grads):
        q.grad = gr
    for _ in range(7):
        b.step()
    for p, q in zip(mine, ref):
        assert float(a.state[p]["step"]) == 7.0
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), rtol=3e-6, atol=3e-6)


def test_adam_entry_point_rejects_bad_arguments():
    import ctypes as C
    from acm_gnn_amd import _lib
    lib = _lib.load()
    cfg = _lib.AdamConfig(0.01, 0.9, 0.999, 1e-8, 0.0, 1, None, None)
    assert lib.acm_adam_step(1, None, C.byref(cfg), None) == 1
    ent = (_lib.AdamTensor * 1)()
    assert lib.acm_adam_step(1, C.cast(ent, C.c_void_p), C.byref(cfg), None) == 1 and b"NULL" in lib.acm_last_error()
    bad = _lib.AdamConfig(0.01, 1.5, 0.999, 1e-8, 0.0, 1, None, None)
    assert lib.acm_adam_step(0, None, C.byref(bad), None) == 1
    assert lib.acm_adam_step(0, None, C.byref(cfg), None) == 0


def test_load_torch_adam_checkpoint_and_step_on_device():
    """A torch.optim.Adam state_dict (CPU `step` tensors) and one mapped to the CPU load into FusedAdam and step on the
    GPU with the same result as torch continuing from the same state."""
    from acm_gnn_amd import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(7, 64), (64, 1), (3, 3), (5000, 3)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    grads = [torch.randn(*s, generator=g) for s in shapes]
    ref_p = [torch.nn.Parameter(t.clone().to(DEV)) for t in base]
    ref = torch.optim.Adam(ref_p, lr=0.01, weight_decay=5e-4, foreach=False)
    for p, gr in zip(ref_p, grads):
        p.grad = gr.to(DEV)
    ref.step()
    sd = ref.state_dict()
    assert sd["state"][0]["step"].device.type == "cpu"               # what the advisor's finding is about
    sd_cpu = {"state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
              "param_groups": sd["param_groups"]}
    import copy
    for which in (sd, sd_cpu):
        mine_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
        mine = FusedAdam(mine_p, lr=0.01, weight_decay=5e-4)
        mine.load_state_dict(copy.deepcopy(which))      # load_state_dict adopts tensors that already sit on the device
        cont_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
        cont = torch.optim.Adam(cont_p, lr=0.01, weight_decay=5e-4, foreach=False)
        cont.load_state_dict(copy.deepcopy(sd))
        for p, q, gr in zip(mine_p, cont_p, grads):
            p.grad, q.grad = (2 * gr).to(DEV), (2 * gr).to(DEV)
        mine.step()
        cont.step()
        torch.cuda.synchronize()
        for p, q in zip(mine_p, cont_p):
            torch.testing.assert_close(p.detach(), q.detach(), rtol=3e-6, atol=3e-7)
        assert all(st["step"].is_cuda and float(st["step"]) == 2.0 for st in mine.state.values())


def _segments_case(seed=0):
    """Hand-made pending second phases over a flat gradient buffer (the shape the backward kernels leave: row-major slabs,
    grouped slabs of 32, a column-blocked destination) + the parameters whose .grad are views of it, one parameter whose
    gradient no segment writes, and a scalar ("the loss") that is nobody's gradient."""
    import ctypes as C
    from acm_gnn_amd import _lib
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(7 * 192 + 64 + 96 + 5, device=DEV)
    views = {"w": flat[:7 * 192].view(7, 192), "b": flat[7 * 192:7 * 192 + 64], "v": flat[7 * 192 + 64:7 * 192 + 160],
             "odd": flat[7 * 192 + 160:]}
    loss = torch.zeros(1, device=DEV)
    nblk = 300
    # w: destination blocked by columns -- element e of row j goes to column (q / 64) * 64 + q % 64 of a [7, 192] matrix,
    #    written as three segments of 64 columns each out of one [nblk, 7 * 192] workspace?  (simpler, as the kernels do:)
    ws_w = torch.randn(nblk, 7 * 192, generator=g).to(DEV)
    ws_b = torch.randn(3, nblk, 32, generator=g).to(DEV)          # grouped slabs: 64 + 1 columns -> 3 groups of 32
    ws_v = torch.randn(nblk + 17, 100, generator=g).to(DEV)
    segs = (_lib.ReduceSeg * 8)()
    n = 0

    def add(partial, nb, row_stride, q0, ln, dst, inner, col_block=0, outer=0, block=0, elem=0):
        nonlocal n
        s = segs[n]
        s.partial, s.nblk, s.row_stride, s.q0, s.len = partial.data_ptr(), nb, row_stride, q0, ln
        s.dst, s.inner, s.col_block, s.outer_stride, s.block_stride, s.elem_stride = dst, inner, col_block, outer, block, elem
        n += 1

    add(ws_w, nblk, 7 * 192, 0, 7 * 192, views["w"].data_ptr(), 192, 64, 192, 64)       # rows of 192 = 3 blocks of 64
    add(ws_b, nblk, 32, 0, 64, views["b"].data_ptr(), 64, elem=nblk * 32)                  # two whole groups of 32
    add(ws_b, nblk, 32, 64, 1, loss.data_ptr(), 1, elem=nblk * 32)                         # the 65th column: the "loss"
    add(ws_v, nblk + 17, 100, 2, 96, views["v"].data_ptr(), 96)
    add(ws_v, nblk + 17, 100, 98, 0, views["v"].data_ptr(), 1)                             # an empty segment
    want = {"w": ws_w.double().sum(0).view(7, 192), "b": ws_b.double().sum(1).reshape(-1)[:64],
            "loss": ws_b.double().sum(1).reshape(-1)[64], "v": ws_v.double().sum(0)[2:98]}
    lst = _lib.ReduceList(n, 8, C.cast(segs, C.POINTER(_lib.ReduceSeg)))
    keep = (segs, ws_w, ws_b, ws_v, flat)
    return lst, views, loss, want, keep


@pytest.mark.parametrize("mode", ["fused", "partly_covered", "no_arrive"])
@pytest.mark.parametrize("decoupled,wd", [(True, 1e-2), (False, 5e-4)])
def test_adam_step_flushes_the_pending_reductions_itself(mode, decoupled, wd):
    """acm_adam_config_t.pending (ABI 23): one launch = the step's deferred second phases + the update.  Bit-identical to
    acm_reduce_flush followed by acm_adam_step: parameters, both moments, the gradients themselves (still stored), the
    non-gradient output, the step counters and also_advance -- over three steps, with a parameter no segment writes (its
    blocks follow the reducing blocks in the grid) and a 5000-element one spread over several update blocks.  The two
    modes that cannot run as one grid (a gradient the segments write only in part; no arrival counter) must give the same
    numbers through the flush-then-update route."""
    import ctypes as C
    from acm_gnn_amd import _lib
    lib = _lib.load()
    results = []
    for fused in (False, True):
        lst, views, loss, want, keep = _segments_case()
        g = torch.Generator().manual_seed(1)
        names = ["w", "b", "v", "odd", "free", "big"]
        grads = dict(views)
        grads["free"] = torch.randn(33, generator=g).to(DEV)
        grads["big"] = torch.randn(5000, generator=g).to(DEV)
        grads["odd"].copy_(torch.randn(5, generator=g))
        if mode == "partly_covered":                          # "v" becomes the front part of a longer gradient
            names = ["w", "b", "vlong", "free", "big"]
            flat = keep[-1]
            grads["vlong"] = flat[7 * 192 + 64:]              # v (96, written by a segment) + odd (5, not written)
        params = {k: torch.randn(grads[k].shape, generator=g).to(DEV) for k in names}
        m = {k: torch.zeros_like(params[k]) for k in names}
        v = {k: torch.zeros_like(params[k]) for k in names}
        steps = {k: torch.zeros((), device=DEV) for k in names}
        counter = torch.zeros(1, dtype=torch.int64, device=DEV)
        arrive = torch.zeros(1, dtype=torch.int32, device=DEV)
        entries = (_lib.AdamTensor * len(names))()
        for e, k in zip(entries, names):
            e.param, e.grad, e.exp_avg, e.exp_avg_sq = params[k].data_ptr(), grads[k].data_ptr(), m[k].data_ptr(), v[k].data_ptr()
            e.step, e.numel = steps[k].data_ptr(), params[k].numel()
        n_seg = lst.n
        for it in range(3):
            lst.n = n_seg
            cfg = _lib.AdamConfig(0.05, 0.9, 0.999, 1e-8, wd, int(decoupled), counter.data_ptr(),
                                  None if mode == "no_arrive" else arrive.data_ptr(), C.addressof(lst) if fused else None)
            if not fused:
                _lib.check(lib.acm_reduce_flush(C.byref(lst), None))
            _lib.check(lib.acm_adam_step(len(names), C.cast(entries, C.c_void_p), C.byref(cfg), None))
            assert lst.n == 0
        torch.cuda.synchronize()
        assert int(counter) == 3 and int(arrive) == 0 and all(float(s) == 3.0 for s in steps.values())
        for k in ("w", "b", "v"):
            torch.testing.assert_close(views[k].double(), want[k].view(views[k].shape), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(loss.double()[0], want["loss"], rtol=1e-5, atol=1e-4)
        results.append({"p": params, "m": m, "v": v, "g": {k: t.clone() for k, t in grads.items()}, "loss": loss.clone()})
    a, b = results
    assert torch.equal(a["loss"], b["loss"])
    for part in ("p", "m", "v", "g"):
        for k in a[part]:
            assert torch.equal(a[part][k], b[part][k]), (part, k)
    assert float((a["p"]["w"] - torch.zeros_like(a["p"]["w"])).abs().max()) > 0


def test_adam_step_with_more_pending_segments_than_one_grid_takes():
    """> 28 segments: the call flushes them through acm_reduce_flush's launches and then updates (same results)."""
    import ctypes as C
    from acm_gnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    n_seg, nblk = 40, 50
    out = []
    for fused in (False, True):
        ws = torch.randn(nblk, n_seg * 8, generator=torch.Generator().manual_seed(5)).to(DEV)
        grad = torch.zeros(n_seg * 8, device=DEV)
        p = torch.ones(n_seg * 8, device=DEV)
        m, v, step = torch.zeros_like(p), torch.zeros_like(p), torch.zeros((), device=DEV)
        arrive = torch.zeros(1, dtype=torch.int32, device=DEV)
        segs = (_lib.ReduceSeg * n_seg)()
        for i, s in enumerate(segs):
            s.partial, s.nblk, s.row_stride, s.q0, s.len = ws.data_ptr(), nblk, n_seg * 8, 8 * i, 8
            s.dst, s.inner = grad.data_ptr() + 32 * i, 8
        lst = _lib.ReduceList(n_seg, n_seg, C.cast(segs, C.POINTER(_lib.ReduceSeg)))
        e = (_lib.AdamTensor * 1)()
        e[0].param, e[0].grad, e[0].exp_avg, e[0].exp_avg_sq, e[0].step, e[0].numel = (p.data_ptr(), grad.data_ptr(), m.data_ptr(),
                                                                                     v.data_ptr(), step.data_ptr(), p.numel())
        cfg = _lib.AdamConfig(0.01, 0.9, 0.999, 1e-8, 0.0, 1, None, arrive.data_ptr(), C.addressof(lst) if fused else None)
        if not fused:
            _lib.check(lib.acm_reduce_flush(C.byref(lst), None))
        _lib.check(lib.acm_adam_step(1, C.cast(e, C.c_void_p), C.byref(cfg), None))
        torch.cuda.synchronize()
        assert lst.n == 0 and float(step) == 1.0
        torch.testing.assert_close(grad.double(), ws.double().sum(0), rtol=1e-5, atol=1e-4)
        out.append((p.clone(), m.clone(), v.clone(), grad.clone()))
    for x, y in zip(*out):
        assert torch.equal(x, y)
    del g


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_train_step_with_the_flush_inside_the_optimizer_launch(use_graph):
    """train.TrainStep(flush_in_optimizer=True) -- the default with FusedAdam / FusedAdamW -- against False: the same
    launches minus one, bit-identical parameters, moments and losses after 6 steps (dropout on: the counter advance rides
    the same launch)."""
    from acm_gnn_amd import GCN, FusedAdamW, data as D, functional as AF, train as T
    from acm_gnn_amd.distributed import make_sharded_operators
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=2)
    low, deg = D.build_filters(adj)
    ops = make_sharded_operators(low, deg, torch.device(DEV))
    x = torch.from_numpy(D.row_normalize_features(x_np)).to(DEV)
    y = torch.from_numpy(y_np).to(DEV)
    w = T.row_weights(torch.from_numpy(tr).to(DEV), x.shape[0])
    runs = []
    for flush_in_optimizer in (False, True):
        torch.manual_seed(0)
        model = GCN(x.shape[1], 64, int(y.max()) + 1, 2, x.shape[0], 0.3, "acmgcnp", 0, variant=False).to(DEV)
        model.dropout_state = AF.DropoutState(torch.device(DEV), seed=5)
        opt = FusedAdamW(model.parameters(), lr=0.02, weight_decay=1e-3)
        step = T.TrainStep(model, opt, x, ops, y, w, use_graph=use_graph, flush_in_optimizer=flush_in_optimizer)
        timer = AF.KernelTimer()
        if not use_graph:
            AF.set_kernel_timer(timer)
        losses = [float(step()) for _ in range(6)]
        AF.set_kernel_timer(None)
        labels = set(timer.summary()) if not use_graph else set()
        runs.append((losses, {k: v.clone() for k, v in model.state_dict().items()},
                     [opt.state[p]["exp_avg_sq"].clone() for p in model.parameters() if p in opt.state], labels))
    (la, sa, va, lab_a), (lb, sb, vb, lab_b) = runs
    assert la == lb and all(np.isfinite(la))
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for p, q in zip(va, vb):
        assert torch.equal(p, q)
    if not use_graph:
        assert "reduce_flush" in lab_a and "adam" in lab_a and "reduce_flush" not in lab_b and "adam+flush" in lab_b, (lab_a, lab_b)
