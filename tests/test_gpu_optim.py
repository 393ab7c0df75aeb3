"""acm_adam_step on the MI355X against torch.optim.Adam / AdamW (single-tensor CPU implementation):
vector and scalar paths, unaligned views, more than one pack of 32 tensors, hipGraph replay."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _shapes():
    return [(7, 64), (64, 1), (3, 3), (64,), (1, 1), (2500,), (5000, 3), (168114, 8), (4099,)] + [(5, i + 1) for i in range(30)]


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
    assert len(mine) > 32
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
