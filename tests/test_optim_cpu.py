"""FusedAdam / FusedAdamW host logic against torch.optim on CPU (kernel replaced by the numpy double):
same trajectories, same state_dict layout, hyper-parameter validation."""
import numpy as np
import pytest
import torch

import fake_lib


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(7, 64), (64, 1), (3, 3), (64,), (1, 1), (2500,), (5000, 3)]
    return [torch.randn(*s, generator=g) for s in shapes]


@pytest.mark.parametrize("decoupled,wd", [(False, 0.0), (False, 5e-4), (True, 0.0), (True, 1e-2)])
def test_trajectory_equals_torch_optim(decoupled, wd, monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import FusedAdam, FusedAdamW
    mine = [torch.nn.Parameter(t.clone()) for t in _params(0)]
    ref = [torch.nn.Parameter(t.clone()) for t in _params(0)]
    unused_a, unused_b = torch.nn.Parameter(torch.zeros(1, 1)), torch.nn.Parameter(torch.zeros(1, 1))
    kw = dict(lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    a = (FusedAdamW if decoupled else FusedAdam)(mine + [unused_a], **kw)
    b = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ref + [unused_b], foreach=False, **kw)
    g = torch.Generator().manual_seed(1)
    for it in range(25):
        for p, q in zip(mine, ref):
            gr = torch.randn(p.shape, generator=g) * (1.0 + it)
            p.grad, q.grad = gr.clone(), gr.clone()
        a.step()
        b.step()
    for p, q in zip(mine, ref):
        np.testing.assert_allclose(p.detach().numpy(), q.detach().numpy(), rtol=2e-6, atol=2e-7)
    assert unused_a not in a.state or len(a.state[unused_a]) == 0          # parameters without a gradient get no state
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa["state"].keys()) == set(sb["state"].keys())
    for k in sa["state"]:
        assert set(sa["state"][k].keys()) == {"step", "exp_avg", "exp_avg_sq"} == set(sb["state"][k].keys())
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 25.0
        for key in ("exp_avg", "exp_avg_sq"):
            want = sb["state"][k][key].numpy()
            np.testing.assert_allclose(sa["state"][k][key].numpy(), want, rtol=2e-6, atol=2e-6 * float(np.abs(want).max()))
    # a torch optimizer continues from our state and vice versa
    b2 = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ref + [unused_b], foreach=False, **kw)
    b2.load_state_dict(sa)
    a2 = (FusedAdamW if decoupled else FusedAdam)(mine + [unused_a], **kw)
    a2.load_state_dict(sb)
    for p, q in zip(mine, ref):
        gr = torch.randn(p.shape, generator=g)
        p.grad, q.grad = gr.clone(), gr.clone()
    a2.step()
    b2.step()
    for p, q in zip(mine, ref):
        np.testing.assert_allclose(p.detach().numpy(), q.detach().numpy(), rtol=3e-6, atol=3e-7)


def test_state_loaded_from_a_torch_checkpoint_is_made_kernel_ready(monkeypatch):
    """torch.optim.Adam keeps `step` as a CPU tensor (a Python number in old checkpoints) and a checkpoint may carry
    fp64 / transposed moments: load_state_dict must hand the kernel 0-dim fp32 `step` tensors on the parameter's device
    and contiguous fp32 moments (on a GPU a host `step` pointer would fault)."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import FusedAdam
    ps = [torch.nn.Parameter(t.clone()) for t in _params(3)[:3]]
    ref = torch.optim.Adam(ps, lr=0.01, foreach=False)
    for p in ps:
        p.grad = torch.ones_like(p)
    ref.step()
    sd = ref.state_dict()
    sd["state"][0]["step"] = 1                                  # legacy: Python number
    sd["state"][1]["exp_avg"] = sd["state"][1]["exp_avg"].double()
    sd["state"][2]["step"] = sd["state"][2]["step"].reshape(1)
    opt = FusedAdam(ps, lr=0.01)
    opt.load_state_dict(sd)
    for p in ps:
        st = opt.state[p]
        assert st["step"].dim() == 0 and st["step"].dtype == torch.float32 and st["step"].device == p.device
        assert float(st["step"]) == 1.0
        for k in ("exp_avg", "exp_avg_sq"):
            assert st[k].dtype == torch.float32 and st[k].is_contiguous() and st[k].shape == p.shape
    opt.step()
    assert all(float(opt.state[p]["step"]) == 2.0 for p in ps)


def test_argument_validation(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import FusedAdam, FusedAdamW
    p = [torch.nn.Parameter(torch.zeros(3))]
    with pytest.raises(NotImplementedError):
        FusedAdam(p, amsgrad=True)
    with pytest.raises(ValueError):
        FusedAdamW(p, lr=-1.0)
    with pytest.raises(ValueError):
        FusedAdam(p, betas=(1.0, 0.9))
    assert FusedAdamW(p).defaults["weight_decay"] == 1e-2 and FusedAdam(p).defaults["weight_decay"] == 0.0
    FusedAdam(p, capturable=True, foreach=None)            # torch-only switches are accepted
    opt = FusedAdam([torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))])
    opt.param_groups[0]["params"][0].grad = torch.zeros(3, dtype=torch.float64)
    with pytest.raises(TypeError):
        opt.step()


def test_no_cpu_path_without_the_double():
    from acm_gnn_amd import FusedAdam
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FusedAdam([p]).step()
