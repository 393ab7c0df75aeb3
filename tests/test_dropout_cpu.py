"""Counter-based dropout, host side (numpy double of the ABI): the mask is a pure function of
(seed, step, tag, row, col); the fused-dropout model equals the F.dropout model fed the same masks;
gradients flow through the regenerated mask; the step counter advances once per optimizer step."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fake_lib
from conftest import graph_tensors


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-7 (kat_vectors: counter / key all zero, all ones, pi digits)."""
    w = fake_lib.philox7_words(0, 0, 0, np.array([0]), np.array([0]))[:, 0]
    assert [hex(int(v)) for v in w] == ["0x5f6fb709", "0xd893f64", "0x4f121f81", "0x4f730a48"]
    seed = 0xFFFFFFFFFFFFFFFF
    w = fake_lib.philox7_words(seed, 0xFFFFFFFFFFFFFFFF, 0xFFFF, np.array([0xFFFFFFFF]), np.array([0xFFFF]))[:, 0]
    assert [hex(int(v)) for v in w] == ["0x5207ddc2", "0x45165e59", "0x4d8ee751", "0x8c52f662"]
    # counter = 243f6a88 85a308d3 13198a2e 03707344, key = a4093822 299f31d0 (digits of pi)
    w = fake_lib.philox7_words((0x299f31d0 << 32) | 0xa4093822, (0x03707344 << 32) | 0x13198a2e, 0x85a3,
                               np.array([0x243f6a88]), np.array([0x08d3]))[:, 0]
    assert [hex(int(v)) for v in w] == ["0x4dfccaba", "0x190a87f0", "0xc47362ba", "0xb6b5242a"]


def test_mask_statistics_and_determinism(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import functional as AF
    st = AF.DropoutState("cpu", seed=1234)
    x = torch.ones(4000, 70)
    a = AF.dropout(x, 0.3, st, tag=0)
    b = AF.dropout(x, 0.3, st, tag=0)
    assert torch.equal(a, b)                                   # same (seed, step, tag): same mask
    keep = (a > 0).float().mean().item()
    assert abs(keep - 0.7) < 0.005 and torch.allclose(a[a > 0], torch.tensor(1 / 0.7))
    assert not torch.equal(a, AF.dropout(x, 0.3, st, tag=1))   # another tag: another mask
    st.advance()
    c = AF.dropout(x, 0.3, st, tag=0)
    assert not torch.equal(a, c) and abs((c > 0).float().mean().item() - 0.7) < 0.005
    # columns are uncorrelated with rows / each other (every (row, col) has its own word)
    m = (a > 0).float()
    assert abs(np.corrcoef(m[:, 0], m[:, 16])[0, 1]) < 0.05 and abs(np.corrcoef(m[0], m[1])[0, 1]) < 0.3
    # padded output: extra columns are exact zeros, leading columns unchanged
    p = AF.dropout(x[:, :7], 0.3, AF.DropoutState("cpu", seed=1234), tag=0, pad_to=8)
    assert p.shape == (4000, 8) and torch.equal(p[:, 7], torch.zeros(4000)) and torch.equal(p[:, :7], a[:, :7])
    # a row shard draws the single-process mask
    sh = AF.dropout(x[1000:2000], 0.3, AF.DropoutState("cpu", seed=1234), tag=0, row_offset=1000)
    assert torch.equal(sh, a[1000:2000])


@pytest.mark.parametrize("model_type,s,variant,f_in", [("acmgcnp", 0, 0, 7), ("acmgcnp", 1, 0, 7), ("acmgcn", 0, 1, 7),
                                                       ("acmgcnpp", 0, 0, 7), ("acmgcnp", 0, 0, 40)])
def test_fused_dropout_model_equals_mask_replay(model_type, s, variant, f_in, monkeypatch):
    """GCN(fused_dropout=True) == the same GCN on F.dropout fed the masks the kernels regenerate."""
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, functional as AF
    low, high, un, _ = graph_tensors("geometric")
    n, p = low.shape[0], 0.4
    torch.manual_seed(0)
    model = GCN(f_in, 64, 3, 2, n, p, model_type, s, variant=bool(variant), attn_layernorm=True)
    x = torch.randn(n, f_in)
    y = torch.randint(0, 3, (n,))
    model.train()
    model.fused_dropout = True
    model.dropout_state = AF.DropoutState("cpu", seed=99)
    out = model(x, low, high, un if s else None)
    F.nll_loss(F.log_softmax(out, 1), y).backward()
    got = {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    # reference: masks as tensors, through the patched F.dropout path
    st = model.dropout_state
    ones = lambda r, c, tag: AF.dropout(torch.ones(r, c), p, st, tag=tag)      # noqa: E731
    masks = [ones(n, f_in, 0)] + ([ones(n, 64, 2)] if model_type == "acmgcnpp" else []) + [ones(n, 64, 1)]
    model.fused_dropout = False
    model.zero_grad()
    monkeypatch.setattr(F, "dropout", lambda t, p_=0.5, training=True, inplace=False: t * masks.pop(0))
    ref = model(x, low, high, un if s else None)
    F.nll_loss(F.log_softmax(ref, 1), y).backward()
    assert not masks
    np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    for k, v in model.named_parameters():
        if v.grad is None:
            assert k not in got
            continue
        np.testing.assert_allclose(got[k].numpy(), v.grad.numpy(), rtol=1e-4, atol=1e-5 * float(v.grad.abs().max()) + 1e-8,
                                   err_msg=k)


def test_train_step_advances_the_counter_once_per_step(monkeypatch):
    fake_lib.install(monkeypatch)
    from acm_gnn_amd import GCN, FusedAdamW, data as D, train as T
    from acm_gnn_amd.graph import CsrGraph, FilterOperators
    adj, x_np, y_np, (tr, _, _), _ = D.synthetic_dataset("tiny", seed=1)
    low, _ = D.build_filters(adj)
    ops = FilterOperators(CsrGraph.from_scipy(low, "cpu"))
    x, y = torch.from_numpy(D.row_normalize_features(x_np)), torch.from_numpy(y_np)
    w = T.row_weights(torch.from_numpy(tr), x.shape[0])
    for fused_opt in (True, False):
        torch.manual_seed(0)
        model = GCN(7, 16, 2, 2, x.shape[0], 0.5, "acmgcnp", 0)
        opt = FusedAdamW(model.parameters(), lr=0.01) if fused_opt else torch.optim.AdamW(model.parameters(), lr=0.01)
        step = T.TrainStep(model, opt, x, ops, y, w)
        assert model.fused_dropout and model.dropout_state is not None
        losses = [float(step()) for _ in range(4)]
        assert int(model.dropout_state.step) == 4 and all(np.isfinite(losses))
    model.eval()
    model(x, ops)                                               # (the first pass over a static input leaves P = A_low X for the next ones)
    assert torch.equal(model(x, ops), model(x, ops))            # eval: no dropout
    off = T.TrainStep(GCN(7, 16, 2, 2, x.shape[0], 0.5, "acmgcnp", 0), opt, x, ops, y, w, fused_dropout=False)
    assert not off.model.fused_dropout
